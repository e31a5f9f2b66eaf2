// pair_dispatch.cuh -- host-side choice and launch of the walker + server kernel pair (pair.cuh); included by dispatch_pair.hip.
#pragma once
#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "engine_host.h"
#include "pair.cuh"

static hipError_t pair_buffers(azg_engine* e, int n_pairs, int per) {
    if (!e->pair_stream) {
        hipError_t rc = hipStreamCreateWithFlags(&e->pair_stream, hipStreamNonBlocking);
        if (rc == hipSuccess) rc = hipEventCreateWithFlags(&e->pair_fork, hipEventDisableTiming);
        if (rc == hipSuccess) rc = hipEventCreateWithFlags(&e->pair_join, hipEventDisableTiming);
        if (rc != hipSuccess) return rc;
    }
    if (n_pairs <= e->pair_alloc_pairs && per <= e->pair_alloc_per) return hipSuccess;
    if (e->d_pair_obs) (void)hipFree(e->d_pair_obs);
    if (e->d_pair_parts) (void)hipFree(e->d_pair_parts);
    if (e->d_pair_cnt) (void)hipFree(e->d_pair_cnt);
    e->d_pair_obs = nullptr; e->d_pair_parts = nullptr; e->d_pair_cnt = nullptr; e->pair_alloc_pairs = 0;
    e->pair_cnt_words = (size_t)n_pairs * PAIR_CNT_PER * PAIR_CNT_STRIDE + 64;   // + 16 tickets, + the abort word (the last one)
    hipError_t rc = hipMalloc((void**)&e->d_pair_obs, (size_t)n_pairs * 128 * sizeof(float));
    if (rc == hipSuccess) rc = hipMalloc((void**)&e->d_pair_parts, (size_t)n_pairs * 2 * per * sizeof(f32x4));
    if (rc == hipSuccess) rc = hipMalloc((void**)&e->d_pair_cnt, e->pair_cnt_words * sizeof(unsigned));
    if (rc != hipSuccess) return rc;
    e->pair_alloc_pairs = n_pairs; e->pair_alloc_per = per;
    return hipSuccess;
}

// One variant: hipErrorNotReady when the pair does not fit a CU side by side.
template <int ENV, int HP, int NREG, int TLDS, bool GMM>
static hipError_t pair_launch(azg_engine* e) {
    constexpr bool CONT = ENV != AZG_ENV_CARTPOLE;
    constexpr int PER = head_chunks<HP>() * (GMM ? 64 : 16);
    auto walker = pair_walker_kernel<ENV, HP, TLDS, GMM>;
    auto server = pair_server_kernel<HP, NREG, GMM>;
    const size_t w_lds = pair_tree_off(e->tab_n, e->cfg.n_sims) + 32 * pair_tree_bytes(e->R, CONT, TLDS);
    const size_t s_lds = (size_t)act_buffers(NREG) * HP * 64;
    // (per device: the dynamic-LDS attribute belongs to the device's copy of the kernel; stored: 0 = not looked at yet, 1 = does not fit, 2 = fits)
    static std::atomic<int> fit_caches[AZG_MAX_DEVICES];
    static std::atomic<size_t> lds_caches[AZG_MAX_DEVICES];
    std::atomic<int>& fit_cache = fit_caches[e->cfg.device_id % AZG_MAX_DEVICES];
    std::atomic<size_t>& lds_cache = lds_caches[e->cfg.device_id % AZG_MAX_DEVICES];
    int fit = fit_cache.load(std::memory_order_relaxed) - 1;
    if (fit < 0 || w_lds != lds_cache.load(std::memory_order_relaxed)) {
        hipFuncAttributes fw, fs;
        hipError_t rc = hipFuncGetAttributes(&fw, (const void*)walker);
        if (rc == hipSuccess) rc = hipFuncGetAttributes(&fs, (const void*)server);
        if (rc != hipSuccess) return rc;
        // side by side on a CU: LDS (160 KB) and the SIMD's 512 registers per lane (allocated in blocks of 8; numRegs counts
        // the unified vector file of this part: architectural + accumulation registers)
        const size_t lds_total = w_lds + fw.sharedSizeBytes + s_lds + fs.sharedSizeBytes;
        const int regs = (fw.numRegs + 7) / 8 * 8 + (fs.numRegs + 7) / 8 * 8;
        fit = (lds_total <= 160 * 1024 && regs <= 512) ? 1 : 0;
        if (getenv("AZG_DEBUG"))
            fprintf(stderr, "azgym pair: walker %d regs, %zu + %zu B LDS; server %d regs, %zu + %zu B LDS -> %s\n", fw.numRegs, w_lds,
                    (size_t)fw.sharedSizeBytes, fs.numRegs, s_lds, (size_t)fs.sharedSizeBytes, fit ? "fits" : "does not fit");
        if (fit) {
            rc = hipFuncSetAttribute((const void*)walker, hipFuncAttributeMaxDynamicSharedMemorySize, (int)w_lds);
            if (rc != hipSuccess) return rc;
        }
        fit_cache.store(fit + 1, std::memory_order_relaxed);
        lds_cache.store(w_lds, std::memory_order_relaxed);
    }
    if (!fit) return hipErrorNotReady;
    const int n_pairs = (e->cfg.n_trees + 31) / 32;
    hipError_t rc = pair_buffers(e, n_pairs, PER);
    if (rc != hipSuccess) return rc;
    rc = hipMemsetAsync(e->d_pair_cnt, 0, e->pair_cnt_words * sizeof(unsigned), e->stream);
    if (rc != hipSuccess) return rc;
    PairCtl T;
    T.obs = e->d_pair_obs; T.parts = e->d_pair_parts; T.cnt = e->d_pair_cnt;
    T.ticket = e->d_pair_cnt + (e->pair_cnt_words - 64);
    T.abort = e->d_pair_cnt + (e->pair_cnt_words - 1);
    T.spin_limit = (unsigned)e->opt.team_spin_limit;
    T.n_pairs = n_pairs;
    // both grids: the same multiple of 8 (every XCD gets grid / 8 workgroups of each kernel), one workgroup of each per CU at most
    const int cus8 = e->n_cus / 8 * 8;
    if (cus8 < 8) return hipErrorNotReady;
    const int grid = n_pairs < cus8 ? (n_pairs + 7) / 8 * 8 : cus8;
    T.per_xcd = grid / 8;
    // the server on its own stream behind the counters' reset, the walker on the engine's; the engine's stream then waits for both
    if ((rc = hipEventRecord(e->pair_fork, e->stream)) != hipSuccess) return rc;
    if ((rc = hipStreamWaitEvent(e->pair_stream, e->pair_fork, 0)) != hipSuccess) return rc;
    hipLaunchKernelGGL(server, dim3(grid), dim3(256), s_lds, e->pair_stream, e->P, T);
    if ((rc = hipGetLastError()) != hipSuccess) return rc;
    hipLaunchKernelGGL(walker, dim3(grid), dim3(256), w_lds, e->stream, e->P, T);
    if ((rc = hipGetLastError()) != hipSuccess) return rc;
    if ((rc = hipEventRecord(e->pair_join, e->pair_stream)) != hipSuccess) return rc;
    if ((rc = hipStreamWaitEvent(e->stream, e->pair_join, 0)) != hipSuccess) return rc;
    e->pair_pending = 1;
    e->kernel_form = 3;
    e->tree_lds = TLDS;
    e->dyn_lds = w_lds;
    e->waves = 4; e->groups = 2;
    return hipSuccess;
}

// The pair serves the networks whose hidden->hidden weights stay in registers, with trees in LDS (8-bit ids).
template <int ENV>
static hipError_t pair_dispatch(azg_engine* e) {
    const bool many = (e->cfg.n_trees + 15) / 16 > e->n_cus;
    if (!(e->opt.pair == 2 || (e->opt.pair == 1 && many))) return hipErrorNotReady;
    if (e->opt.force_global_tree || e->opt.waves || e->opt.groups) return hipErrorNotReady;   // (a forced shape means the one-kernel form)
    const long nmax = (long)e->carry_max + e->cfg.n_sims + 2;
    if (!(e->Kp == 16 && e->R <= 255 && nmax < 65536)) return hipErrorNotReady;
    if (e->P.ncomp >= 2) return hipErrorNotReady;
    if (e->HP == 256 && e->nreg == 1) return pair_launch<ENV, 256, 1, TS_LDS8, false>(e);
    if (e->HP == 128 && e->nreg == 1) return pair_launch<ENV, 128, 1, TS_LDS8, false>(e);
    return hipErrorNotReady;
}
