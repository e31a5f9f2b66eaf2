// ls_dispatch.cuh -- host-side launch sequence of the lock-step path (included by dispatch_lockstep.hip).
#pragma once
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "engine_host.h"
#include "lockstep.cuh"
#include "team.cuh"

// The launch sequence of one search.  The batch is cut into e->opt.ls_pipes independent pipelines (ranges of tree-group pairs),
// pipeline p on stream p: per simulation step a tree kernel (+ first layer) and one kernel per hidden->hidden layer.  The
// pipelines share nothing but read-only data, so a pipeline's small latency-bound tree kernel runs beside the others' layer
// kernels.  (Measured at config E: two pipelines +6 % time, four +40 %; a captured hipGraph of the sequence ran in the same
// time as the plain launches, the path is not host-bound.  One pipeline is the default.)
template <int ENV, int HP, bool GMM>
static hipError_t ls_enqueue(azg_engine* e, hipStream_t main) {
    constexpr int NS = HP / 256, NCH = HP / 64;
    const int G = (e->cfg.n_trees + TREES_PER_WG - 1) / TREES_PER_WG;
    const size_t tab_bytes = ((size_t)e->tab_n * 8 + (size_t)(e->cfg.n_sims + 2) * 4 + 15) / 16 * 16;
    const size_t act_bytes = (size_t)HP * 64;
    const bool fuse0 = e->opt.ls_fuse0 == 1;
    auto tk = fuse0 ? ls_tree_kernel<ENV, GMM, NCH, HP, true> : ls_tree_kernel<ENV, GMM, NCH, HP, false>;
    auto hk = ls_hidden_kernel<HP, false>;
    auto hl = ls_hidden_kernel<HP, true>;
    if (act_bytes > 48 * 1024) {
        hipError_t rc = hipFuncSetAttribute((const void*)hk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)act_bytes);
        if (rc == hipSuccess) rc = hipFuncSetAttribute((const void*)hl, hipFuncAttributeMaxDynamicSharedMemorySize, (int)act_bytes);
        if (rc != hipSuccess) return rc;
    }
    // hidden layers: the LDS-tiled kernel; AZG_LS_TILED=0 keeps the 16-tree x 256-unit weight-streaming kernel (diagnostics)
    // (32 trees x 64 units per workgroup: two workgroups per CU at 1024 trees x 1024 units)
    auto tkh = ls_hidden_tiled_kernel<HP, false, 2, 4>;
    auto tkl = ls_hidden_tiled_kernel<HP, true, 2, 4>;
    auto tkh0 = ls_hidden_tiled_kernel<HP, false, 2, 4, true>;   // first hidden layer with the network's first layer made in its staging
    auto tkl0 = ls_hidden_tiled_kernel<HP, true, 2, 4, true>;
    const int TQ_all = (G + 1) / 2, NU = HP / 64;
    const size_t tiled_bytes = (size_t)2 * (4 + 2) * LS_KC * 64 * 16;   // two stages of A (4 tiles) + B (2 groups)
    const bool tiled = e->opt.ls_tiled != 0;
    const bool l0in = tiled && e->opt.ls_fuse0 == 2;
    if (tiled) {
        hipError_t rc = hipFuncSetAttribute((const void*)tkh, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tiled_bytes);
        if (rc == hipSuccess) rc = hipFuncSetAttribute((const void*)tkl, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tiled_bytes);
        if (rc == hipSuccess) rc = hipFuncSetAttribute((const void*)tkh0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tiled_bytes);
        if (rc == hipSuccess) rc = hipFuncSetAttribute((const void*)tkl0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tiled_bytes);
        if (rc != hipSuccess) return rc;
    }
    int pipes = e->opt.ls_pipes;
    if (pipes > TQ_all) pipes = TQ_all;
    if (pipes > LS_MAX_PIPES) pipes = LS_MAX_PIPES;
    if (pipes < 1) pipes = 1;
    // fork: the other pipelines' streams start behind everything already queued on the main stream
    if (pipes > 1) {
        hipError_t rc = hipEventRecord(e->ls_fork, main);
        for (int p = 1; p < pipes && rc == hipSuccess; ++p) rc = hipStreamWaitEvent(e->ls_streams[p], e->ls_fork, 0);
        if (rc != hipSuccess) return rc;
    }
    for (int p = 0; p < pipes; ++p) {
        hipStream_t st = p == 0 ? main : e->ls_streams[p];
        const int tq0 = (int)((long)TQ_all * p / pipes), tq1 = (int)((long)TQ_all * (p + 1) / pipes);
        const int TQ = tq1 - tq0, g_base = 2 * tq0;
        int Gp = 2 * TQ;                      // tree groups of this pipeline (the last one may end on an odd group)
        if (g_base + Gp > G) Gp = G - g_base;
        hipLaunchKernelGGL(tk, dim3(Gp), dim3(256), tab_bytes, st, e->P, e->ls, -2, g_base);
        for (int sim = -1; sim < e->cfg.n_sims; ++sim) {
            if (!fuse0 && !l0in) hipLaunchKernelGGL((ls_layer0_kernel<HP>), dim3(Gp * NS), dim3(256), 0, st, e->P, e->ls, g_base);
            for (int l = 1; l < e->n_hidden; ++l) {
                const bool last = l == e->n_hidden - 1;
                if (tiled) {
                    auto k = last ? (l == 1 && l0in ? tkl0 : tkl) : (l == 1 && l0in ? tkh0 : tkh);
                    hipLaunchKernelGGL(k, dim3(TQ * NU), dim3(256), tiled_bytes, st, e->P, e->ls, l, (l - 1) & 1, TQ, g_base);
                } else if (last) {
                    hipLaunchKernelGGL(hl, dim3(Gp * NS), dim3(256), act_bytes, st, e->P, e->ls, l, (l - 1) & 1, g_base);
                } else {
                    hipLaunchKernelGGL(hk, dim3(Gp * NS), dim3(256), act_bytes, st, e->P, e->ls, l, (l - 1) & 1, g_base);
                }
            }
            hipLaunchKernelGGL(tk, dim3(Gp), dim3(256), tab_bytes, st, e->P, e->ls, sim, g_base);
        }
    }
    // join
    for (int p = 1; p < pipes; ++p) {
        hipError_t rc = hipEventRecord(e->ls_join[p], e->ls_streams[p]);
        if (rc == hipSuccess) rc = hipStreamWaitEvent(main, e->ls_join[p], 0);
        if (rc != hipSuccess) return rc;
    }
    return hipGetLastError();
}

// The persistent team kernel (team.cuh) when every workgroup of its grid can be resident at once; hipErrorNotReady: not here.
template <int ENV, int HP, bool GMM>
static hipError_t ls_team_run(azg_engine* e) {
    constexpr int NU = HP / 64;
    if (e->n_hidden - 1 >= TEAM_CNT_L0) return hipErrorNotReady;
    const int G = (e->cfg.n_trees + TREES_PER_WG - 1) / TREES_PER_WG, TQ = (G + 1) / 2;
    const size_t lds = (size_t)LS_TILE_STAGE_F4 * 16 + (size_t)e->tab_n * 8 + (size_t)(e->cfg.n_sims + 2) * 4;
    auto kern = ls_team_kernel<ENV, HP, GMM>;
    static std::atomic<int> per_cu_cache{-1};
    int per_cu = per_cu_cache.load(std::memory_order_relaxed);
    if (per_cu < 0 || lds != e->ls_team_lds) {
        hipError_t rc = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (rc != hipSuccess) return rc;
        rc = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, 256, lds);
        if (rc != hipSuccess) return rc;
        per_cu_cache.store(per_cu, std::memory_order_relaxed);
        e->ls_team_lds = lds;
    }
    // the occupancy query can answer one block per CU too many where the SGPR file is what limits residency; for 256-thread
    // blocks that limit is floor(800 / (ceil(sgpr / 16) * 16 + 16)) >= 6 whatever the kernel's sgpr count (<= 112): answers
    // up to 6 are safe to take as they are (and every wait in the kernel is bounded should this ever be wrong)
    const int usable = per_cu < 6 ? per_cu : 6;
    if (usable < 1 || (long)TQ * NU > (long)usable * e->n_cus) return hipErrorNotReady;
    hipError_t rc = hipMemsetAsync(e->d_team_cnt, 0, e->team_cnt_bytes, e->stream);
    if (rc != hipSuccess) return rc;
    TeamCtl T;
    T.cnt = e->d_team_cnt;
    T.abort = e->d_team_cnt + (e->team_cnt_bytes / 4 - 1);   // the last word
    hipLaunchKernelGGL(kern, dim3(TQ * NU), dim3(256), lds, e->stream, e->P, e->ls, T, TQ);
    e->team_pending = 1;
    return hipGetLastError();
}

template <int ENV, int HP, bool GMM>
static hipError_t ls_run(azg_engine* e) {
    if (e->opt.ls_team) {
        hipError_t rc = ls_team_run<ENV, HP, GMM>(e);
        if (rc != hipErrorNotReady) return rc;
    }
    return ls_enqueue<ENV, HP, GMM>(e, e->stream);
}

template <int ENV>
static hipError_t ls_dispatch(azg_engine* e) {
    const bool gmm = ENV != AZG_ENV_CARTPOLE && e->P.ncomp >= 2;
    if (e->HP == 512) {
        if constexpr (ENV != AZG_ENV_CARTPOLE) { if (gmm) return ls_run<ENV, 512, true>(e); }
        return ls_run<ENV, 512, false>(e);
    }
    if constexpr (ENV != AZG_ENV_CARTPOLE) { if (gmm) return ls_run<ENV, 1024, true>(e); }
    return ls_run<ENV, 1024, false>(e);
}

