// dispatch.cuh -- host-side choice and launch of the persistent search kernel variants (included by the dispatch_*.hip
// translation units, each of which instantiates one family of them).
#pragma once
#include <hip/hip_ext.h>
#include <atomic>
#include <cstdlib>

#include "engine_host.h"
#include "env.cuh"
#include "mlp.cuh"
#include "tree.cuh"
#include "search_kernel.cuh"

// One kernel variant: checks that its LDS plan fits the 160 KB of a CU (static + dynamic), then launches.
// Returns hipErrorInvalidConfiguration (nothing launched) when it does not fit.
template <int ENV, int HP, int NREG, int TLDS, bool GMM, int NW, int NG, int NT = 16, int SPEC = 0>
static hipError_t launch_g(azg_engine* e) {
    auto kern = search_kernel<ENV, HP, NREG, TLDS, GMM, NW, NG, NT, SPEC>;
    static std::atomic<int> static_lds_cache{-1};   // per kernel variant; engines of several host threads may race to fill it
    int static_lds = static_lds_cache.load(std::memory_order_relaxed);
    if (static_lds < 0) {
        hipFuncAttributes fa;
        hipError_t rc = hipFuncGetAttributes(&fa, (const void*)kern);
        if (rc != hipSuccess) return rc;
        static_lds = (int)fa.sharedSizeBytes;
        static_lds_cache.store(static_lds, std::memory_order_relaxed);
    }
    // discrete LDS trees: the expanded nodes' env states go to LDS too when the CU has room for them -- and when that does not
    // cost a second resident workgroup: a batch with more workgroups than CUs runs two of them side by side on a CU if their LDS
    // allows it, which is worth far more (CartPole, 8192 trees, 2x128: 0.62 ms per search against 0.94 ms)
    constexpr bool CONT = EnvFamily<ENV>::CONT;
    LdsLayout L = lds_layout(e->tab_n, e->cfg.n_sims, HP, NG, act_buffers(NREG), e->R, CONT, TLDS, 1, NT);
    const long n_wg = (e->cfg.n_trees + NT * NG - 1) / (NT * NG);
    const size_t with_state = L.total + (size_t)static_lds;
    const size_t without = lds_layout(e->tab_n, e->cfg.n_sims, HP, NG, act_buffers(NREG), e->R, CONT, TLDS, 0, NT).total + (size_t)static_lds;
    const bool costs_a_neighbour = n_wg > e->n_cus && 2 * without <= 160 * 1024 && 2 * with_state > 160 * 1024;
    e->P.lds_state = (!CONT && TLDS != TS_GLOBAL && with_state <= 160 * 1024 && !costs_a_neighbour && !getenv("AZG_NO_LDS_STATE")) ? 1 : 0;
    if (!e->P.lds_state) L = lds_layout(e->tab_n, e->cfg.n_sims, HP, NG, act_buffers(NREG), e->R, CONT, TLDS, 0, NT);
    if (L.total + (size_t)static_lds > 160 * 1024) return hipErrorInvalidConfiguration;
    if (L.total > 48 * 1024) {
        hipError_t rc = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.total);
        if (rc != hipSuccess) return rc;
    }
    const int tpw = NT * NG;
    dim3 grid((e->cfg.n_trees + tpw - 1) / tpw), block(64 * NW);
    e->tree_lds = TLDS;
    e->dyn_lds = L.total;
    e->waves = NW; e->groups = NG; e->tile_trees = NT; e->spec = SPEC;
    e->kernel_form = 0;
    // the launch carries its own start / stop events (azg_last_search_ms): stamps of the dispatch packet itself -- event records
    // around it are two more packets in the queue, 7 us per search on a stream of back-to-back searches (measured, config C)
    hipExtLaunchKernelGGL(kern, grid, block, (unsigned)L.total, e->stream, e->ev0, e->ev1, 0, e->P);
    e->launch_timed = 1;
    return hipGetLastError();
}

// Kernels specialised at compile time for the common parameter set (tree_phases.cuh: Spec<1> -- no epsilon-greedy selection,
// lowest-index ties, and in the discrete family CartPole's two actions) exist for the shapes the BASELINE configurations and their
// neighbours run on: one register-resident hidden->hidden layer, full tiles, trees in LDS with 8-bit ids.  Everything else, and
// every engine whose parameters differ, runs the general kernels (AZG_NO_SPEC=1 forces them: tests).
template <int ENV, int HP, int NREG, int TLDS, int NW, int NG, int NT = 16>
static hipError_t launch_t(azg_engine* e) {
    if constexpr (EnvFamily<ENV>::CONT && NW == 4 && NT == 16 && TLDS != TS_LDS9) {   // (mixture heads: LDS trees of up to 255 records, else global)
        if (e->P.ncomp >= 2) return launch_g<ENV, HP, NREG, TLDS, true, NW, NG>(e);
    }
    if constexpr (NREG == 1 && TLDS == TS_LDS8 && NT == 16 && ENV != AZG_ENV_ACROBOT) {
        // (+ Pendulum-v1 in the Pendulum family; + no carried root count beyond the sqrt table in the discrete family)
        const bool common = e->cfg.epsilon == 0.0 && e->cfg.tie_break == AZG_TIE_FIRST &&
                            (ENV != AZG_ENV_PENDULUM_V1 || e->cfg.env_id == AZG_ENV_PENDULUM_V1) &&
                            (ENV != AZG_ENV_CARTPOLE || (e->cfg.env_id == AZG_ENV_CARTPOLE && e->cfg.num_actions == 2 &&
                                                         (long)e->carry_max + e->cfg.n_sims + 2 <= (long)e->tab_n));
        if (common && !e->opt.no_spec) return launch_g<ENV, HP, NREG, TLDS, false, NW, NG, NT, 1>(e);
    }
    return launch_g<ENV, HP, NREG, TLDS, false, NW, NG, NT>(e);
}

// Variant choice.  Trees live in LDS when they fit (8-bit record ids, 16-bit counts, <= 16 children per node, the CU's 160 KB).
// The general shape is the 4-wave / 16-tree workgroup.  2x256 squashed-Normal networks (continuous mode) run with eight waves: while
// every 16-tree group can have a CU of its own, as 16-tree workgroups whose eight waves share the network phase and whose first four
// walk the trees (4 % faster than four waves; with all eight walking it only tied); once a batch has more groups than the device has
// CUs, as 32-tree workgroups: two groups per network phase, the second wave of a SIMD in the other's LDS / memory waits (1.27x at 8192 trees).
// AZG_WAVES=4|8, AZG_GROUPS=1|2 force a shape (tests).
template <int ENV, int HP, int NREG>
static hipError_t launch(azg_engine* e) {
    const int ns = e->cfg.n_sims;
    // LDS trees: <= 16 children per node; 8-bit ids / 16-bit counts up to 255 records, 9-bit ids / 11-bit counts up to 511
    // (node counts: the root's is the largest, carried count + n_sims)
    int ts = TS_GLOBAL;
    const long nmax = (long)e->carry_max + ns + 2;
    e->lds_exit = AZG_LDS_EXIT_CHILDREN;          // (what azg_search_info reports when the trees end up in global memory)
    if (e->Kp == 16) {
        e->lds_exit = AZG_LDS_EXIT_RECORDS;
        if (e->R <= 255 && nmax < 65536) ts = TS_LDS8;
        else if (e->R <= 511 && nmax < 2048) ts = TS_LDS9;
        if (ts != TS_GLOBAL) e->lds_exit = AZG_LDS_EXIT_SIZE;   // from here on only the CU's 160 KB can push them out
    }
    // Trees of 256 .. 511 records stay in LDS (9-bit ids) for the squashed-Normal / discrete heads of networks up to 256 wide; the
    // mixture head's and the wide networks' kernels exist for 8-bit ids and for global trees only (round 6: 32 instantiations fewer)
    if (ts == TS_LDS9 && (HP >= 512 || e->P.ncomp >= 2)) { ts = TS_GLOBAL; e->lds_exit = AZG_LDS_EXIT_RECORDS; }
    if (e->opt.force_global_tree) { ts = TS_GLOBAL; e->lds_exit = AZG_LDS_EXIT_FORCED; }
    // (Continuous mode only.  The discrete family's 8-wave shapes were measured slower than its 4-wave ones -- CartPole, 8192 trees,
    // 2x256: 1.03 ms against 0.99 ms per search, and they were the only kernels of the family that spilled registers -- and are gone.)
    if constexpr (HP == 256 && NREG == 1 && EnvFamily<ENV>::CONT) {
        bool two = (e->cfg.n_trees + 15) / 16 > e->n_cus;
        if (e->opt.groups == 2) two = true;
        if (e->opt.groups == 1) two = false;
        // eight waves either way (round 4): 32-tree workgroups when the batch has more groups than CUs, else 16-tree workgroups whose
        // eight waves share the network phase while four of them walk the trees (search_kernel.cuh).  AZG_WAVES=4 forces the old shape.
        bool want8 = true;
        if (e->opt.waves == 4) want8 = false;
        if (want8 && ts == TS_LDS8 && e->P.ncomp < 2) {
            hipError_t rc = hipErrorInvalidConfiguration;
            if (two) rc = launch_t<ENV, HP, NREG, TS_LDS8, 8, 2>(e);
            if (rc == hipErrorInvalidConfiguration) rc = launch_t<ENV, HP, NREG, TS_LDS8, 8, 1>(e);
            if (rc != hipErrorInvalidConfiguration) return rc;
        }
    }
    // Half-filled tiles (small register-resident networks, LDS trees): batches of at most 8 trees per CU run as 8-tree workgroups
    // (search_kernel.cuh has the measurements: with more trees than that, full tiles win).  AZG_TILE_TREES=16|8 forces a shape (tests).
    if constexpr (HP <= 128 && NREG == 1) {
        if (ts == TS_LDS8 && e->P.ncomp < 2) {
            int nt = e->cfg.n_trees <= 8L * e->n_cus ? 8 : 16;
            if (e->opt.tile_trees) nt = e->opt.tile_trees;
            if (nt == 8) {
                hipError_t rc = launch_t<ENV, HP, NREG, TS_LDS8, 4, 1, 8>(e);
                if (rc != hipErrorInvalidConfiguration) return rc;
            }
        }
    }
    if (ts == TS_LDS8) {
        hipError_t rc = launch_t<ENV, HP, NREG, TS_LDS8, 4, 1>(e);
        if (rc != hipErrorInvalidConfiguration) return rc;
    }
    if constexpr (HP < 512) {
        if (ts == TS_LDS9) {
            hipError_t rc = launch_t<ENV, HP, NREG, TS_LDS9, 4, 1>(e);
            if (rc != hipErrorInvalidConfiguration) return rc;
        }
    }
    return launch_t<ENV, HP, NREG, TS_GLOBAL, 4, 1>(e);
}

// hidden widths (padded) up to 128
template <int ENV>
static hipError_t dispatch_small(azg_engine* e) {
    const int HP = e->HP, NR = e->nreg;
    // (one or two hidden->hidden layers are kept in registers; deeper trunks stream their weights from L2: any depth)
    if (HP == 64) {
        if (NR == 1) return launch<ENV, 64, 1>(e);
        if (NR == 2) return launch<ENV, 64, 2>(e);
        return launch<ENV, 64, 0>(e);
    }
    if (HP == 128) {
        if (NR == 1) return launch<ENV, 128, 1>(e);
        if (NR == 2) return launch<ENV, 128, 2>(e);
        return launch<ENV, 128, 0>(e);
    }
    return hipErrorInvalidValue;
}
// 256 and wider
template <int ENV>
static hipError_t dispatch_large(azg_engine* e) {
    const int HP = e->HP, NR = e->nreg;
    if (HP == 256) {
        if (NR == 1) return launch<ENV, 256, 1>(e);
        return launch<ENV, 256, 0>(e);
    }
    if (HP == 512) return launch<ENV, 512, 0>(e);
    if (HP == 1024) return launch<ENV, 1024, 0>(e);
    return hipErrorInvalidValue;
}

