// team_dispatch.cuh -- host-side choice and launch of the persistent team kernel (team.cuh); included by dispatch_team.hip.
#pragma once
#include <atomic>
#include <cstdlib>

#include "engine_host.h"
#include "team.cuh"

// chunk lengths of the forms (k-blocks per staged chunk)
#ifndef TEAM_KC32
#define TEAM_KC32 2         // 32-tree teams, two workgroups per CU (weights-direct tile, MI355X, ms per search at 1024 trees: 2: 12.78, 4: 13.2, 8: 13.24)
#endif
#ifndef TEAM_KC32W
#define TEAM_KC32W 2        // 32-tree teams, three per CU (1536 trees: 2: 17.96, 4: 18.48)
#endif
#ifndef TEAM_KC64
#define TEAM_KC64 2         // 64-tree teams, two / three per CU (2048 trees: 2: 22.45, 4: 23.04; 3072: 32.13 / 33.73)
#endif

// One variant: hipErrorNotReady when its workgroups cannot all be resident at once (or the shape is not its).
// wider_form_exists: a form built for more workgroups per CU follows for this shape, so this one takes batches of up to MINB per CU only;
// otherwise it takes whatever the occupancy query allows (up to 6).  dry: check only, launch nothing.
template <int ENV, int HP, bool GMM, int TLDS, int KC, int MINB, int SPEC = 0, int TT = 32>
static hipError_t team_launch_form(azg_engine* e, bool wider_form_exists, int g_base, int G, bool dry = false) {
    constexpr int NU = HP / 64, TPW = TT / NU, TGN = TT / 16;
    constexpr bool CONT = EnvFamily<ENV>::CONT;
    if (e->n_hidden - 1 >= TEAM_CNT_XB) return hipErrorNotReady;   // (one counter per hidden layer)
    const int TQ = (G + TGN - 1) / TGN;   // (this launch: G tree groups from g_base)
    const bool lds_cold = TEAM_LDS_COLD && HP == 1024 && TT == 32 && MINB == 2 && TLDS == TS_LDS8;
    const size_t lds = team_tree_off(e->tab_n, e->cfg.n_sims, KC, TGN) + (size_t)TPW * (team_tree_bytes(e->R, CONT, TLDS) + (lds_cold ? team_cold_bytes(e->R) : 0));
    if ((lds + 1024) * MINB > 160 * 1024) return hipErrorNotReady;
    auto kern = ls_team_kernel<ENV, HP, GMM, TLDS, KC, MINB, SPEC, TT>;
    // (per device: the dynamic-LDS attribute belongs to the device's copy of the kernel)
    static std::atomic<int> per_cu_caches[AZG_MAX_DEVICES];
    static std::atomic<size_t> lds_caches[AZG_MAX_DEVICES];
    std::atomic<int>& per_cu_cache = per_cu_caches[e->cfg.device_id % AZG_MAX_DEVICES];
    std::atomic<size_t>& lds_cache = lds_caches[e->cfg.device_id % AZG_MAX_DEVICES];
    int per_cu = per_cu_cache.load(std::memory_order_relaxed);
    if (per_cu <= 0 || lds != lds_cache.load(std::memory_order_relaxed)) {
        hipError_t rc = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (rc != hipSuccess) return rc;
        rc = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, 256, lds);
        if (rc != hipSuccess) return rc;
        per_cu_cache.store(per_cu, std::memory_order_relaxed);
        lds_cache.store(lds, std::memory_order_relaxed);
    }
    // the occupancy query can answer one block per CU too many where the SGPR file is what limits residency; for 256-thread
    // blocks that limit is floor(800 / (ceil(sgpr / 16) * 16 + 16)) >= 6 whatever the kernel's sgpr count (<= 112): answers
    // up to 6 are safe to take as they are (and every wait in the kernel is bounded should this ever be wrong)
    int usable = per_cu < 6 ? per_cu : 6;
    if (wider_form_exists && usable > MINB) usable = MINB;   // (the next form takes the larger batches)
    if (usable < 1 || (long)TQ * NU > (long)usable * e->n_cus) return hipErrorNotReady;
    if (dry) return hipSuccess;   // (residency check only: team_launch asks about every part of a two-part search before it launches the first)
    hipError_t rc = g_base == 0 ? hipMemsetAsync(e->d_team_cnt, 0, e->team_cnt_bytes, e->stream) : hipSuccess;
    if (rc != hipSuccess) return rc;
    TeamCtl T;
    T.cnt = e->d_team_cnt;
    T.abort = e->d_team_cnt + (e->team_cnt_bytes / 4 - 1);   // the last word
    T.spin_limit = (unsigned)e->opt.team_spin_limit;
    hipLaunchKernelGGL(kern, dim3(TQ * NU), dim3(256), lds, e->stream, e->P, e->ls, T, TQ, g_base);
    e->team_pending = 1;
    e->kernel_form = 2;
    e->tree_lds = TLDS;
    e->dyn_lds = lds;
    e->team_kc = KC; e->team_minb = MINB; e->spec = SPEC; e->team_tt = TT;
    return hipGetLastError();
}

// Forms by batch size (HP = 1024, LDS trees: the shapes that were measured; MI355X, 4x1024 network, 200 simulations):
//   <= 2 workgroups of 32-tree teams per CU (1024 trees: BASELINE config E per GPU)   long chunks, two per CU            13.1 ms
//   <= 3 per CU (1536 trees)                                                           short chunks, three per CU         18.8 ms
//   <= 2 workgroups of 64-TREE teams per CU (2048 trees)   64 x 64 tiles: a third fewer staged bytes per MFMA             23.3 ms (0.70 of the
//        fp32 MFMA peak; round 4's four 32-tree workgroups per CU, 62 spilled VGPRs: 24.3-25.0 ms -- removed)
//   <= 3 of those per CU (3072 trees)                                                                                      32.9 ms (0.75; per-layer
//        launches: 38.2 ms)
// AZG_TEAM_WIDE=0: the first form only; AZG_TEAM_TT=32 / 64: only teams of that size (A/B runs).
template <int ENV, int HP, bool GMM, int TLDS>
static hipError_t team_launch_part(azg_engine* e, int g_base, int G, bool dry = false) {
    constexpr bool WIDE = HP == 1024 && !GMM && TLDS == TS_LDS8 && ENV == AZG_ENV_PENDULUM_V1;
    if constexpr (!WIDE) return team_launch_form<ENV, HP, GMM, TLDS, TEAM_KC32, 2>(e, false, g_base, G, dry);
    else {
        // (the BASELINE shape's tree phases compiled for the common parameter set -- dispatch.cuh: SPEC --; AZG_NO_SPEC=1: the general kernel)
        const bool common = e->cfg.epsilon == 0.0 && e->cfg.tie_break == AZG_TIE_FIRST && e->cfg.env_id == AZG_ENV_PENDULUM_V1 && !e->opt.no_spec;
        const bool wide = e->opt.team_wide != 0, t32 = e->opt.team_tt != 64, t64 = e->opt.team_tt != 32 && wide;
        hipError_t rc = hipErrorNotReady;
        if (t32) rc = common ? team_launch_form<ENV, HP, GMM, TLDS, TEAM_KC32, 2, 1>(e, wide, g_base, G, dry) : team_launch_form<ENV, HP, GMM, TLDS, TEAM_KC32, 2>(e, wide, g_base, G, dry);
        if (rc == hipErrorNotReady && wide) rc = azg_team_wide_forms(e, g_base, G, dry, common, t32, t64);   // (their own translation unit)
        return rc;
    }
}

// The forms for batches beyond two 32-tree workgroups per CU (BASELINE config E's network, LDS trees) live in their own translation unit,
// dispatch_team_wide.hip, which is compiled with -mllvm -disable-machine-licm: with every loop-invariant constant and address hoisted out
// of the simulation loop the three-per-CU forms spill (64-tree teams: 35 VGPRs, 140 B of scratch per lane; without the hoisting 6 / 24 B)
// and run 1-3 % slower (1536 trees 18.05 -> 17.54 ms, 3072 trees 32.11 -> 31.82); the two-per-CU 32-tree form of config E itself is
// 1.5 % FASTER with the hoisting (12.72 against 12.91 ms) and stays in dispatch_team.hip (MI355X, same box; profiles/r06_ab_experiments.txt).
#ifdef AZG_TEAM_WIDE_TU
hipError_t azg_team_wide_forms(azg_engine* e, int g_base, int G, bool dry, bool common, bool t32, bool t64) {
    constexpr int ENV = AZG_ENV_PENDULUM_V1, HP = 1024, TLDS = TS_LDS8;
    constexpr bool GMM = false;
    hipError_t rc = hipErrorNotReady;
    if (t32) rc = common ? team_launch_form<ENV, HP, GMM, TLDS, TEAM_KC32W, 3, 1>(e, true, g_base, G, dry) : team_launch_form<ENV, HP, GMM, TLDS, TEAM_KC32W, 3>(e, true, g_base, G, dry);
    // (64-tree teams: the long chunks' stages + four 200-simulation trees are 84 KB, two of that do not fit a CU; short chunks: 52 KB)
    if (rc == hipErrorNotReady && t64) rc = common ? team_launch_form<ENV, HP, GMM, TLDS, TEAM_KC64, 2, 1, 64>(e, true, g_base, G, dry) : team_launch_form<ENV, HP, GMM, TLDS, TEAM_KC64, 2, 0, 64>(e, true, g_base, G, dry);
    if (rc == hipErrorNotReady && t64) rc = common ? team_launch_form<ENV, HP, GMM, TLDS, TEAM_KC64, 3, 1, 64>(e, false, g_base, G, dry) : team_launch_form<ENV, HP, GMM, TLDS, TEAM_KC64, 3, 0, 64>(e, false, g_base, G, dry);
    return rc;
}
#endif

// Batches beyond the widest form (3072 trees at HP = 1024) run as TWO launches of equal parts, one after the other on the engine's
// stream (trees are independent; a launch's workgroups all have to be resident at once): 4096 trees 46.5 ms against 48.7 ms for the
// per-layer launches.  Those get better with the batch (8192 trees: 92.7 ms = 0.71 of the fp32 MFMA peak; three team launches: 99.2 ms):
// batches of more than two parts are theirs.
template <int ENV, int HP, bool GMM, int TLDS>
static hipError_t team_launch(azg_engine* e) {
    constexpr bool WIDE = HP == 1024 && !GMM && TLDS == TS_LDS8 && ENV == AZG_ENV_PENDULUM_V1;
    const int G = (e->cfg.n_trees + TREES_PER_WG - 1) / TREES_PER_WG;
    const int G_MAX = 3 * e->n_cus / (HP / 64) * 4;                      // 64-tree teams, three workgroups on every CU
    e->team_parts = 1;
    if (!WIDE || G <= G_MAX || !e->opt.team_wide || e->opt.team_tt == 32) return team_launch_part<ENV, HP, GMM, TLDS>(e, 0, G);
    const int parts = (G + G_MAX - 1) / G_MAX, per = ((G + parts - 1) / parts + 3) / 4 * 4;   // (whole 64-tree teams)
    if (parts > 2) return hipErrorNotReady;
    // both parts' residency (occupancy, LDS) is checked before the first one is launched: a second part that does not fit
    // would otherwise send the whole search to another form behind an orphaned first launch
    hipError_t rc = hipSuccess;
    for (int g = 0; g < G && rc == hipSuccess; g += per) rc = team_launch_part<ENV, HP, GMM, TLDS>(e, g, G - g < per ? G - g : per, true);
    if (rc != hipSuccess) return rc;
    for (int g = 0; g < G && rc == hipSuccess; g += per) rc = team_launch_part<ENV, HP, GMM, TLDS>(e, g, G - g < per ? G - g : per);
    e->team_parts = parts;
    return rc;
}

// trees in the workgroups' LDS when they fit (same rule as the persistent search kernel's), else in global memory
template <int ENV, int HP, bool GMM>
static hipError_t team_storage(azg_engine* e) {
    const long nmax = (long)e->carry_max + e->cfg.n_sims + 2;
    hipError_t rc = hipErrorNotReady;
    if (e->Kp == 16 && !e->opt.force_global_tree) {
        if (e->R <= 255 && nmax < 65536) rc = team_launch<ENV, HP, GMM, TS_LDS8>(e);
        else if (e->R <= 511 && nmax < 2048) rc = team_launch<ENV, HP, GMM, TS_LDS9>(e);
    }
    if (rc == hipErrorNotReady) rc = team_launch<ENV, HP, GMM, TS_GLOBAL>(e);
    return rc;
}

template <int ENV>
static hipError_t team_dispatch(azg_engine* e) {
    const bool gmm = EnvFamily<ENV>::CONT && e->P.ncomp >= 2;
    if (e->HP == 512) {
        if constexpr (EnvFamily<ENV>::CONT) { if (gmm) return team_storage<ENV, 512, true>(e); }
        return team_storage<ENV, 512, false>(e);
    }
    if (e->HP == 1024) {
        if constexpr (EnvFamily<ENV>::CONT) { if (gmm) return team_storage<ENV, 1024, true>(e); }
        return team_storage<ENV, 1024, false>(e);
    }
    return hipErrorNotReady;
}
