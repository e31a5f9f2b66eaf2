// team_dispatch.cuh -- host-side choice and launch of the persistent team kernel (team.cuh); included by dispatch_team.hip.
#pragma once
#include <atomic>
#include <cstdlib>

#include "engine_host.h"
#include "team.cuh"

// One variant: hipErrorNotReady when its workgroups cannot all be resident at once (or the shape is not its).
// wider_form_exists: a form built for more workgroups per CU follows for this shape, so this one takes batches of up to MINB per CU only;
// otherwise it takes whatever the occupancy query allows (up to 6).
template <int ENV, int HP, bool GMM, int TLDS, int KC, int MINB, int SPEC = 0>
static hipError_t team_launch_form(azg_engine* e, bool wider_form_exists) {
    constexpr int NU = HP / 64, TPW = 32 / NU;
    constexpr bool CONT = EnvFamily<ENV>::CONT;
    if (e->n_hidden - 1 >= TEAM_CNT_XB) return hipErrorNotReady;   // (one counter per hidden layer)
    const int G = (e->cfg.n_trees + TREES_PER_WG - 1) / TREES_PER_WG, TQ = (G + 1) / 2;
    const size_t lds = team_tree_off(e->tab_n, e->cfg.n_sims, KC) + (size_t)TPW * team_tree_bytes(e->R, CONT, TLDS);
    if ((lds + 1024) * MINB > 160 * 1024) return hipErrorNotReady;
    auto kern = ls_team_kernel<ENV, HP, GMM, TLDS, KC, MINB, SPEC>;
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
    hipError_t rc = hipMemsetAsync(e->d_team_cnt, 0, e->team_cnt_bytes, e->stream);
    if (rc != hipSuccess) return rc;
    TeamCtl T;
    T.cnt = e->d_team_cnt;
    T.abort = e->d_team_cnt + (e->team_cnt_bytes / 4 - 1);   // the last word
    T.spin_limit = (unsigned)e->opt.team_spin_limit;
    hipLaunchKernelGGL(kern, dim3(TQ * NU), dim3(256), lds, e->stream, e->P, e->ls, T, TQ);
    e->team_pending = 1;
    e->kernel_form = 2;
    e->tree_lds = TLDS;
    e->dyn_lds = lds;
    e->team_kc = KC; e->team_minb = MINB; e->spec = SPEC;
    return hipGetLastError();
}

// Two workgroups per CU with the long chunks while the batch fits that (1024 trees at HP = 1024: BASELINE config E per GPU); three,
// then four per CU with short chunks for larger batches (HP = 1024, LDS trees: the shapes that were measured).
template <int ENV, int HP, bool GMM, int TLDS>
static hipError_t team_launch(azg_engine* e) {
    constexpr bool WIDE = HP == 1024 && !GMM && TLDS == TS_LDS8 && ENV == AZG_ENV_PENDULUM_V1;
    hipError_t rc;
    // (the BASELINE shape's tree phases compiled for the common parameter set -- dispatch.cuh: SPEC --; AZG_NO_SPEC=1: the general kernel)
    const bool common = WIDE && e->cfg.epsilon == 0.0 && e->cfg.tie_break == AZG_TIE_FIRST && e->cfg.env_id == AZG_ENV_PENDULUM_V1 && !e->opt.no_spec;
    if constexpr (WIDE) {
        if (common) rc = team_launch_form<ENV, HP, GMM, TLDS, LS_KC, 2, 1>(e, e->opt.team_wide);
        else rc = team_launch_form<ENV, HP, GMM, TLDS, LS_KC, 2>(e, e->opt.team_wide);
    } else {
        rc = team_launch_form<ENV, HP, GMM, TLDS, LS_KC, 2>(e, false);
    }
    if constexpr (WIDE) {
        if (rc == hipErrorNotReady && e->opt.team_wide) rc = team_launch_form<ENV, HP, GMM, TLDS, 2, 3>(e, true);
        if (rc == hipErrorNotReady && e->opt.team_wide) rc = team_launch_form<ENV, HP, GMM, TLDS, 2, 4>(e, false);
    }
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
