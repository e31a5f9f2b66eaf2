// ls_dispatch.cuh -- host-side launch sequence of the lock-step path (included by dispatch_lockstep.hip).
#pragma once
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "engine_host.h"
#include "lockstep.cuh"

// The launch sequence of one search: per simulation step a tree kernel, the first layer and one LDS-tiled kernel per hidden->hidden
// layer, all on the engine's stream.  (A captured hipGraph of the sequence ran in the same time as the plain launches: the path is
// not host-bound.  Cutting the batch into pipelines on several streams measured +6 % / +40 % time at config E: removed, HISTORY.md.)
template <int ENV, int HP, bool GMM>
static hipError_t ls_enqueue(azg_engine* e, hipStream_t st) {
    constexpr int NS = HP / 256, NCH = HP / 64;
    const int G = (e->cfg.n_trees + TREES_PER_WG - 1) / TREES_PER_WG;
    const size_t tab_bytes = ((size_t)e->tab_n * 8 + (size_t)(e->cfg.n_sims + 2) * 4 + 15) / 16 * 16;
    auto tk = ls_tree_kernel<ENV, GMM, NCH, HP>;
    // hidden layers: 32 trees x 64 units per workgroup (two workgroups per CU at 1024 trees x 1024 units)
    auto tkh = ls_hidden_tiled_kernel<HP, false, 2, 4>;
    auto tkl = ls_hidden_tiled_kernel<HP, true, 2, 4>;
    const int TQ = (G + 1) / 2, NU = HP / 64;
    // two stages of A (4 tiles) + B (2 groups); the weights-direct tile stages the activations only (and passes the head chain through
    // the first TG * 64 entries)
    const size_t tiled_bytes = LS_LAYER_WD ? (size_t)2 * 2 * LS_LAYER_KC * 64 * 16 : (size_t)2 * (4 + 2) * LS_KC * 64 * 16;
    hipError_t rc = hipFuncSetAttribute((const void*)tkh, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tiled_bytes);
    if (rc == hipSuccess) rc = hipFuncSetAttribute((const void*)tkl, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tiled_bytes);
    if (rc != hipSuccess) return rc;
    hipLaunchKernelGGL(tk, dim3(G), dim3(256), tab_bytes, st, e->P, e->ls, -2, 0);
    for (int sim = -1; sim < e->cfg.n_sims; ++sim) {
        hipLaunchKernelGGL((ls_layer0_kernel<HP>), dim3(G * NS), dim3(256), 0, st, e->P, e->ls, 0);
        for (int l = 1; l < e->n_hidden; ++l) {
            auto k = l == e->n_hidden - 1 ? tkl : tkh;
            hipLaunchKernelGGL(k, dim3(TQ * NU), dim3(256), tiled_bytes, st, e->P, e->ls, l, (l - 1) & 1, TQ, 0);
        }
        hipLaunchKernelGGL(tk, dim3(G), dim3(256), tab_bytes, st, e->P, e->ls, sim, 0);
    }
    return hipGetLastError();
}

template <int ENV, int HP, bool GMM>
static hipError_t ls_run(azg_engine* e) {
    if (e->opt.ls_team) {
        hipError_t rc = ENV == AZG_ENV_CARTPOLE ? azg_team_dispatch_cartpole(e)
                                                : (ENV == AZG_ENV_MOUNTAINCAR_CONT ? azg_team_dispatch_mcc(e) : azg_team_dispatch_pendulum(e));
        if (rc != hipErrorNotReady) return rc;
    }
    e->kernel_form = 1;
    return ls_enqueue<ENV, HP, GMM>(e, e->stream);
}

template <int ENV>
static hipError_t ls_dispatch(azg_engine* e) {
    const bool gmm = EnvFamily<ENV>::CONT && e->P.ncomp >= 2;
    if (e->HP == 512) {
        if constexpr (EnvFamily<ENV>::CONT) { if (gmm) return ls_run<ENV, 512, true>(e); }
        return ls_run<ENV, 512, false>(e);
    }
    if constexpr (EnvFamily<ENV>::CONT) { if (gmm) return ls_run<ENV, 1024, true>(e); }
    return ls_run<ENV, 1024, false>(e);
}

