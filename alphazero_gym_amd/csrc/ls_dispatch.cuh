// ls_dispatch.cuh -- host-side launch sequence of the lock-step path (included by dispatch_lockstep.hip).
#pragma once
#include <cstdlib>

#include "engine_host.h"
#include "lockstep.cuh"

template <int ENV, int HP, bool GMM>
static hipError_t ls_run(azg_engine* e) {
    constexpr int NS = HP / 256, NCH = HP / 64;
    const int G = (e->cfg.n_trees + TREES_PER_WG - 1) / TREES_PER_WG;
    const size_t tab_bytes = ((size_t)e->tab_n * 8 + (size_t)(e->cfg.n_sims + 2) * 4 + 15) / 16 * 16;
    const size_t act_bytes = (size_t)HP * 64;
    auto tk = ls_tree_kernel<ENV, GMM, NCH>;
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
    const int TQ = (G + 1) / 2, NU = HP / 64;
    const size_t tiled_bytes = (size_t)2 * (4 + 2) * LS_KC * 64 * 16;   // two stages of A (4 tiles) + B (2 groups)
    const bool tiled = e->opt.ls_tiled != 0;
    if (tiled) {
        hipError_t rc = hipFuncSetAttribute((const void*)tkh, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tiled_bytes);
        if (rc == hipSuccess) rc = hipFuncSetAttribute((const void*)tkl, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tiled_bytes);
        if (rc != hipSuccess) return rc;
    }
    hipLaunchKernelGGL(tk, dim3(G), dim3(256), tab_bytes, e->stream, e->P, e->ls, -2);
    for (int sim = -1; sim < e->cfg.n_sims; ++sim) {
        hipLaunchKernelGGL((ls_layer0_kernel<HP>), dim3(G * NS), dim3(256), 0, e->stream, e->P, e->ls);
        for (int l = 1; l < e->n_hidden; ++l) {
            const bool last = l == e->n_hidden - 1;
            if (tiled) {
                if (last) hipLaunchKernelGGL(tkl, dim3(TQ * NU), dim3(256), tiled_bytes, e->stream, e->P, e->ls, l, (l - 1) & 1, TQ);
                else hipLaunchKernelGGL(tkh, dim3(TQ * NU), dim3(256), tiled_bytes, e->stream, e->P, e->ls, l, (l - 1) & 1, TQ);
            } else if (last) {
                hipLaunchKernelGGL(hl, dim3(G * NS), dim3(256), act_bytes, e->stream, e->P, e->ls, l, (l - 1) & 1);
            } else {
                hipLaunchKernelGGL(hk, dim3(G * NS), dim3(256), act_bytes, e->stream, e->P, e->ls, l, (l - 1) & 1);
            }
        }
        hipLaunchKernelGGL(tk, dim3(G), dim3(256), tab_bytes, e->stream, e->P, e->ls, sim);
    }
    return hipGetLastError();
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

