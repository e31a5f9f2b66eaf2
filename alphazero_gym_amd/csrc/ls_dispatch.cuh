// ls_dispatch.cuh -- host-side launch sequence of the lock-step path (included by dispatch_lockstep.hip).
#pragma once
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "engine_host.h"
#include "lockstep.cuh"

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
    const int TQ_all = (G + 1) / 2, NU = HP / 64;
    // two stages of A (4 tiles) + B (2 groups); the weights-direct tile stages the activations only (and passes the head chain through
    // the first TG * 64 entries)
    const size_t tiled_bytes = LS_LAYER_WD ? (size_t)2 * 2 * LS_LAYER_KC * 64 * 16 : (size_t)2 * (4 + 2) * LS_KC * 64 * 16;
    const bool tiled = e->opt.ls_tiled != 0;
    if (tiled) {
        hipError_t rc = hipFuncSetAttribute((const void*)tkh, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tiled_bytes);
        if (rc == hipSuccess) rc = hipFuncSetAttribute((const void*)tkl, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tiled_bytes);
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
            if (!fuse0) hipLaunchKernelGGL((ls_layer0_kernel<HP>), dim3(Gp * NS), dim3(256), 0, st, e->P, e->ls, g_base);
            for (int l = 1; l < e->n_hidden; ++l) {
                const bool last = l == e->n_hidden - 1;
                if (tiled) {
                    auto k = last ? tkl : tkh;
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

