// search_kernel.cuh -- the persistent search kernel: one launch = all n_sims traces of B trees (16 trees per workgroup).
#pragma once
#include "records.h"
#include "env.cuh"
#include "mlp.cuh"
#include "tree.cuh"
#include "tree_phases.cuh"

template <int ENV, int HP, int NREG, bool TLDS, bool GMM>
__global__ __launch_bounds__(256, 1) void search_kernel(KParams P) {
    constexpr bool CONT = (ENV != AZG_ENV_CARTPOLE);
    constexpr int S = CONT ? 2 : 4;
    typedef typename TreeStore<TLDS>::Rec Rec;
    typedef typename TreeStore<TLDS>::Id Id;
    constexpr int NCH = head_chunks<HP>();   // partial head sums per tree
    __shared__ f32x4 s_parts[NCH * 64];
    __shared__ float s_obsT[4 * 16];
    __shared__ float s_bhead[16];
    __shared__ float s_ln[2 * 64];
    extern __shared__ double s_dyn[];   // sqrt_tab [tab_n], pw_need [n_sims+2] ints, two activation buffers, (TLDS) the 16 trees' hot records

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int sub = lane & 15;
    const int tl = wave * 4 + (lane >> 4);          // tree within the workgroup
    const int tree = blockIdx.x * TREES_PER_WG + tl;
    const bool live = tree < P.B;
    const unsigned gtree = (unsigned)(P.tree_base + tree);

    double* s_sqrt = s_dyn;
    int* s_pw = (int*)(s_dyn + P.tab_n);
    const size_t act_off = ((size_t)P.tab_n * 8 + (size_t)(P.n_sims + 2) * 4 + 15) / 16 * 16;
    f32x4* s_actA = (f32x4*)((char*)s_dyn + act_off);
    f32x4* s_actB = s_actA + HP / 16 * 64;
    for (int i = tid; i < P.tab_n; i += 256) s_sqrt[i] = P.sqrt_tab[i];
    if (CONT) for (int i = tid; i < P.n_sims + 2; i += 256) s_pw[i] = P.pw_need[i];
    if (tid < 16) s_bhead[tid] = P.bhead[tid];

    // register-resident weights
    WRegs<HP, NREG> wr;
    if constexpr (HP <= 256) {
        constexpr int NTW0 = HP / 64;
#pragma unroll
        for (int i = 0; i < NTW0; ++i) {
            wr.w0[i] = P.W0[(wave * NTW0 + i) * 64 + lane];
            wr.b0[i] = P.b0[(wave * NTW0 + i) * 64 + lane];
        }
    }
    if (NREG > 0) {
        constexpr int NTW = HP / 64, S4 = HP / 16;
#pragma unroll
        for (int l = 0; l < NREG; ++l) {
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                wr.b[l][i] = P.bl[l][(wave * NTW + i) * 64 + lane];
#pragma unroll
                for (int s4 = 0; s4 < S4; ++s4) wr.w[l][i][s4] = P.Wl[l][((wave * NTW + i) * S4 + s4) * 64 + lane];
            }
        }
#pragma unroll
        for (int i = 0; i < NTW; ++i) wr.wh[i] = P.Whead[(wave * NTW + i) * 64 + lane];
    }

    const size_t tb = (size_t)(live ? tree : 0) * P.R;
    Cold* cold = P.cold + tb;
    double* edge_W = P.edge_W + tb;
    float* action = P.action + tb;
    TreeStore<TLDS> ts;
    if (TLDS) {
        // per tree: R records of 16 B, then (continuous) R x Kp child ids or (discrete) R priors
        size_t off = act_off + (size_t)2 * HP * 64;
        size_t per = (size_t)P.R * 16 + (CONT ? (size_t)P.R * P.Kp : (size_t)P.R * 4);
        per = (per + 15) / 16 * 16;
        char* base = (char*)s_dyn + off + per * tl;
        ts.hot = (Rec*)base;
        ts.child = (Id*)(base + (size_t)P.R * 16);
        ts.prior = (float*)(base + (size_t)P.R * 16);
    } else {
        ts.hot = (Rec*)(P.hot + tb);
        ts.child = (Id*)(P.child + tb * P.Kp);
        ts.prior = P.prior + tb;
    }

#ifdef AZG_STAMPS
    unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    TreeState st;
    tree_init_root<ENV, TLDS>(P, st, ts, cold, edge_W, action, tree, live, sub, tl, gtree, s_obsT);
    __syncthreads();

    for (int sim = -1; sim < P.n_sims; ++sim) {
        // ================= network phase: evaluate the 16 pending leaves =================
        STAMP(t_a);
        int any = __syncthreads_or(st.need_eval ? 1 : 0);
        STAMP(t_b);
#ifdef AZG_STAMPS
        if (any) mlp_forward<HP, NREG>(P, wr, s_obsT, s_actA, s_actB, s_parts, s_ln, wave, lane, st_acc);
#else
        if (any) mlp_forward<HP, NREG>(P, wr, s_obsT, s_actA, s_actB, s_parts, s_ln, wave, lane);
#endif
        STAMP(t_c);
        // ================= tree phase A: finish the evaluated leaf, back up =================
        if (live) tree_phase_a<ENV, TLDS, GMM, NCH>(P, st, ts, cold, edge_W, action, tb, sim, sub, tl, gtree, s_parts, s_bhead);
        if (sim == P.n_sims - 1) break;
        __threadfence_block();
        STAMP(t_d);
        // ================= tree phase B: next trace: select down, step the env, expand =================
        st.need_eval = false;
        if (live) tree_phase_b<ENV, TLDS, GMM>(P, st, ts, cold, edge_W, action, tb, sub, tl, gtree, s_sqrt, s_pw, s_obsT);
        __threadfence_block();
        STAMP(t_e);
        STAMP_ADD(0, t_a, t_b);   // wait at the barrier in front of the network phase
        STAMP_ADD(1, t_b, t_c);   // network phase
        STAMP_ADD(2, t_c, t_d);   // finish leaf + backup
        STAMP_ADD(3, t_d, t_e);   // select / step / expand
    }
    const int nrec = st.nrec;
#ifdef AZG_STAMPS
    if (lane == 0) for (int i = 0; i < 16; ++i) P.stamps[((size_t)blockIdx.x * 4 + wave) * 16 + i] = st_acc[i];
#endif
    if (live) {
        if (sub == 0) P.n_rec[tree] = nrec;
        if (TLDS) {
            // publish the LDS-resident tree in the global format
            RecL* gh = P.hot + tb;
            for (int j = sub; j < nrec; j += 16) {
                Rec h = ts.hot[j];
                RecL o;
                o.Q = h.Q; o.edge_n = h.edge_n; o.node_n = h.node_n; o.parent = (short)h.parent; o.n_child = h.n_child;
                o.first = h.first; o.flags = h.flags; o.pad = 0;
                gh[j] = o;
                if (CONT) {
                    for (int i = 0; i < (int)h.n_child; ++i) P.child[(tb + j) * P.Kp + i] = ts.child[j * P.Kp + i];
                } else {
                    P.prior[tb + j] = ts.prior[j];
                }
            }
        }
    }
}
