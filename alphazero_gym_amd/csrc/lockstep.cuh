// lockstep.cuh -- wide policy/value networks (hidden width >= 512; BASELINE config E: 4x1024): one simulation step = a few
// grid-wide launches instead of one persistent kernel, so that a network layer of ALL trees is spread over ALL CUs.
//   ls_tree_kernel     one workgroup per 16 trees: phase A (finish leaf, backup) + phase B (select, step, expand); trees and
//                      the per-tree state live in global memory between launches
//   ls_layer0_kernel   first layer for (tree group, 256-unit slice)
//   ls_hidden_kernel   one hidden->hidden layer for (tree group, 256-unit slice): the group's activations are staged in LDS,
//                      the slice's weights stream from L2 (blockIdx % 8 selects the XCD and blockIdx % NS the slice, so every
//                      XCD's L2 holds exactly one 1 MB slice); the last layer also leaves the partial head sums
// The arithmetic (MFMA chains, chunked head sums) is the persistent kernel's, bit for bit.
#pragma once
#include "records.h"
#include "env.cuh"
#include "mlp.cuh"
#include "tree.cuh"
#include "tree_phases.cuh"

struct LsLane { int my_depth, pid; double pr, pW; float eps_c; float pad; };
struct LsTree { int nrec; unsigned eps_draws; int leaf, need_eval, path_D, kbase; };

struct LockStep {
    float* obsT;        // [G][4][16]
    f32x4* act[2];      // [G][HP/16][64]   ping-pong activations (D-register layout)
    f32x4* parts;       // [G][HP/64][64]   partial head sums, one per 64-unit chunk
    int* any;           // [G]              some tree of the group needs an evaluation this step
    LsTree* tree;       // [B]
    LsLane* lane;       // [B][16]
};

template <int ENV, bool GMM, int NCH>
__global__ __launch_bounds__(256) void ls_tree_kernel(KParams P, LockStep L, int sim) {
    constexpr bool CONT = (ENV != AZG_ENV_CARTPOLE);
    extern __shared__ double s_dyn[];   // sqrt_tab [tab_n], pw_need [n_sims+2]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, sub = lane & 15;
    const int tl = wave * 4 + (lane >> 4);
    const int tg = blockIdx.x;
    const int tree = tg * TREES_PER_WG + tl;
    const bool live = tree < P.B;
    const unsigned gtree = (unsigned)(P.tree_base + tree);
    double* s_sqrt = s_dyn;
    int* s_pw = (int*)(s_dyn + P.tab_n);
    for (int i = tid; i < P.tab_n; i += 256) s_sqrt[i] = P.sqrt_tab[i];
    if (CONT) for (int i = tid; i < P.n_sims + 2; i += 256) s_pw[i] = P.pw_need[i];
    __syncthreads();
    const size_t tb = (size_t)(live ? tree : 0) * P.R;
    Cold* cold = P.cold + tb;
    double* edge_W = P.edge_W + tb;
    float* action = P.action + tb;
    TreeStore<false> ts;
    ts.hot = P.hot + tb;
    ts.child = P.child + tb * P.Kp;
    ts.prior = P.prior + tb;
    float* obsT = L.obsT + (size_t)tg * 64;
    TreeState st;
    if (sim == -2) {
        tree_init_root<ENV, false>(P, st, ts, cold, edge_W, action, tree, live, sub, tl, gtree, obsT);
    } else {
        const LsTree t = L.tree[live ? tree : 0];
#ifdef AZG_STAMPS
        unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // diagnostic build: discarded here
#endif
        const LsLane ln = L.lane[(size_t)(live ? tree : 0) * 16 + sub];
        st.nrec = t.nrec; st.eps_draws = t.eps_draws; st.leaf = t.leaf; st.need_eval = live && t.need_eval; st.path_D = t.path_D;
        st.kbase = t.kbase; st.my_depth = ln.my_depth; st.pid = ln.pid; st.pr = ln.pr; st.pW = ln.pW; st.eps_c = ln.eps_c;
        if (live) tree_phase_a<ENV, false, GMM, NCH>(P, st, ts, cold, edge_W, action, tb, sim, sub, tl, gtree,
                                                     L.parts + (size_t)tg * NCH * 64, P.bhead);
        st.need_eval = false;
        if (sim < P.n_sims - 1) {
            __threadfence_block();
            if (live) tree_phase_b<ENV, false, GMM>(P, st, ts, cold, edge_W, action, tb, sub, tl, gtree, s_sqrt, s_pw, obsT STAMP_ARG);
            else if (sub < 4) obsT[sub * 16 + tl] = 0.0f;
        } else if (live && sub == 0) {
            P.n_rec[tree] = st.nrec;
        }
    }
    if (live) {
        if (sub == 0) {
            LsTree t;
            t.nrec = st.nrec; t.eps_draws = st.eps_draws; t.leaf = st.leaf; t.need_eval = st.need_eval ? 1 : 0; t.path_D = st.path_D;
            t.kbase = st.kbase;
            L.tree[tree] = t;
        }
        LsLane ln;
        ln.my_depth = st.my_depth; ln.pid = st.pid; ln.pr = st.pr; ln.pW = st.pW; ln.eps_c = st.eps_c; ln.pad = 0.0f;
        L.lane[(size_t)tree * 16 + sub] = ln;
    }
    int any = __syncthreads_or(st.need_eval ? 1 : 0);
    if (tid == 0) L.any[tg] = any;
}

template <int HP>
__global__ __launch_bounds__(256) void ls_layer0_kernel(KParams P, LockStep L) {
    constexpr int NS = HP / 256;
    const int tg = blockIdx.x / NS, sl = blockIdx.x % NS;
    if (!L.any[tg]) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float b = L.obsT[(size_t)tg * 64 + lane];
    f32x4* out = L.act[0] + (size_t)tg * (HP / 16) * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int tile = sl * 16 + wave * 4 + i;
        f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x4f32(P.W0[tile * 64 + lane], b, P.b0[tile * 64 + lane], 0, 0, 0);
        out[tile * 64 + lane] = act4<true>(P.act, acc);
    }
}

template <int HP, bool LAST>
__global__ __launch_bounds__(256) void ls_hidden_kernel(KParams P, LockStep L, int layer, int in_buf) {
    constexpr int NS = HP / 256, S4 = HP / 16;
    extern __shared__ f32x4 s_in[];   // the tree group's input activations: HP/16 tiles x 64 lanes
    const int tg = blockIdx.x / NS, sl = blockIdx.x % NS;
    if (!L.any[tg]) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: weight addresses stay in SGPRs
    const f32x4* bb = P.bl[layer - 1];
    const int t0 = sl * 16 + wave * 4;   // this wave's 4 output tiles
    const f32x4* W = P.Wl[layer - 1] + (size_t)t0 * S4 * 64 + lane;   // + (i * S4 + s4) * 64 with wave-uniform i, s4
    // weight stream: DEPTH k-blocks (4 tiles x 16 B per lane each) are kept in flight per wave -- with one block in flight the
    // loop ran at L2 latency (29 GB/s per CU), not at the matrix pipe's rate
    constexpr int DEPTH = 8;
    static_assert(S4 % DEPTH == 0, "k-blocks per layer must be a multiple of the prefetch depth");
    f32x4 q[DEPTH][4];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int i = 0; i < 4; ++i) q[d][i] = W[(i * S4 + d) * 64];
    // stage the group's activations: all loads first (one round trip), then the LDS stores
    const f32x4* in = L.act[in_buf] + (size_t)tg * S4 * 64;
    constexpr int NST = S4 * 64 / 256;
    f32x4 stg[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) stg[i] = in[i * 256 + tid];
#pragma unroll
    for (int i = 0; i < NST; ++i) s_in[i * 256 + tid] = stg[i];
    __syncthreads();
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = bb[(t0 + i) * 64 + lane];
#pragma unroll 1
    for (int s4 = 0; s4 < S4; s4 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const f32x4 b = s_in[(s4 + d) * 64 + lane];
            f32x4 a[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = q[d][i];
            const int sn = s4 + d + DEPTH < S4 ? s4 + d + DEPTH : S4 - 1;   // (the tail re-reads the last block: harmless)
#pragma unroll
            for (int i = 0; i < 4; ++i) q[d][i] = W[(i * S4 + sn) * 64];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b.x, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b.y, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b.z, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b.w, acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);   // keep this block's loads here: the scheduler otherwise bunches all of an
                                                 // iteration's loads at its end and the next iteration waits for them at once
        }
    }
    f32x4 h[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = act4<true>(P.act, acc[i]);
    if (!LAST) {
        f32x4* out = L.act[in_buf ^ 1] + (size_t)tg * S4 * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) out[(t0 + i) * 64 + lane] = h[i];
    } else {
        // this wave's 64 units are one head chunk (chunk index = sl * 4 + wave): a chain from 0 over its 4 tiles
        f32x4 hs = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < 4; ++i) hs = mfma4(P.Whead[(t0 + i) * 64 + lane], h[i], hs);
        L.parts[((size_t)tg * (HP / 64) + sl * 4 + wave) * 64 + lane] = hs;
    }
}
