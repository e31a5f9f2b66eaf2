// search_q.cuh -- EXPERIMENT, NOT BUILT INTO THE LIBRARY (round 2): the persistent search kernel for 2x256 networks as an
// asynchronous pipeline of three wave roles.  Result on MI355X at BASELINE config C (4096 trees x 200 sims): bit-exact against
// the oracle on the whole parity suite, but 2.95 ms per search against 1.65 ms for the lock-step 16-tree workgroups.  Why:
// v_mfma_f32_4x4x1_16B_f32 is one instruction per ~10 cycles, most of which it holds the SIMD's vector issue port (the 16x16x4
// form holds it 8 cycles of 32).  With a stage-1 wave, a stage-2 wave and a tree wave on every SIMD, the matrix instructions'
// issue slots (2 x 4 quarters x ~170 instructions x ~10 cycles = 13.6k cycles per step) and the tree wave's ~1800 vector
// instructions (4 cycles each) no longer overlap: the port, not the matrix pipe, is what the SIMD runs out of.  The 16x16x4 form
// in the same pipeline would need >= 2 full 16-tree groups per CU to fill its 16 columns (i.e. >= 8192 trees), and those do
// not fit the LDS next to the hand-off buffers.  Kept for the record with tools/probes/mfma44.hip (the instruction's layout,
// fma semantics and rates).  To try it: copy into alphazero_gym_amd/csrc, add the q_* operand layouts to KParams /
// azg_set_weights and the launch to dispatch.cuh (git history of round 2 has both).
//
// search_kernel (search_kernel.cuh) runs a 16-tree workgroup in lock step: first layer, hidden layer, heads, tree phases, one
// after the other, so per simulation step the matrix pipes idle during the tree phases (a third of the step) and the vector
// ALUs during the hidden layer.  A step of ONE tree cannot be shortened -- select, evaluate, back up depend on each other --
// but the 16 trees of a CU need not move in step.  Here the workgroup has 12 waves (3 per SIMD) in three roles:
//     T  four tree waves, one QUARTER (4 trees, 16 lanes each) per wave: phases A + B of its trees, nothing else
//     A  four stage-1 waves: first half of the hidden layer's k range, for whichever quarter is ready
//     B  four stage-2 waves: second half of the k range, activation, value / policy heads
// Pair w (waves A_w, B_w, on one SIMD) owns output units [64w, 64w+64) of the hidden layer and keeps their 256x... weights split
// by k between the two waves (128 VGPRs each): A_w starts every accumulator from the bias and runs canonical positions 0..127,
// hands the accumulators to B_w through LDS, B_w continues 128..255 -- the same k-ordered fma chain as everywhere else.  The
// four quarters circulate through  T -> (first layer, A and B each their half of it) -> A -> B -> heads -> T  out of phase, each
// at the pace its own dependencies allow, synchronised by monotonic counters in LDS (one wave polls, no workgroup barrier
// after start-up).  The matrix pipe of a SIMD is fed by its A and B waves while its T wave walks trees.
//
// MFMA shape: v_mfma_f32_4x4x1_16B_f32 (sixteen independent 4x4 blocks, K = 1): rows = 64 output units, columns = the quarter's 4
// trees (replicated over the blocks), one fma per element and instruction -- measured on gfx950 (tools/probes/mfma44.hip): lane
// l supplies row l%4 of block l/4 (A) and column l%4 (B) and receives column l%4 of block l/4 in its 4 D registers; an exact
// float fma chain; 9.8 cycles per instruction (13.3 on a single accumulator) = 82 % of the 16x16x4 form's rate, which a 4-tree
// tile would use to a quarter.  Rows of a block are 4 consecutive canonical positions of the NEXT layer's input, so a lane's D
// float4 is stored as one ds_write_b128 into the [tree][position] activation buffers the next stage reads its B operand from.
// Heads: the 8 chunk chains (32 positions each, from 0) are the 16 blocks' accumulators (blocks 8..15 mirror 0..7): 32 steps.
#pragma once
#include "records.h"
#include "env.cuh"
#include "mlp.cuh"
#include "tree.cuh"
#include "tree_phases.cuh"

#define Q_HP 256
#define Q_WAVES 12
#define Q_SPIN_LIMIT (1u << 22)

// counters (LDS, monotonic): index
#define QC_OBS(q) (q)              // observations of quarter q for evaluation k published: k + 1
#define QC_XA(q) (4 + (q))         // first layer, positions 0..127 of quarter q: 4 (k + 1) when all four A waves are done
#define QC_XB(q) (8 + (q))         // positions 128..255 (B waves)
#define QC_S1(w, q) (12 + 4 * (w) + (q))   // pair w's stage-1 accumulators of quarter q handed over: k + 1
#define QC_H(q) (28 + (q))         // hidden layer of quarter q written: 4 (k + 1)
#define QC_PARTS(q) (32 + (q))     // head partials of quarter q: k + 1
#define QC_ABORT 36
#define QC_N 40

struct QLayout {
    size_t tab_off, tree_off, per_tree, x_off, h_off, acc_off, parts_off, obs_off, wh_off, cnt_off, total;
};
__host__ __device__ inline QLayout q_layout(int tab_n, int n_sims, int R, bool cont) {
    QLayout L;
    L.tab_off = 0;
    L.tree_off = ((size_t)tab_n * 8 + (size_t)(n_sims + 2) * 2 + 15) / 16 * 16;
    L.per_tree = ((size_t)R * 16 + (cont ? (size_t)POOL_UNITS(R) * 4 : (size_t)R * 4) + 15) / 16 * 16;
    L.x_off = L.tree_off + 16 * L.per_tree;
    L.h_off = L.x_off + 4 * 4 * Q_HP * 4;          // X [4 quarters][4 trees][256 positions] float
    L.acc_off = L.h_off + 4 * 4 * Q_HP * 4;        // H likewise
    L.parts_off = L.acc_off + 4 * 4 * 64 * 16;     // ACC [4 pairs][4 quarters][64 lanes] float4
    L.obs_off = L.parts_off + 4 * 8 * 4 * 16;      // PARTS [4 quarters][8 chunks][4 trees] float4 (outputs 0..3)
    L.wh_off = L.obs_off + 4 * 16 * 4;             // OBS [4 quarters][4 features][4 trees] float
    L.cnt_off = L.wh_off + 32 * 64 * 4;            // WH [32 steps][64 lanes] float: the heads' A operand
    L.total = L.cnt_off + QC_N * 4;
    return L;
}

__device__ __forceinline__ bool q_wait(unsigned* cnt, int idx, unsigned target) {
    unsigned spins = 0;
    while (__hip_atomic_load(cnt + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 1023u) == 0u) {
            if (spins > Q_SPIN_LIMIT) __hip_atomic_store(cnt + QC_ABORT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (__hip_atomic_load(cnt + QC_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) return false;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return true;
}
__device__ __forceinline__ void q_signal(unsigned* cnt, int idx, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // this wave's LDS writes are done before the counter moves
    if (lane == 0) __hip_atomic_fetch_add(cnt + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ f32x4 mfma44(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }

template <int ENV>
__global__ __launch_bounds__(64 * Q_WAVES, 1) void search_kernel_q(KParams P) {
    constexpr bool CONT = (ENV != AZG_ENV_CARTPOLE);
    typedef TreeStore<TS_LDS8>::Rec Rec;
    extern __shared__ double s_dyn[];
    const QLayout L = q_layout(P.tab_n, P.n_sims, P.R, CONT);
    char* lds = (char*)s_dyn;
    double* s_sqrt = (double*)(lds + L.tab_off);
    unsigned short* s_pw = (unsigned short*)(s_sqrt + P.tab_n);
    float* s_X = (float*)(lds + L.x_off);
    float* s_H = (float*)(lds + L.h_off);
    f32x4* s_ACC = (f32x4*)(lds + L.acc_off);
    f32x4* s_PARTS = (f32x4*)(lds + L.parts_off);
    float* s_OBS = (float*)(lds + L.obs_off);
    float* s_WH = (float*)(lds + L.wh_off);
    unsigned* s_cnt = (unsigned*)(lds + L.cnt_off);
    __shared__ float s_bhead[16];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int role = wave >> 2, w = wave & 3;        // 0: A (stage 1), 1: B (stage 2), 2: T (trees of quarter w)
    // start-up (the only workgroup barriers)
    for (int i = tid; i < P.tab_n; i += 64 * Q_WAVES) s_sqrt[i] = P.sqrt_tab[i];
    if (CONT) for (int i = tid; i < P.n_sims + 2; i += 64 * Q_WAVES) s_pw[i] = (unsigned short)(P.pw_need[i] < 65535 ? P.pw_need[i] : 65535);
    for (int i = tid; i < 32 * 64; i += 64 * Q_WAVES) s_WH[i] = P.q_wh[i];
    for (int i = tid; i < QC_N; i += 64 * Q_WAVES) s_cnt[i] = 0u;
    for (int i = tid; i < 4 * 16; i += 64 * Q_WAVES) s_OBS[i] = 0.0f;
    if (tid < 16) s_bhead[tid] = P.bhead[tid];
    __syncthreads();

    if (role == 2) {
        // ================================================================ T: the trees of quarter w
        const int q = w, tj = lane >> 4, sub = lane & 15;
        const int tl = 4 * q + tj;                               // tree within the workgroup
        const int tree = blockIdx.x * 16 + tl;
        const bool live = tree < P.B;
        const unsigned gtree = (unsigned)(P.tree_base + tree);
        const size_t tb = (size_t)(live ? tree : 0) * P.R;
        Cold* cold = P.cold + tb;
        double* edge_W = P.edge_W + tb;
        float* action = P.action + tb;
        TreeStore<TS_LDS8> ts;
        char* base = lds + L.tree_off + L.per_tree * tl;
        ts.hot = (Rec*)base;
        ts.pool = (unsigned char*)(base + (size_t)P.R * 16);
        ts.prior = (float*)(base + (size_t)P.R * 16);
        float* obs = s_OBS + q * 16;                             // [4 features][4 trees]
        const f32x4* parts = s_PARTS + q * 8 * 4;                // [8 chunks][4 trees]
        TreeState st = {};
        tree_init_root<ENV, TS_LDS8, 4>(P, st, ts, cold, edge_W, action, tree, live, sub, tj, gtree, obs);
        q_signal(s_cnt, QC_OBS(q), lane);
#ifdef AZG_STAMPS
        unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
        for (int k = 0; k <= P.n_sims; ++k) {
            if (!q_wait(s_cnt, QC_PARTS(q), (unsigned)(k + 1))) return;
            if (live) tree_phase_a<ENV, TS_LDS8, false, 8, 4>(P, st, ts, cold, edge_W, action, tb, k - 1, sub, tj, gtree, parts, s_bhead, s_sqrt);
            st.need_eval = false;
            if (k < P.n_sims) {
                __threadfence_block();
                if (live) tree_phase_b<ENV, TS_LDS8, false, 4, unsigned short>(P, st, ts, cold, edge_W, action, tb, sub, tj, gtree, s_sqrt, s_pw, obs STAMP_ARG);
                q_signal(s_cnt, QC_OBS(q), lane);
            }
        }
        // the trees as the results kernels read them
        if (live) {
            if (sub == 0) P.n_rec[tree] = st.nrec;
            RecL* gh = P.hot + tb;
            for (int j = sub; j < st.nrec; j += 16) {
                Rec h = ts.hot[j];
                RecL o;
                o.Q = h.Q; o.edge_n = h.edge_n; o.node_n = h.node_n; o.parent = (short)h.parent; o.n_child = h.n_child;
                o.first = CONT ? 0 : h.first; o.flags = h.flags; o.pad = 0;
                gh[j] = o;
                if (CONT) {
                    for (int i = 0; i < (int)h.n_child; ++i) P.child[(tb + j) * P.Kp + i] = (unsigned short)ts.child_at(j, h, i, P.Kp);
                } else {
                    P.prior[tb + j] = ts.prior[j];
                }
            }
        }
        return;
    }

    // ==================================================================== A / B: the network
    const int stage = role;                                      // 0: positions 0..127, 1: positions 128..255
    const int tcol = lane & 3, blk = lane >> 2;                  // B-operand column (tree of the quarter), block
    // this wave's half of pair w's hidden-layer weights: wr[pp] = W1[unit of position 64 w + lane][unit of position 128 stage + pp]
    float wr[128];
    {
        const float* src = P.q_w1 + ((size_t)(w * 2 + stage) * 128) * 64 + lane;
#pragma unroll
        for (int pp = 0; pp < 128; ++pp) wr[pp] = src[(size_t)pp * 64];
    }
    // first layer, this wave's 32 positions [128 stage + 32 w, +32): row of lane l = position base + l % 32
    float w0[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) w0[kk] = P.q_w0[((stage * 4 + w) * 4 + kk) * 64 + lane];
    const f32x4 b0 = P.q_b0[(stage * 4 + w) * 64 + lane];
    const f32x4 b1 = P.q_b1[w * 64 + lane];                       // stage 1: the accumulators start from the bias
    for (int k = 0; k <= P.n_sims; ++k) {
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {
            float* Xq = s_X + q * 4 * Q_HP;
            float* Hq = s_H + q * 4 * Q_HP;
            // ---- first layer of the quarter's new leaves: K = obs_dim <= 4, one fma per feature from the bias
            if (!q_wait(s_cnt, QC_OBS(q), (unsigned)(k + 1))) return;
            {
                const float* o = s_OBS + q * 16;
                f32x4 a0 = b0;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) a0 = mfma44(w0[kk], o[kk * 4 + tcol], a0);
                const f32x4 h0 = act4<false>(P.act, a0);
                if (lane < 32) *(f32x4*)(Xq + tcol * Q_HP + 128 * stage + 32 * w + 4 * blk) = h0;
                q_signal(s_cnt, stage == 0 ? QC_XA(q) : QC_XB(q), lane);
            }
            // ---- hidden layer, this wave's half of the k range, 64 output units x the quarter's 4 trees
            f32x4 acc;
            if (stage == 0) {
                if (!q_wait(s_cnt, QC_XA(q), (unsigned)(4 * (k + 1)))) return;
                acc = b1;
            } else {
                if (!q_wait(s_cnt, QC_XB(q), (unsigned)(4 * (k + 1)))) return;
                if (!q_wait(s_cnt, QC_S1(w, q), (unsigned)(k + 1))) return;
                acc = s_ACC[(w * 4 + q) * 64 + lane];
            }
            {
                const f32x4* xb = (const f32x4*)(Xq + tcol * Q_HP + 128 * stage);
#pragma unroll
                for (int p4 = 0; p4 < 32; ++p4) {
                    const f32x4 x = xb[p4];
                    acc = mfma44(wr[4 * p4 + 0], x.x, acc);
                    acc = mfma44(wr[4 * p4 + 1], x.y, acc);
                    acc = mfma44(wr[4 * p4 + 2], x.z, acc);
                    acc = mfma44(wr[4 * p4 + 3], x.w, acc);
                }
            }
            if (stage == 0) {
                s_ACC[(w * 4 + q) * 64 + lane] = acc;
                q_signal(s_cnt, QC_S1(w, q), lane);
                continue;
            }
            // ---- stage 2: activation, the layer's output in [tree][position] order, then (pair q) the heads
            *(f32x4*)(Hq + tcol * Q_HP + 64 * w + 4 * blk) = act4<false>(P.act, acc);
            q_signal(s_cnt, QC_H(q), lane);
            if (w != q) continue;
            if (!q_wait(s_cnt, QC_H(q), (unsigned)(4 * (k + 1)))) return;
            {
                // 8 chunk chains of 32 positions from 0: block c (and its mirror c + 8) = chunk c % 8; rows = outputs 0..3
                const f32x4* hb = (const f32x4*)(Hq + tcol * Q_HP + 32 * (blk & 7));
                f32x4 ph = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int s4 = 0; s4 < 8; ++s4) {
                    const f32x4 x = hb[s4];
                    ph = mfma44(s_WH[(4 * s4 + 0) * 64 + lane], x.x, ph);
                    ph = mfma44(s_WH[(4 * s4 + 1) * 64 + lane], x.y, ph);
                    ph = mfma44(s_WH[(4 * s4 + 2) * 64 + lane], x.z, ph);
                    ph = mfma44(s_WH[(4 * s4 + 3) * 64 + lane], x.w, ph);
                }
                if (lane < 32) s_PARTS[(q * 8 + blk) * 4 + tcol] = ph;
                q_signal(s_cnt, QC_PARTS(q), lane);
            }
        }
    }
}
