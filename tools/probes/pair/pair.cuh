// pair.cuh -- EXPERIMENT (off by default, AZG_PAIR=1|2; parity-tested like every other form).  Batches with two or more
// 16-tree groups per CU: the search as a WALKER kernel and a SERVER kernel that run side by side on every CU for the whole search.
//
// The persistent search kernel (search_kernel.cuh) runs network phase and tree phases of a workgroup's trees one after the
// other: with one group per CU nothing else is there to run, but with two the matrix pipe still idles through every tree
// phase, because a layer's weights are spread over all waves of the workgroup (every network phase needs every wave) and the
// two groups cannot be run out of phase.  Here the roles are split between two kernels with different register budgets that
// share a CU (4 + 4 waves, one of each per SIMD: 168 + 336 of the SIMD's 512 registers per lane):
//   server  pair_server_kernel : the hidden->hidden weights in its registers for the whole search (as in search_kernel), it
//                                evaluates one 16-tree group after the other (mlp_forward, the same arithmetic);
//   walker  pair_walker_kernel : 32 trees (two groups) resident in its LDS; every wave walks 4 trees of group A while the
//                                server evaluates group B's leaves, then swaps.
// Per step and group they exchange 64 floats (observations, walker -> server) and the head partials (server -> walker)
// through global memory, one monotonic counter per direction and group.  The walker has no workgroup barrier in its loop: each
// wave hands in the observations of its own 4 trees and polls for its own results (the server waits for 4 arrivals), so a
// wave with short traces starts on the other group while its neighbours are still walking.
//
// Pairing and visibility: see pair_slot() and PairLink below (partners are matched per XCD, hand-offs live in that XCD's L2).
// Residency: both kernels must be running at the same time (two streams); every wait is bounded, a time-out raises the abort
// word and the host reruns the search with the one-kernel form (azg_engine.hip: team_check).
//
// Measured (MI355X, 8192 trees x 200 simulations, 2x256 ELU; tools/time_pair.py, tools/pair_profile.py): bit-identical
// results, 3.15 ms per search against 2.67 ms for search_kernel<..., 8, 2>.  Per half-step (one group's evaluation + walk) the
// server is busy 14.4k cycles (network 13.7k; 12.1k when it has the SIMDs to itself) and the walker 15.4k (finish leaf +
// backup 3.5k, select / step / expand 10.6k -- 1.9k and 5.4k without a network running beside it): next to a wave that issues
// MFMAs back to back every vector instruction of another wave costs 8 cycles more (tools/probes/valu_rate.hip; s_setprio does
// not change it), about 650 of them per walk, and what the overlap wins that tax takes back.  With the two
// workgroups of a pair on different XCDs (sc1 hand-offs, about 7k cycles each) it was 3.7 ms.
#pragma once
#include "records.h"
#include "env.cuh"
#include "mlp.cuh"
#include "tree.cuh"
#include "tree_phases.cuh"
#include "search_kernel.cuh"
#include "lockstep.cuh"
#include "team.cuh"   // (TSTAMP / TADD)

#define PAIR_CNT_STRIDE 32      // counters 128 bytes apart
#define PAIR_CNT_PER 4          // per pair: 0, 1: observations of group g handed in (4 wave arrivals per step); 2, 3: results of group g

struct PairCtl {
    float* obs;            // [pairs][2][64]           observations, [feature][tree of the group]
    f32x4* parts;          // [pairs][2][NCH * PSTR]   head partials of a group's evaluation
    unsigned* cnt;         // [pairs][PAIR_CNT_PER][PAIR_CNT_STRIDE]
    unsigned* ticket;      // [2 roles][8 XCDs]: workgroups of a role that have started on an XCD
    unsigned* abort;       // != 0: a wait timed out (or the workgroups are not spread evenly over the XCDs), everybody leaves
    unsigned spin_limit;   // polls a wait may take
    int n_pairs;
    int per_xcd;           // gridDim.x / 8: workgroups of each kernel that every XCD receives
};

// Who is my partner?  The two kernels are dispatched independently, so workgroup i of one need not share an XCD (let alone a CU)
// with workgroup i of the other.  Each workgroup therefore draws a ticket on its own XCD: the n-th walker and the n-th server
// that start on XCD x form slot x * per_xcd + n.  Both grids are the same multiple of 8 and the dispatcher deals workgroups
// round-robin over the 8 XCDs, so every XCD receives per_xcd of each and the tickets match up exactly; should an XCD ever get
// more, the surplus workgroup raises the abort word (host: fall back to the one-kernel form).  Returns the slot, or -1.
__device__ __forceinline__ int pair_slot(int role, const PairCtl& T, volatile int* s_slot) {
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 7u;
        const unsigned n = __hip_atomic_fetch_add(T.ticket + role * 8 + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int slot = (int)(xcc * (unsigned)T.per_xcd + n);
        if (n >= (unsigned)T.per_xcd) { __hip_atomic_store(T.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); slot = -1; }
        *s_slot = slot;
    }
    __syncthreads();
    const int slot = *s_slot;
    __syncthreads();   // (the caller may reuse the word)
    return slot;
}

// One side's view of a pair's hand-off memory.  Both workgroups sit on the same XCD (pair_slot), so that XCD's L2 is their
// point of coherence: stores stay plain (they write through the CU's L1 into the L2; s_waitcnt vmcnt(0) = the L2 has them),
// loads and polls go around the L1 only (sc0), counters are L2 atomics.  (With sc1 / agent-scope operations, as in team.cuh, a
// hand-off costs about 7k cycles instead of about 1k: measured with the two workgroups on different XCDs.)
struct PairLink {
    __amdgpu_buffer_rsrc_t r_obs, r_parts;
    unsigned* cnt;
    __device__ __forceinline__ PairLink(const PairCtl& T, int pp, int per) : cnt(T.cnt + (size_t)pp * PAIR_CNT_PER * PAIR_CNT_STRIDE) {
        r_obs = __builtin_amdgcn_make_buffer_rsrc((void*)(T.obs + (size_t)pp * 128), 0, 0x7fffffff, 0x00020000);
        r_parts = __builtin_amdgcn_make_buffer_rsrc((void*)(T.parts + (size_t)pp * 2 * per), 0, 0x7fffffff, 0x00020000);
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __device__ __forceinline__ float load_obs(int i) const { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_obs, i * 4, 0, 16)); }
    __device__ __forceinline__ void store_obs(int i, float v) const { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r_obs, i * 4, 0, 0); }
    __device__ __forceinline__ f32x4 load_parts(int i) const { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_parts, i * 16, 0, 16)); }
    __device__ __forceinline__ void store_parts(int i, f32x4 v) const { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r_parts, i * 16, 0, 0); }
    // (one lane) the counter as the other side left it in the L2
    __device__ __forceinline__ unsigned peek(const unsigned* c) const {
        unsigned v;
        asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(c) : "memory");
        return v;
    }
    __device__ __forceinline__ void bump(unsigned* c) const { __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    // one lane polls, the wave follows; false: aborted
    __device__ __forceinline__ bool wait_wave(const unsigned* c, unsigned target, const PairCtl& T) const {
        int ok = 1;
        if ((threadIdx.x & 63) == 0) {
            unsigned spins = 0;
            if (T.spin_limit == 0u) {   // (tests: every wait counts as timed out)
                __hip_atomic_store(T.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
            } else while (peek(c) < target) {
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 63u) == 0u) {
                    if (spins > T.spin_limit) __hip_atomic_store(T.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__hip_atomic_load(T.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
                }
            }
        }
        return __builtin_amdgcn_readfirstlane(ok) != 0;
    }
    // workgroup-wide form (the server)
    __device__ __forceinline__ bool wait_wg(const unsigned* c, unsigned target, const PairCtl& T, volatile int* s_ok) const {
        if (threadIdx.x < 64) { const bool ok = wait_wave(c, target, T); if (threadIdx.x == 0) *s_ok = ok; }
        __syncthreads();
        const bool ok = *s_ok != 0;
        __syncthreads();
        return ok;
    }
};

// dynamic LDS of the walker: sqrt_tab, pw_need, then 32 trees
__host__ __device__ inline size_t pair_tree_off(int tab_n, int n_sims) { return ((size_t)tab_n * 8 + (size_t)(n_sims + 2) * 2 + 15) / 16 * 16; }
__host__ __device__ inline size_t pair_tree_bytes(int R, bool cont, int tlds) {
    size_t per = (size_t)R * 16 + (cont ? (size_t)POOL_UNITS(R) * (tlds == TS_LDS9 ? 8 : 4) : (size_t)R * 4);
    return (per + 15) / 16 * 16;
}

#ifndef PAIR_WALKER_WAVES
#define PAIR_WALKER_WAVES 3
#endif

template <int ENV, int HP, int TLDS, bool GMM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PAIR_WALKER_WAVES, PAIR_WALKER_WAVES))) void pair_walker_kernel(KParams P, PairCtl T) {
    constexpr bool CONT = (ENV != AZG_ENV_CARTPOLE);
    typedef typename TreeStore<TLDS>::Rec Rec;
    constexpr int NCH = head_chunks<HP>();
    constexpr int PSTR = GMM ? 64 : 16;
    constexpr int PER = NCH * PSTR;                 // float4 entries of a group's head partials
    __shared__ f32x4 s_parts[2 * PER];
    __shared__ float s_obsT[2 * 64];
    __shared__ float s_bhead[16];
    extern __shared__ double s_dyn[];               // sqrt_tab [tab_n], pw_need [n_sims + 2] u16, the trees' hot records

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, sub = lane & 15;
    const int slot = wave * 4 + (lane >> 4);        // the lane's tree within a group
    double* s_sqrt = s_dyn;
    unsigned short* s_pw = (unsigned short*)(s_dyn + P.tab_n);
    for (int i = tid; i < P.tab_n; i += 256) s_sqrt[i] = P.sqrt_tab[i];
    if (CONT) for (int i = tid; i < P.n_sims + 2; i += 256) s_pw[i] = (unsigned short)(P.pw_need[i] < 65535 ? P.pw_need[i] : 65535);
    if (tid < 16) s_bhead[tid] = P.bhead[tid];
    __syncthreads();
    const size_t per_tree = pair_tree_bytes(P.R, CONT, TLDS);
    char* tree_base = (char*)s_dyn + pair_tree_off(P.tab_n, P.n_sims);

#ifdef AZG_STAMPS
    // diagnostic build: cycles of this wave in  0 wait for results | 1 fetch them | 2 finish leaf + backup | 3 select / step / expand |
    // 4 hand in | 5 whole loop;  the server (row of wave 0, slots 8..): 8 wait for observations | 9 fetch + network | 10 store + arrive | 11 whole loop
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // (the tree phases' own stamps: discarded here)
#endif
    __shared__ int s_slot;
    const int slot0 = pair_slot(0, T, &s_slot);
    if (slot0 < 0) return;
    for (int pp = slot0; pp < T.n_pairs; pp += gridDim.x) {
        PairLink link(T, pp, PER);
        unsigned* cnt = link.cnt;
        // per group: tree, its arrays, its state.  `a` = the group whose results are awaited next, `b` = the other one
        struct Grp {
            TreeState st;
            TreeStore<TLDS> ts;
            Cold* cold; double* edge_W; float* action;
            size_t tb; int tree; bool live; unsigned gtree; int g;
        } a, b;
        auto setup = [&](Grp& x, int g) {
            x.g = g;
            x.tree = pp * 32 + g * 16 + slot;
            x.live = x.tree < P.B;
            x.gtree = (unsigned)(P.tree_base + x.tree);
            x.tb = (size_t)(x.live ? x.tree : 0) * P.R;
            x.cold = P.cold + x.tb; x.edge_W = P.edge_W + x.tb; x.action = P.action + x.tb;
            if constexpr (TLDS != TS_GLOBAL) {
                char* base = tree_base + per_tree * (g * 16 + slot);
                x.ts.hot = (Rec*)base;
                x.ts.pool = (typename TreeStore<TLDS>::PoolId*)(base + (size_t)P.R * 16);
                x.ts.prior = (float*)(base + (size_t)P.R * 16);
            } else {
                x.ts.hot = (Rec*)(P.hot + x.tb);
                x.ts.child = P.child + x.tb * P.Kp;
                x.ts.prior = P.prior + x.tb;
            }
            x.st = TreeState{};
            tree_init_root<ENV, TLDS, 16>(P, x.st, x.ts, x.cold, x.edge_W, x.action, x.tree, x.live, sub, slot, x.gtree, s_obsT + g * 64);
        };
        // this wave's 16 observation values of group g (features 0..3 of its trees 4 wave .. 4 wave + 3) -> the server
        auto hand_in = [&](int g) {
            if (lane < 16) link.store_obs(g * 64 + (lane >> 2) * 16 + wave * 4 + (lane & 3), s_obsT[g * 64 + (lane >> 2) * 16 + wave * 4 + (lane & 3)]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) link.bump(cnt + g * PAIR_CNT_STRIDE);
        };
        setup(a, 0);
        hand_in(0);
        setup(b, 1);
        hand_in(1);
        TSTAMP(t_begin);
        for (int h = 0; h < 2 * (P.n_sims + 1); ++h) {          // half-step: evaluation k of group a.g is due
            const int k = h >> 1, sim = k - 1;
            TSTAMP(t0);
            if (!link.wait_wave(cnt + (2 + a.g) * PAIR_CNT_STRIDE, (unsigned)(k + 1), T)) return;
            TSTAMP(t1);
            // the head partials of this wave's 4 trees: chunk c, output row group q, column slot -> s_parts[g][c * PSTR + q * 16 + slot]
            f32x4* my_parts = s_parts + a.g * PER;
            constexpr int NQ = PSTR / 16;
            for (int i = lane; i < NCH * NQ * 4; i += 64) {
                const int c = i / (NQ * 4), q = (i / 4) % NQ, col = wave * 4 + (i & 3);
                my_parts[c * PSTR + q * 16 + col] = link.load_parts(a.g * PER + c * PSTR + q * 16 + col);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            TSTAMP(t2);
            if (a.live) tree_phase_a<ENV, TLDS, GMM, NCH, PSTR>(P, a.st, a.ts, a.cold, a.edge_W, a.action, a.tb, sim, sub, slot, a.gtree, my_parts, s_bhead, s_sqrt);
            a.st.need_eval = false;
            TSTAMP(t3);
            TADD(0, t0, t1); TADD(1, t1, t2); TADD(2, t2, t3);
            if (k < P.n_sims) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (a.live) tree_phase_b<ENV, TLDS, GMM, 16, unsigned short>(P, a.st, a.ts, a.cold, a.edge_W, a.action, a.tb, sub, slot, a.gtree, s_sqrt, s_pw, s_obsT + a.g * 64 STAMP_ARG);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                TSTAMP(t4);
                hand_in(a.g);
                TSTAMP(t5);
                TADD(3, t3, t4); TADD(4, t4, t5);
            }
            // swap the roles, member by member (a whole-struct copy would also move the structs' padding bytes, which the
            // compiler keeps in scratch memory: 22 dependent scratch accesses, 3.5k cycles per half-step)
            {
                auto sw = [](auto& x, auto& y) { auto t = x; x = y; y = t; };
                sw(a.st.nrec, b.st.nrec); sw(a.st.eps_draws, b.st.eps_draws); sw(a.st.leaf, b.st.leaf); sw(a.st.need_eval, b.st.need_eval);
                sw(a.st.path_D, b.st.path_D); sw(a.st.my_depth, b.st.my_depth); sw(a.st.pid, b.st.pid); sw(a.st.pr, b.st.pr); sw(a.st.pW, b.st.pW);
                sw(a.st.kbase, b.st.kbase); sw(a.st.eps_c, b.st.eps_c); sw(a.st.ptop, b.st.ptop);
                sw(a.ts.hot, b.ts.hot); sw(a.ts.prior, b.ts.prior);
                if constexpr (TLDS != TS_GLOBAL) sw(a.ts.pool, b.ts.pool); else sw(a.ts.child, b.ts.child);
                sw(a.cold, b.cold); sw(a.edge_W, b.edge_W); sw(a.action, b.action); sw(a.tb, b.tb); sw(a.tree, b.tree); sw(a.live, b.live);
                sw(a.gtree, b.gtree); sw(a.g, b.g);
            }
#ifdef AZG_STAMPS
            { const unsigned long long t6 = __builtin_amdgcn_s_memtime(); tacc[6] += t6 - t0; tacc[7] += 1; }
#endif
        }
        TSTAMP(t_end);
        TADD(5, t_begin, t_end);
        // ---- the trees as the results kernels read them
        for (int gi = 0; gi < 2; ++gi) {
            const Grp& x = gi == 0 ? a : b;
            if (!x.live) continue;
            if (sub == 0) P.n_rec[x.tree] = x.st.nrec;
            if constexpr (TLDS != TS_GLOBAL) {
                RecL* gh = P.hot + x.tb;
                for (int j = sub; j < x.st.nrec; j += 16) {
                    Rec hh = x.ts.hot[j];
                    RecL o;
                    o.Q = hh.Q; o.edge_n = hh.edge_n; o.node_n = hh.node_n; o.parent = (short)hh.parent; o.n_child = hh.n_child;
                    o.first = CONT ? 0 : hh.first; o.flags = hh.flags; o.pad = 0;
                    gh[j] = o;
                    if (CONT) {
                        for (int i = 0; i < (int)hh.n_child; ++i) P.child[(x.tb + j) * P.Kp + i] = (unsigned short)x.ts.child_at(j, hh, i, P.Kp);
                    } else {
                        P.prior[x.tb + j] = x.ts.prior[j];
                    }
                }
            }
        }
    }
#ifdef AZG_STAMPS
    if (lane == 0) for (int i = 0; i < 8; ++i) P.stamps[((size_t)slot0 * 4 + wave) * 16 + i] = tacc[i];
#endif
}

// The server's weights: hidden->hidden matrices and the first layer in registers like WRegs<>; the biases (accumulator
// initial values: 4 distinct float4 per tile) and the head weights are re-read from LDS every step -- 48 registers per lane
// that the walker on the same SIMD needs more.
struct LdsBias {
    const f32x4* p;   // [tile of this wave][4 lane groups]
    int q;            // lane >> 4
    __device__ __forceinline__ f32x4 operator[](int i) const { return p[i * 4 + q]; }
};
// head weights (MFMA A operand: lane = output row lane & 15, k-group lane >> 4).  ROWS = 4: only output rows 0..3 exist
// (value + Normal / 2-action heads), the other lanes' operand is zero; ROWS = 16: all of them (mixture heads)
template <int ROWS>
struct LdsHead {
    const f32x4* p;   // [tile of this wave][4 k-groups][ROWS]
    int lane;
    __device__ __forceinline__ f32x4 operator[](int i) const {
        const int row = lane & 15;
        f32x4 v = p[i * 4 * ROWS + (lane >> 4) * ROWS + (ROWS == 16 ? row : (row & 3))];
        if (ROWS < 16 && row >= ROWS) v = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        return v;
    }
};
template <int HP, int NREG, int ROWS>
struct WRegsServer {
    static constexpr int NTW = HP / 64;
    static constexpr int S4 = HP / 16;
    f32x4 w[NREG][NTW][S4];
    float w0[NTW];
    LdsBias b0, b[NREG];
    LdsHead<ROWS> wh;
};

template <int HP, int NREG, bool GMM>
__global__ __launch_bounds__(256) void pair_server_kernel(KParams P, PairCtl T) {
    constexpr int NCH = head_chunks<HP>();
    constexpr int PSTR = GMM ? 64 : 16;
    constexpr int PER = NCH * PSTR;
    __shared__ f32x4 s_parts[PER];
    __shared__ float s_obsT[64];
    __shared__ float s_ln[1];
    __shared__ int s_ok;
    extern __shared__ f32x4 s_act[];                // activation buffers: act_buffers(NREG) x HP*64 bytes
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    f32x4* s_actA = s_act;
    f32x4* s_actB = act_buffers(NREG) == 2 ? s_actA + (HP / 16 * 64) : s_actA;

    constexpr int NTW = HP / 64, NT = HP / 16, S4 = HP / 16;
    static_assert(HP <= 256 && NREG > 0, "the server keeps all its weights in registers");
    __shared__ f32x4 s_bias[(1 + NREG) * NT * 4];   // [layer][tile][lane group]
    constexpr int ROWS = GMM ? 16 : 4;
    __shared__ f32x4 s_wh[NT * 4 * ROWS];           // head weights, [tile][k-group][output row]
    for (int i = tid; i < NT * 4; i += 256) {
        s_bias[i] = P.b0[(i / 4) * 64 + (i % 4) * 16];
#pragma unroll
        for (int l = 0; l < NREG; ++l) s_bias[(1 + l) * NT * 4 + i] = P.bl[l][(i / 4) * 64 + (i % 4) * 16];
    }
    for (int i = tid; i < NT * 4 * ROWS; i += 256) s_wh[i] = P.Whead[(i / (4 * ROWS)) * 64 + ((i / ROWS) % 4) * 16 + i % ROWS];
    WRegsServer<HP, NREG, ROWS> wr;
    wr.b0 = LdsBias{s_bias + wave * NTW * 4, lane >> 4};
    wr.wh = LdsHead<ROWS>{s_wh + wave * NTW * 4 * ROWS, lane};
#pragma unroll
    for (int i = 0; i < NTW; ++i) wr.w0[i] = P.W0[(wave * NTW + i) * 64 + lane];
#pragma unroll
    for (int l = 0; l < NREG; ++l) {
        wr.b[l] = LdsBias{s_bias + ((1 + l) * NT + wave * NTW) * 4, lane >> 4};
#pragma unroll
        for (int i = 0; i < NTW; ++i)
#pragma unroll
            for (int s4 = 0; s4 < S4; ++s4) wr.w[l][i][s4] = P.Wl[l][((wave * NTW + i) * S4 + s4) * 64 + lane];
    }
    __syncthreads();
#ifdef AZG_STAMPS
    unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tacc[4] = {0, 0, 0, 0};
#endif
    const int slot0 = pair_slot(1, T, &s_ok);
    if (slot0 < 0) return;
    for (int pp = slot0; pp < T.n_pairs; pp += gridDim.x) {
        PairLink link(T, pp, PER);
        unsigned* cnt = link.cnt;
        TSTAMP(t_begin);
        for (int h = 0; h < 2 * (P.n_sims + 1); ++h) {
            const int k = h >> 1, g = h & 1;
            TSTAMP(t0);
            if (!link.wait_wg(cnt + g * PAIR_CNT_STRIDE, (unsigned)(4 * (k + 1)), T, &s_ok)) return;
            TSTAMP(t1);
            if (tid < 64) s_obsT[tid] = link.load_obs(g * 64 + tid);
            __syncthreads();
#ifdef PAIR_X_NOMLP   // (experiment: the walker's speed without a network next to it -- results are wrong)
            if (tid < PER) s_parts[tid] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            __syncthreads();
#elif defined(AZG_STAMPS)
            mlp_forward<HP, NREG, 4, 1, PSTR, WRegsServer<HP, NREG, ROWS>>(P, wr, s_obsT, s_actA, s_actB, s_parts, s_ln, wave, lane, st_acc);
#else
            mlp_forward<HP, NREG, 4, 1, PSTR, WRegsServer<HP, NREG, ROWS>>(P, wr, s_obsT, s_actA, s_actB, s_parts, s_ln, wave, lane);
#endif
            TSTAMP(t2);
            for (int i = tid; i < PER; i += 256) link.store_parts(g * PER + i, s_parts[i]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) link.bump(cnt + (2 + g) * PAIR_CNT_STRIDE);
            TSTAMP(t3);
            TADD(0, t0, t1); TADD(1, t1, t2); TADD(2, t2, t3);
        }
        TSTAMP(t_end);
        TADD(3, t_begin, t_end);
    }
#ifdef AZG_STAMPS
    if (tid == 0) for (int i = 0; i < 4; ++i) P.stamps[(size_t)slot0 * 4 * 16 + 8 + i] = tacc[i];
#endif
}
