// lockstep.cuh -- wide policy/value networks (hidden width >= 512; BASELINE config E: 4x1024): one simulation step = a few
// grid-wide launches instead of one persistent kernel, so that a network layer of ALL trees is spread over ALL CUs.
//   ls_tree_kernel     one workgroup per 16 trees: phase A (finish leaf, backup) + phase B (select, step, expand); trees and
//                      the per-tree state live in global memory between launches
//   ls_layer0_kernel   first layer for (tree group, 256-unit slice)
//   ls_hidden_tiled_kernel   one hidden->hidden layer as an LDS-tiled GEMM, 32 trees x 64 units per workgroup, both operands
//                      double-buffered through LDS, two workgroups per CU; the last layer also leaves the partial head sums
// The arithmetic (MFMA chains, chunked head sums) is the persistent kernel's, bit for bit.
#pragma once
#include <type_traits>

#include "records.h"
#include "env.cuh"
#include "mlp.cuh"
#include "tree.cuh"
#include "tree_phases.cuh"

// g_base = first tree group of the launch.  (Measured and removed, HISTORY.md: the earlier 16-tree x 256-unit weight-streaming layer
// kernel; the first layer in this kernel's tail, +3.6 % time; the batch cut into pipelines on several streams, +6 % / +40 %.)
template <int ENV, bool GMM, int NCH, int HP>
__global__ __launch_bounds__(256) void ls_tree_kernel(KParams P, LockStep L, int sim, int g_base) {
    constexpr bool CONT = EnvFamily<ENV>::CONT;
    extern __shared__ double s_dyn[];   // sqrt_tab [tab_n], pw_need [n_sims+2]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, sub = lane & 15;
    const int tl = wave * 4 + (lane >> 4);
    const int tg = g_base + blockIdx.x;
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
                                                     L.parts + (size_t)tg * NCH * 64, P.bhead, s_sqrt STAMP_ARG, s_pw);
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
}

template <int HP>
__global__ __launch_bounds__(256) void ls_layer0_kernel(KParams P, LockStep L, int g_base) {
    constexpr int NS = HP / 256;
    const int tg = g_base + blockIdx.x / NS, sl = blockIdx.x % NS;
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

// One hidden->hidden layer as an LDS-tiled GEMM: a workgroup owns TG tree groups x UT unit tiles.  Both operands stream through
// LDS in chunks of LS_KC k-blocks (16 k each), double buffered, so that every weight block is read from L2 once per 16*TG trees
// and every activation block once per 16*UT units, and the MFMAs are fed by ds_read_b128 only.  Each accumulator still runs
// over k in ascending order: same arithmetic as everywhere else.
// Buffers are padded to a multiple of 4 tree groups (the padding groups compute on whatever is there and are never read).
// Measured on config E (1024 trees, 4x1024): 16x256 streaming kernel 26.5 us per layer; 64x64 tiles, one workgroup per CU 25 us;
// 32x64 tiles, two workgroups per CU (one's barrier / LDS-refill bubbles under the other's MFMAs) 22 us; the register-only
// MFMA loop of the same length is 16.5 us per launch (tools/mfma_rate.py).
#ifndef LS_KC
#define LS_KC 4
#endif
#ifndef LS_PIPE
#define LS_PIPE 4        // operand staging schedule of the tiled layer kernel (see there)
#endif
#ifndef LS_LAYER_WD
#define LS_LAYER_WD 0    // the per-layer launches' tile: 1 = ls_tile_wd (weights straight into registers), 0 = ls_tile (measured: 15.93 against
                         // 15.75 ms per search at 1024 trees, 91.5 against 90.9 at 8192 -- a kernel per layer has no hand-off waits to shorten)
#endif
#ifndef LS_LAYER_KC
#define LS_LAYER_KC 2    // its chunk length (k-blocks)
#endif
#ifndef LS_XCD_2D
#define LS_XCD_2D 1      // XCD-rectangle block mapping (0: unit slices per XCD)
#endif
// Tile shape: TG tree groups x UT unit tiles per workgroup.  4 x 4 (one workgroup per CU at 1024 trees x 1024 units) or half
// of that -- 4 x 2 or 2 x 4: twice the workgroups, two of them resident per CU, so that one's barrier / LDS-refill bubbles are
// covered by the other's MFMAs.  The last layer keeps UT = 4: its 64 units are one head chunk (with TG = 2 the chunk's chain
// passes from the wave that owns tiles 0-1 to the one that owns tiles 2-3 through LDS).
// Memory access of the tile routine.  Launched as a kernel per layer (SC1 = false) it uses plain loads and stores: the kernel
// boundary makes the previous layer's output visible.  Inside the persistent team kernel (SC1 = true) the activations, the
// observations and the head partials are handed from workgroup to workgroup WITHIN the launch: every such byte is stored and
// loaded with the sc1 bit (write-through past the L2, loads around the per-CU vector L1, which another CU's stores never
// refresh), as buffer instructions so that the compiler keeps counting them in vmcnt.
// (wt = false: the stores stay plain -- they still write through the L1 to the XCD's L2, where a reader on the SAME XCD finds
// them with its sc1 loads; the team kernel uses that once it has verified at run time that its team sits on one XCD.)
template <bool SC1>
struct TileMem {
    __amdgpu_buffer_rsrc_t r;
    const f32x4* p;
    bool wt;
    __device__ __forceinline__ explicit TileMem(const void* base, bool write_through = true) : p((const f32x4*)base), wt(write_through) {
        if constexpr (SC1) r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    }
    __device__ __forceinline__ f32x4 load4(size_t i) const {   // float4 element i
        if constexpr (SC1) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(i * 16), 0, 16);
            return __builtin_bit_cast(f32x4, v);
        } else return p[i];
    }
    __device__ __forceinline__ float load1(size_t i) const {   // float element i
        if constexpr (SC1) return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)(i * 4), 0, 16));
        else return ((const float*)p)[i];
    }
    __device__ __forceinline__ void store1(size_t i, float v) const {
        if constexpr (SC1) {
            if (wt) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)(i * 4), 0, 16);
            else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)(i * 4), 0, 0);
        } else ((float*)p)[i] = v;
    }
    __device__ __forceinline__ void store4(size_t i, f32x4 v) const {
        if constexpr (SC1) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            if (wt) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)(i * 16), 0, 16);
            else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)(i * 16), 0, 0);
        } else ((f32x4*)p)[i] = v;
    }
};

// One output tile of a hidden->hidden layer: TG tree groups (from g0) x UT unit tiles (slice us), by one 256-thread workgroup.
// KC_: k-blocks per staged chunk (LS_KC by default; the team kernel's three- and four-workgroups-per-CU forms use shorter chunks so
// that their stages fit the CU's LDS side by side).
// DBG (tools/probes/tile8 only; 0 in the product): 1 no global loads in the loop, 2 no staging stores / barriers, 4 no LDS operand reads
// (Round 5 measured requesting chunk 0's weights BEFORE the team kernel's wait for the layer's input, so that their L2 round trip runs
// under the wait: 13.38 against 13.23 ms per search at 1024 trees on one box, 24.17 against 24.25 ms at 2048 -- the registers that hold
// the requested weights through the wait cost more than the round trip; not kept.)
template <int HP, bool LAST, int TG, int UT, bool SC1, int KC_ = LS_KC, int DBG = 0>
__device__ __forceinline__ void ls_tile(const KParams& P, const LockStep& L, int layer, int in_buf, int us, int g0, f32x4* s_ab, bool wt = true) {
    static_assert(!LAST || UT == 4, "a head chunk is 4 tiles");
    static_assert(TG == 4 || TG == 2, "4 waves: one or two per tree group");
    constexpr int WPG = 4 / TG;            // waves per tree group
    constexpr int WT = UT / WPG;           // unit tiles per wave
    constexpr int S4 = HP / 16, NU = HP / (16 * UT), KC = KC_, NCHUNK = S4 / KC;
    constexpr int PIPE = LS_PIPE == 4 && KC < 4 ? 3 : LS_PIPE;   // (schedule 4 pays with long chunks only: see below)
    static_assert(S4 % KC == 0, "k-blocks per layer must be a multiple of the chunk");
    constexpr int ASZ = UT * KC * 64, BSZ = TG * KC * 64;   // float4 entries of a stage's A [UT tiles][KC][64] and B [TG groups][KC][64]
    constexpr int STAGE = ASZ + BSZ;
    constexpr int NLA = ASZ / 256, NLB = BSZ / 256;       // float4 loads per thread per chunk
    static_assert(ASZ % 256 == 0 && BSZ % 256 == 0, "chunk does not divide over the workgroup");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t0 = us * UT;                          // the UT output tiles
    const int wg = wave % TG, wt0 = (wave / TG) * WT;   // this wave: tree group g0 + wg, tiles t0 + wt0 .. + WT
    const f32x4* W = P.Wl[layer - 1];
    const TileMem<SC1> in(L.act[in_buf]), out(L.act[in_buf ^ 1], wt), parts(L.parts, wt);
    f32x4 ra[NLA], rb[NLB];
    // piece j of a chunk's staging: NLA float4 of the weights, then NLB of the activations, per thread
    auto load_one = [&](int c, int j) {
        const int jj = j < NLA ? j : j - NLA;
        const int e = jj * 256 + tid, i = e / (KC * 64), r = e % (KC * 64);   // tile / tree group, offset in its chunk
        if (j < NLA) ra[jj] = W[((size_t)(t0 + i) * S4 + c * KC) * 64 + r];
        else rb[jj] = in.load4(((size_t)(g0 + i) * S4 + c * KC) * 64 + r);
    };
    f32x4 dbg_const = {1.0f, 2.0f, 3.0f, 4.0f};
    if (DBG & 8) asm volatile("" : "+v"(dbg_const));
    auto store_one = [&](int st, int j) {
        if (DBG & 8) { s_ab[st * STAGE + j * 256 + tid] = dbg_const; return; }
        if (j < NLA) s_ab[st * STAGE + j * 256 + tid] = ra[j];
        else s_ab[st * STAGE + ASZ + (j - NLA) * 256 + tid] = rb[j - NLA];
    };
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int j = 0; j < NLA + NLB; ++j) load_one(c, j);
    };
    auto store_chunk = [&](int st) {
#pragma unroll
        for (int j = 0; j < NLA + NLB; ++j) store_one(st, j);
    };
    f32x4 acc[WT];   // bias first: store_chunk's wait for the chunk then covers it
#pragma unroll
    for (int i = 0; i < WT; ++i) acc[i] = P.bl[layer - 1][(t0 + wt0 + i) * 64 + lane];
    load_chunk(0);
    store_chunk(0);
    if (PIPE >= 2 && NCHUNK > 1) load_chunk(1);
    __syncthreads();
    // the bias has to have arrived before the loop: a wait for it inside the loop would, from the second pass on, wait for the
    // next chunk's loads instead (waitcnt placement is static)
#pragma unroll
    for (int i = 0; i < WT; ++i) asm volatile("" : "+v"(acc[i]));
    // Staging schedules (LS_PIPE):
    //   1  chunk c+1 requested at the top of iteration c, stored to LDS behind the iteration's MFMAs
    //   2  chunk c+1 (requested an iteration ago) stored at the top of iteration c, chunk c+2 requested right after
    //   3  like 2, but the stores and requests are dealt out one piece at a time BETWEEN the iteration's MFMA groups: they issue
    //      in the shadow of the MFMA in flight, so that a wave's stretch without matrix work per chunk shrinks to the barrier.
    //      (Profile of schedule 1: both workgroups of a CU run in phase, MFMAs together and staging together, and the matrix pipe
    //      idles for the whole staging stretch: 3.1k cycles per chunk for 2k cycles of MFMA.)
    constexpr int NL = NLA + NLB, NGRP = 4 * KC;          // staging pieces, MFMA groups per chunk
    constexpr int SLOT0 = 1, SLOTD = (NGRP - 2) / NL > 0 ? (NGRP - 2) / NL : 1;   // piece j goes behind MFMA group SLOT0 + j * SLOTD
    // one chunk: HAS1 / HAS2 = chunks c+1 / c+2 exist (compile-time, so that the loop body has no branches: the wait counts in
    // front of the stores then name exactly the loads they need, not "everything in flight")
    auto chunk = [&](int c, auto has1_t, auto has2_t) {
        constexpr bool has1 = decltype(has1_t)::value, has2 = decltype(has2_t)::value;
        if (PIPE == 2) {
            if (has1) store_chunk((c + 1) & 1);
            if (has2) load_chunk(c + 2);
        } else if (PIPE == 1 && has1) load_chunk(c + 1);       // in flight under this chunk's MFMAs
        const f32x4* sB = s_ab + (c & 1) * STAGE + ASZ + wg * KC * 64;
        const f32x4* sA = s_ab + (c & 1) * STAGE + wt0 * KC * 64;
        // operands of k-block s+1 are read from LDS while the MFMAs of k-block s run
        f32x4 a[WT], b, an[WT], bn;
        b = sB[lane];
#pragma unroll
        for (int i = 0; i < WT; ++i) a[i] = sA[(i * KC) * 64 + lane];
#pragma unroll
        for (int s = 0; s < KC; ++s) {
            if (s + 1 < KC && !(DBG & 4)) {
                bn = sB[(s + 1) * 64 + lane];
#pragma unroll
                for (int i = 0; i < WT; ++i) an[i] = sA[(i * KC + s + 1) * 64 + lane];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cmp = 0; cmp < 4; ++cmp) {
#pragma unroll
                for (int i = 0; i < WT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][cmp], b[cmp], acc[i], 0, 0, 0);
                if (PIPE == 3) {
                    const int q = 4 * s + cmp;
                    if (q >= SLOT0 && (q - SLOT0) % SLOTD == 0 && (q - SLOT0) / SLOTD < NL) {
                        const int j = (q - SLOT0) / SLOTD;
                        __builtin_amdgcn_sched_barrier(0);
                        if (has1 && !(DBG & 2)) store_one((c + 1) & 1, j);
                        if (has2 && !(DBG & 1)) load_one(c + 2, j);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (s + 1 < KC && !(DBG & 4)) {
                b = bn;
#pragma unroll
                for (int i = 0; i < WT; ++i) a[i] = an[i];
            }
        }
        if (PIPE == 1 && has1) store_chunk((c + 1) & 1);   // the other stage: its readers passed the previous barrier
        if (!(DBG & 2)) __syncthreads();
    };
    //   4  like 3, and the chunk's barrier stands in FRONT of its last k-block's MFMAs instead of behind them: that block's operands
    //      are in registers by then and every staging store of the chunk is issued (the pieces are dealt out over the first KC - 1
    //      k-blocks), so the first operands of the NEXT chunk are requested right behind the barrier and arrive under those MFMAs --
    //      the wave no longer starts every chunk with an LDS round trip that nothing covers (a workgroup that has its CU to itself:
    //      one wave per SIMD).  The barrier is a bare s_barrier behind lgkmcnt(0): the global loads in flight are not its business.
    //      Measured (MI355X, 4x1024 network, ms per search, schedule 3 -> 4): 512 trees (one workgroup per CU) 8.97 -> 8.71, 1024 trees
    //      13.08 -> 12.98, per-layer launches at 8192 trees 92.1 -> 90.0; with KC = 2 the pieces crowd into one k-block and it loses
    //      (1536 trees 18.64 -> 19.09, 3072 trees 32.63 -> 32.99): those forms keep schedule 3.
    constexpr int NS4 = 4 * (KC - 1), SLOTD4 = NS4 / NL > 0 ? NS4 / NL : 1, SLOT04 = NS4 > NL * SLOTD4 ? 1 : 0;
    static_assert(PIPE != 4 || NS4 >= NL, "schedule 4 deals the staging pieces out over the first KC - 1 k-blocks");
    f32x4 ca[WT], cb;   // schedule 4: operands of the k-block that runs next, carried from chunk to chunk
    auto chunk4 = [&](int c, auto has1_t, auto has2_t) {
        constexpr bool has1 = decltype(has1_t)::value, has2 = decltype(has2_t)::value;
        const f32x4* sB = s_ab + (c & 1) * STAGE + ASZ + wg * KC * 64;
        const f32x4* sA = s_ab + (c & 1) * STAGE + wt0 * KC * 64;
        f32x4 an[WT], bn;
#pragma unroll
        for (int s = 0; s < KC; ++s) {
            if (s + 1 < KC) {
                if (!(DBG & 4)) {
                    bn = sB[(s + 1) * 64 + lane];
#pragma unroll
                    for (int i = 0; i < WT; ++i) an[i] = sA[(i * KC + s + 1) * 64 + lane];
                }
            } else {
                if (!(DBG & 2)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (has1 && !(DBG & 4)) {
                    const f32x4* sBn = s_ab + ((c + 1) & 1) * STAGE + ASZ + wg * KC * 64;
                    const f32x4* sAn = s_ab + ((c + 1) & 1) * STAGE + wt0 * KC * 64;
                    bn = sBn[lane];
#pragma unroll
                    for (int i = 0; i < WT; ++i) an[i] = sAn[(i * KC) * 64 + lane];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cmp = 0; cmp < 4; ++cmp) {
#pragma unroll
                for (int i = 0; i < WT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[i][cmp], cb[cmp], acc[i], 0, 0, 0);
                const int q = 4 * s + cmp;
                if (s + 1 < KC && q >= SLOT04 && (q - SLOT04) % SLOTD4 == 0 && (q - SLOT04) / SLOTD4 < NL) {
                    const int j = (q - SLOT04) / SLOTD4;
                    __builtin_amdgcn_sched_barrier(0);
                    if (has1 && !(DBG & 2)) store_one((c + 1) & 1, j);
                    if (has2 && !(DBG & 1)) load_one(c + 2, j);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if ((s + 1 < KC || has1) && !(DBG & 4)) {
                cb = bn;
#pragma unroll
                for (int i = 0; i < WT; ++i) ca[i] = an[i];
            }
        }
    };
    static_assert(NCHUNK >= 2, "the staging pipeline is written for at least two chunks");
    (void)chunk; (void)chunk4;
    if constexpr (PIPE == 4) {
        cb = s_ab[ASZ + wg * KC * 64 + lane];
#pragma unroll
        for (int i = 0; i < WT; ++i) ca[i] = s_ab[(wt0 + i) * KC * 64 + lane];
#pragma unroll 1
        for (int c = 0; c < NCHUNK - 2; ++c) chunk4(c, std::true_type{}, std::true_type{});
        chunk4(NCHUNK - 2, std::true_type{}, std::false_type{});
        chunk4(NCHUNK - 1, std::false_type{}, std::false_type{});
    } else {
#pragma unroll 1
        for (int c = 0; c < NCHUNK - 2; ++c) chunk(c, std::true_type{}, std::true_type{});
        chunk(NCHUNK - 2, std::true_type{}, std::false_type{});
        chunk(NCHUNK - 1, std::false_type{}, std::false_type{});
    }
    if (DBG & 8) {
#pragma unroll
        for (int j = 0; j < NLA; ++j) asm volatile("" ::"v"(ra[j]));
#pragma unroll
        for (int j = 0; j < NLB; ++j) asm volatile("" ::"v"(rb[j]));
    }
    f32x4 h[WT];
#pragma unroll
    for (int i = 0; i < WT; ++i) h[i] = act4<true>(P.act, acc[i]);
    const int tg = g0 + wg;
    if constexpr (!LAST) {
#pragma unroll
        for (int i = 0; i < WT; ++i) out.store4(((size_t)tg * S4 + t0 + wt0 + i) * 64 + lane, h[i]);
    } else {
        // the slice's 64 units are one head chunk (chunk index = us): a chain from 0 over its 4 tiles, in tile order
        f32x4 hs = {0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (WPG == 2) {
            // tiles 0-1 live in waves 0..TG-1, tiles 2-3 in waves TG..3: the chain's running sum crosses through LDS
            // (all MFMA reads of the stages are behind the loop's last barrier)
            if (wt0 == 0) {
#pragma unroll
                for (int i = 0; i < WT; ++i) hs = mfma4(P.Whead[(t0 + i) * 64 + lane], h[i], hs);
                s_ab[wg * 64 + lane] = hs;
            }
            __syncthreads();
            if (wt0 != 0) {
                hs = s_ab[wg * 64 + lane];
#pragma unroll
                for (int i = 0; i < WT; ++i) hs = mfma4(P.Whead[(t0 + wt0 + i) * 64 + lane], h[i], hs);
                parts.store4(((size_t)tg * NU + us) * 64 + lane, hs);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) hs = mfma4(P.Whead[(t0 + i) * 64 + lane], h[i], hs);
            parts.store4(((size_t)tg * NU + us) * 64 + lane, hs);
        }
    }
}

// The same tile (TG tree groups x 4 unit tiles, one 256-thread workgroup) with the WEIGHTS NOT STAGED: wave w owns unit tile w for all
// TG tree groups, so a weight block is the A operand of exactly one wave -- it goes from global memory (already in the MFMA's lane
// layout) straight into that wave's registers, two chunks ahead, and only the activations, which all four waves share, pass through
// LDS.  Against ls_tile: the same global loads per thread, a third (TG = 2) or half (TG = 4) of the staging stores, one LDS operand
// read fewer per k-block, stages a third / half the size (LDS left for longer chunks: fewer barriers), and the weights' round trip has
// two chunks to complete instead of one.  Same arithmetic: every accumulator is the same k-ordered chain from the bias.
// Schedule: ls_tile's number 4 (barrier in front of the chunk's last k-block, the next chunk's first operands behind it).
// s_b: two stages of TG * KC * 64 float4.
template <int HP, bool LAST, int TG, bool SC1, int KC>
__device__ __forceinline__ void ls_tile_wd(const KParams& P, const LockStep& L, int layer, int in_buf, int us, int g0, f32x4* s_b, bool wt = true) {
    constexpr int S4 = HP / 16, NU = HP / 64, NCHUNK = S4 / KC;
    constexpr int BSZ = TG * KC * 64;                 // float4 entries of a stage: [TG groups][KC][64]
    constexpr int NLB = BSZ / 256;                    // float4 of the activations per thread per chunk
    static_assert(S4 % KC == 0 && BSZ % 256 == 0 && NCHUNK >= 4 && NCHUNK % 2 == 0, "chunking");
    static_assert(KC >= 2 && 4 * (KC - 1) >= NLB, "the staging pieces are dealt out over the first KC - 1 k-blocks");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = us * 4 + wave;                   // this wave's unit tile
    const f32x4* W = P.Wl[layer - 1] + (size_t)tile * S4 * 64 + lane;   // k-block kb of it: W[kb * 64]
    // (timing experiment, round 5: the activations through the L2 with plain loads -- stale data, wrong results -- 13.68 against 12.78 ms:
    // they push the slices' weights out of the XCD's L2; the sc1 loads around it are the faster way as well as the coherent one)
    const TileMem<SC1> in(L.act[in_buf]), out(L.act[in_buf ^ 1], wt), parts(L.parts, wt);
    f32x4 aw[2][KC];                                  // the weights of two chunks (set = chunk index & 1)
    f32x4 rb[NLB];
    auto load_b = [&](int c, int j) {
        const int e = j * 256 + tid, i = e / (KC * 64), r = e % (KC * 64);
        rb[j] = in.load4(((size_t)(g0 + i) * S4 + c * KC) * 64 + r);
    };
    auto store_b = [&](int st, int j) { s_b[st * BSZ + j * 256 + tid] = rb[j]; };
    f32x4 acc[TG];
#pragma unroll
    for (int g = 0; g < TG; ++g) acc[g] = P.bl[layer - 1][tile * 64 + lane];
    f32x4 whead = {0.0f, 0.0f, 0.0f, 0.0f};          // (last layer: this tile's head weights, requested here -- the head chain runs from
    if constexpr (LAST) whead = P.Whead[tile * 64 + lane];   // wave to wave, a round trip to L2 in each link would be four in a row)
#pragma unroll
    for (int s = 0; s < KC; ++s) aw[0][s] = W[s * 64];
#pragma unroll
    for (int j = 0; j < NLB; ++j) load_b(0, j);
#pragma unroll
    for (int j = 0; j < NLB; ++j) store_b(0, j);
#pragma unroll
    for (int s = 0; s < KC; ++s) aw[1][s] = W[(KC + s) * 64];
#pragma unroll
    for (int j = 0; j < NLB; ++j) load_b(1, j);
    __syncthreads();
#pragma unroll
    for (int g = 0; g < TG; ++g) asm volatile("" : "+v"(acc[g]));   // (the bias has arrived: see ls_tile)
    f32x4 b[TG], bn[TG];                              // operands of the k-block that runs next, carried from chunk to chunk
#pragma unroll
    for (int g = 0; g < TG; ++g) b[g] = s_b[(g * KC) * 64 + lane];
    constexpr int NS = 4 * (KC - 1), SLOTD = NS / NLB > 0 ? NS / NLB : 1, SLOT0 = NS > NLB * SLOTD ? 1 : 0;
    auto chunk = [&](int c, auto set_t, auto has1_t, auto has2_t) {
        constexpr int SET = decltype(set_t)::value;
        constexpr bool has1 = decltype(has1_t)::value, has2 = decltype(has2_t)::value;
        const f32x4* sB = s_b + SET * BSZ;
#pragma unroll
        for (int s = 0; s < KC; ++s) {
            if (s + 1 < KC) {
#pragma unroll
                for (int g = 0; g < TG; ++g) bn[g] = sB[(g * KC + s + 1) * 64 + lane];
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (has1) {
                    const f32x4* sBn = s_b + (SET ^ 1) * BSZ;
#pragma unroll
                    for (int g = 0; g < TG; ++g) bn[g] = sBn[(g * KC) * 64 + lane];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            const f32x4 a = aw[SET][s];
#pragma unroll
            for (int cmp = 0; cmp < 4; ++cmp) {
#pragma unroll
                for (int g = 0; g < TG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cmp], b[g][cmp], acc[g], 0, 0, 0);
                const int q = 4 * s + cmp;
                if (s + 1 < KC && q >= SLOT0 && (q - SLOT0) % SLOTD == 0 && (q - SLOT0) / SLOTD < NLB) {
                    const int j = (q - SLOT0) / SLOTD;
                    __builtin_amdgcn_sched_barrier(0);
                    if (has1) store_b(SET ^ 1, j);
                    if (has2) load_b(c + 2, j);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (has2) {   // this k-block's weights of the chunk after next, into the registers that just fed the matrix pipe
                __builtin_amdgcn_sched_barrier(0);
                aw[SET][s] = W[((c + 2) * KC + s) * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
            if (s + 1 < KC || has1) {
#pragma unroll
                for (int g = 0; g < TG; ++g) b[g] = bn[g];
            }
        }
    };
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
#pragma unroll 1
    for (int c = 0; c < NCHUNK - 2; c += 2) {
        chunk(c, S0{}, std::true_type{}, std::true_type{});
        chunk(c + 1, S1{}, std::true_type{}, std::true_type{});
    }
    chunk(NCHUNK - 2, S0{}, std::true_type{}, std::false_type{});
    chunk(NCHUNK - 1, S1{}, std::false_type{}, std::false_type{});
    f32x4 h[TG];
#pragma unroll
    for (int g = 0; g < TG; ++g) h[g] = act4<true>(P.act, acc[g]);
    if constexpr (!LAST) {
#pragma unroll
        for (int g = 0; g < TG; ++g) out.store4(((size_t)(g0 + g) * S4 + tile) * 64 + lane, h[g]);
    } else {
        // the slice's 64 units are one head chunk (chunk index = us): a chain from 0 over its 4 tiles, in tile order -- here from
        // wave to wave through LDS (every read of the stages is behind the loop's last barrier)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (wave == w) {
#pragma unroll
                for (int g = 0; g < TG; ++g) {
                    f32x4 hs = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (w > 0) hs = s_b[g * 64 + lane];
                    hs = mfma4(whead, h[g], hs);
                    if (w < 3) s_b[g * 64 + lane] = hs;
                    else parts.store4(((size_t)(g0 + g) * NU + us) * 64 + lane, hs);
                }
            }
            if (w < 3) __syncthreads();
        }
    }
}

// A hidden->hidden layer as a launch of its own: one tile per workgroup.
template <int HP, bool LAST, int TG, int UT>
__global__ __launch_bounds__(256) void ls_hidden_tiled_kernel(KParams P, LockStep L, int layer, int in_buf, int TQ, int g_base) {
    constexpr int NU = HP / (16 * UT);
    extern __shared__ f32x4 s_ab[];                        // two stages
    // Blocks of one XCD (blockIdx % 8; placement is a speed matter only) work on one rectangle of the output: a quarter of the
    // unit slices x half of the tree-group pairs, so that the XCD's 4 MB L2 holds both the weights (NU/4 slices) and the
    // activations (TQ/2 pairs) its blocks share -- 3 MB at 1024 trees x 1024 units; with the earlier mapping (all trees x two
    // slices per XCD: 4.5 MB) the activations kept falling out to the Infinity Cache.
    const int nb = TQ * NU;
    int us, tq;
    if (LS_XCD_2D && NU % 4 == 0 && TQ % 2 == 0) {
        const int x = blockIdx.x % 8, j = blockIdx.x / 8;          // XCD, index within the XCD's share (nb / 8 blocks)
        const int ub = NU / 4, tb = TQ / 2;                          // rectangle: ub slices x tb pairs
        us = (x % 4) * ub + j % ub;                                  // neighbours in time share an activation block (j / ub)
        tq = (x / 4) * tb + j / ub;
    } else {
        int m = blockIdx.x;
        if (nb % 8 == 0) m = (blockIdx.x % 8) * (nb / 8) + blockIdx.x / 8;
        us = m / TQ; tq = m % TQ;
    }
    if constexpr (LS_LAYER_WD && UT == 4) ls_tile_wd<HP, LAST, TG, false, LS_LAYER_KC>(P, L, layer, in_buf, us, g_base + tq * TG, s_ab);
    else ls_tile<HP, LAST, TG, UT, false>(P, L, layer, in_buf, us, g_base + tq * TG, s_ab);
}
