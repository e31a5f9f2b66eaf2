// team.cuh -- wide networks (hidden width >= 512), the whole search as ONE persistent launch of cooperating workgroups.
//
// The lock-step path (lockstep.cuh) pays five kernel boundaries per simulation step and runs its small, latency-bound tree
// kernel (64 workgroups) while the other 192 CUs wait.  Here the batch is cut into TEAMS: 32 trees (two 16-tree groups) and the
// NU = HP/64 workgroups that compute the 64-unit slices of every layer for exactly those trees.  A team never needs data of
// another team, so nothing in the launch is grid-wide.  Every workgroup of a team also OWNS 32/NU of the team's trees (2 at
// HP = 1024) for the whole search -- resident in its LDS when they fit, like the persistent search kernel's -- and walks them
// with the first lanes of its first wave.  Per simulation step a team goes
//     tree phases + first layer (every workgroup, its own trees)  ->  hidden layer 1 .. n (every workgroup, one tile each)
// and hands activations and head partials from workgroup to workgroup through global memory, one monotonic counter per
// hand-off (arrive = one atomic add per workgroup, wait = one lane polling).
//
// Visibility (MI355X: a CU's vector L1 is never refreshed by other CUs' stores, the XCD L2s are not coherent with each other):
// every handed-off byte is stored AND loaded with sc1 buffer instructions (TileMem<true>: write-through past the L2, loads
// around the L1); an arriving workgroup drains its stores (s_waitcnt vmcnt(0) in every wave), meets at its barrier, then one
// lane adds to the counter; a waiting workgroup polls with an sc1 load from one lane and releases the others through its
// barrier.  Placement (blockIdx % 8 = XCD under round-robin dispatch) is used for speed only: by default a team's slices are
// spread over the XCDs so that each XCD's L2 holds its slices' weights.  (In the other mapping, a team on one XCD, the team
// checks at run time that it really sits on one -- its workgroups' XCC_IDs -- and then keeps its hand-off stores plain: they
// write through the L1 into that XCD's L2, where the team's sc1 loads find them.)
// Deadlock: every workgroup of the grid has to be resident; the host checks the occupancy before choosing this path, and every
// wait is bounded -- on a time-out the launch raises an abort flag, all workgroups leave, and the host falls back.
// The arithmetic is the lock-step path's (ls_tile, tree_phase_a/b): bit-identical results.
//
// Measured at config E (1024 trees, 4x1024, MI355X): 13.3 ms per search against 14.9 ms for the per-layer launches (16.6 ms in
// round 1).  Per step (tools/team_profile.py, 153k cycles = 64 us): the three tiles 104k (two workgroups per CU side by side:
// 94 % of the matrix pipe's rate while they run), hand-off waits 21k (4 per step, across XCDs), tree phases + first layer
// 21k.  The two workgroups of a CU (blocks b and b + 256: tools/team_census.py) belong to different teams; shifting one team
// by half a step (re-measured in round 5 with the weights-direct tile: 12.80 / 12.82 ms with a 30k / 60k-cycle head start against 12.81)
// or gating it on its partner's progress did not pay (two tiles side by side take 35k cycles, one alone 24k:
// running them together is the efficient state), nor did making the first layer inside the first hidden layer's staging
// (+18k cycles per step) or as a team phase of MFMA tiles behind an observation hand-off (same time as the vector-ALU form
// here, one hand-off more).
#pragma once
#include "lockstep.cuh"

#define TEAM_CNT_STRIDE 32      // counters 128 bytes apart
#define TEAM_MAX_CNT 8          // counter 0: first layer of the step written; l: hidden layer l written
#define TEAM_CNT_XA (TEAM_MAX_CNT - 1)   // max XCC_ID of the team's workgroups
#define TEAM_CNT_XB (TEAM_MAX_CNT - 2)   // max (7 - XCC_ID)
#ifndef TEAM_SPREAD
#define TEAM_SPREAD 1           // 1: a team's unit slices spread over the XCDs (weights L2-resident); 0: a team on one XCD
#endif
#ifndef TEAM_SAME_XCD
#define TEAM_SAME_XCD 1         // plain hand-off stores (kept in the XCD's L2) once the team is seen to sit on one XCD
#endif
#ifndef TEAM_DEFER
#define TEAM_DEFER 1            // the new node's bookkeeping behind the workgroup's arrival at the hand-off (0: A/B builds)
#endif
#ifndef TEAM_LDS_COLD
#define TEAM_LDS_COLD 1         // the default form (32-tree teams, two workgroups per CU, trees of <= 255 records): the cold records, returns and actions in LDS
#endif                          // too (the weights-direct tile left the room), written out once at the end of the search
// bytes per tree of that: Cold[R] + edge_W[R] + action[R]
__host__ __device__ inline size_t team_cold_bytes(int R) { return ((size_t)R * (64 + 8 + 4) + 15) / 16 * 16; }
#ifndef TEAM_TILE_DBG
#define TEAM_TILE_DBG 0          // timing experiments only (lockstep.cuh: ls_tile's DBG): wrong results
#endif
#ifndef TEAM_WD
#define TEAM_WD 1                // 1: ls_tile_wd (weights straight into registers, only the activations staged); 0: ls_tile
#endif
#if TEAM_WD
#define LS_TILE_STAGE_F4(KC, TG) (2 * (TG) * (KC) * 64)          // float4 entries of the tile routine's two stages (activations only)
#else
#define LS_TILE_STAGE_F4(KC, TG) (2 * (4 + (TG)) * (KC) * 64)   // float4 entries of the tile routine's two stages (TG tree groups, UT = 4)
#endif

struct TeamCtl {
    unsigned* cnt;         // [teams][TEAM_MAX_CNT][TEAM_CNT_STRIDE]
    unsigned* abort;       // != 0: a wait timed out, everybody leaves
    unsigned spin_limit;   // polls (about 0.15 us each) a wait may take: 1 << 23 is more than a second
};

// dynamic LDS of a team workgroup: the tile stages, sqrt_tab, pw_need, then (LDS trees) the workgroup's trees
__host__ __device__ inline size_t team_tree_bytes(int R, bool cont, int tlds) {
    if (tlds == TS_GLOBAL) return 0;
    size_t per = (size_t)R * 16 + (cont ? (size_t)POOL_UNITS(R) * (tlds == TS_LDS9 ? 8 : 4) : (size_t)R * 4);
    return (per + 15) / 16 * 16;
}
// (the stages double as the head partials' landing area between tiles: NCH * 64 float4, at most 16 * 64)
__host__ __device__ inline size_t team_table_off(int kc, int tg = 2) { size_t f4 = LS_TILE_STAGE_F4(kc, tg); return (f4 < 1024 ? 1024 : f4) * 16; }
__host__ __device__ inline size_t team_tree_off(int tab_n, int n_sims, int kc, int tg = 2) {
    return (team_table_off(kc, tg) + (size_t)tab_n * 8 + (size_t)(n_sims + 2) * 4 + 15) / 16 * 16;
}

// all of this workgroup's hand-off stores are on their way: drain, meet, one lane arrives
__device__ __forceinline__ void team_arrive(unsigned* c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wait until the counter reaches `target`; false (uniform over the workgroup): aborted
__device__ __forceinline__ bool team_wait(const unsigned* c, unsigned target, const TeamCtl& T, volatile int* s_ok) {
    unsigned* abort = T.abort;
    if (threadIdx.x == 0) {
        int ok = 1;
        unsigned spins = 0;
        if (T.spin_limit == 0u) {   // (tests: every wait counts as timed out)
            __hip_atomic_store(abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = 0;
        } else {
            // (Round 5 measured two polls in flight, half a round trip apart, so that a poll issued just before the counter moves is
            // followed by one that sees it sooner: 13.20 against 13.15 ms per search at 1024 trees, 23.32 against 23.29 at 2048 -- a wait
            // ends when the slowest of the team's workgroups arrives, not when the poll notices; not kept.)
            while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(2);
                if ((++spins & 15u) == 0u) {
                    if (spins > T.spin_limit) __hip_atomic_store(abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
                }
            }
        }
        *s_ok = ok;
    }
    __syncthreads();
    const bool ok = *s_ok != 0;
    __syncthreads();   // everybody has read the verdict before the next wait overwrites it
    return ok;
}

#ifdef AZG_STAMPS
#define TSTAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define TADD(slot, a, b) tacc[slot] += (b) - (a)
#else
#define TSTAMP(v)
#define TADD(slot, a, b)
#endif

// KC: k-blocks per staged chunk of the tile routine; MINB: workgroups that have to fit a CU side by side (2: the default form,
// 1024 trees at HP = 1024; 3 and 4 (shorter chunks: smaller stages, fewer registers): larger batches -- while one workgroup of a
// CU waits at a hand-off or walks its trees, the others keep the matrix pipe busy).
// SPEC: compile-time knowledge about run-time parameters for the tree phases (tree_phases.cuh: Spec<>; 0 = the general code).
// TT: trees of a team, 32 or 64 (TGN = 2 or 4 tree groups: the tile is TT trees x 64 units; 64 halves the weight bytes staged per
// MFMA and the barriers per MFMA, and needs twice the batch for the same number of workgroups).
template <int ENV, int HP, bool GMM, int TLDS, int KC = LS_KC, int MINB = 2, int SPEC = 0, int TT = 32>
__global__ __launch_bounds__(256, MINB) void ls_team_kernel(KParams P, LockStep L, TeamCtl T, int TQ, int g_base) {
    constexpr bool CONT = EnvFamily<ENV>::CONT;
    constexpr int NU = HP / 64, NCH = HP / 64, TGN = TT / 16;
    constexpr int TPW = TT / NU;        // trees a workgroup owns: 16 lanes each, the first 16 * TPW lanes of wave 0
    static_assert(TT == 32 || TT == 64, "a team is two or four 16-tree groups");
    static_assert(TPW >= 1 && TPW <= 4, "a team is TT trees over HP/64 workgroups");
    typedef typename TreeStore<TLDS>::Rec Rec;
    extern __shared__ f32x4 s_ab[];     // the tile routine's two stages (between tiles: the head partials of this workgroup's
                                        // trees), sqrt_tab [tab_n], pw_need [n_sims + 2], (LDS trees) the trees
    __shared__ float s_obs[32];         // [4 features][TPW] observations of this workgroup's new leaves, zero padded to a line
    __shared__ int s_ok;
    double* s_sqrt = (double*)((char*)s_ab + team_table_off(KC, TGN));
    int* s_pw = (int*)(s_sqrt + P.tab_n);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, sub = lane & 15;
    // team and slice of this workgroup: the NU workgroups of a team have equal blockIdx % 8 (one XCD under round-robin placement)
    int tq, us;
#if TEAM_SPREAD
    // A team's slices are SPREAD over the XCDs (blockIdx % 8 = XCD under round-robin dispatch: NU / 8 slices on each), so that
    // every XCD's 4 MB L2 keeps just its own slices' weights resident (1.5 MB at 4x1024) instead of streaming all 12.6 MB of them
    // from the Infinity Cache every step; the hand-offs then cross XCDs (sc1 stores and loads).  Measured 13.1 ms per search
    // against 14.8 ms with each team on one XCD (plain hand-off stores into its L2; round 5, 64-tree teams: 23.3 / 24.2 ms at 2048 trees,
    // 32.8 / 33.4 at 3072): the weights matter more.
    if (NU % 8 == 0) { const int x = blockIdx.x % 8, j = blockIdx.x / 8, sp = NU / 8; tq = j / sp; us = x * sp + j % sp; }
    else
#endif
    if (TQ % 8 == 0) { const int x = blockIdx.x % 8, j = blockIdx.x / 8; tq = x * (TQ / 8) + j / NU; us = j % NU; }
    else { tq = blockIdx.x / NU; us = blockIdx.x % NU; }
    // (a batch too large for one launch runs as several, one after the other: this launch's TQ teams start at tree group g_base)
    const int g0 = g_base + TGN * tq;                        // the team's tree groups g0 .. g0 + TGN - 1
    unsigned* cnt = T.cnt + (size_t)(g0 / 2) * TEAM_MAX_CNT * TEAM_CNT_STRIDE;   // (one block of counters per 32 trees: a 64-tree team uses every other one)
    const int n_layers = P.n_hidden - 1;                     // hidden->hidden layers 1 .. n_layers

    // ---- this workgroup's trees
    const int tj = lane >> 4;                                // tree slot of the lane
    const bool has_tree = wave == 0 && tj < TPW;
    const int tt = us * TPW + (has_tree ? tj : 0);           // tree within the team: group g0 + tt / 16, column tt % 16
    const int tree = g0 * 16 + tt;
    const bool live = has_tree && tree < P.B;
    const unsigned gtree = (unsigned)(P.tree_base + tree);
    const size_t tb = (size_t)(live ? tree : 0) * P.R;
    Cold* cold = P.cold + tb;
    double* edge_W = P.edge_W + tb;
    float* action = P.action + tb;
    TreeStore<TLDS> ts;
    if constexpr (TLDS != TS_GLOBAL) {
        char* base = (char*)s_ab + team_tree_off(P.tab_n, P.n_sims, KC, TGN) + team_tree_bytes(P.R, CONT, TLDS) * (has_tree ? tj : 0);
        ts.hot = (Rec*)base;
        ts.pool = (typename TreeStore<TLDS>::PoolId*)(base + (size_t)P.R * 16);
        ts.prior = (float*)(base + (size_t)P.R * 16);
        ts.state = nullptr;
    } else {
        ts.hot = (Rec*)(P.hot + tb);
        ts.child = P.child + tb * P.Kp;
        ts.prior = P.prior + tb;
    }
    constexpr bool LDS_COLD = TEAM_LDS_COLD && HP == 1024 && TT == 32 && MINB == 2 && TLDS == TS_LDS8;   // (<= 255 records: 68 KB at most, two per CU)
    if constexpr (LDS_COLD) {
        char* cb = (char*)s_ab + team_tree_off(P.tab_n, P.n_sims, KC, TGN) + team_tree_bytes(P.R, CONT, TLDS) * TPW + team_cold_bytes(P.R) * (has_tree ? tj : 0);
        cold = (Cold*)cb;
        edge_W = (double*)(cb + (size_t)P.R * 64);
        action = (float*)(cb + (size_t)P.R * 72);
    }
    TreeState st = {};
    // (the step's critical path through a workgroup ends with its trees' observations: what phase B does for the new node beyond that
    // -- records, child list, reward, cold record -- waits until the workgroup has arrived at the hand-off)
    // (MI355X, ms per search, same box: 1024 trees 12.71 -> 12.61, 1536 trees 17.56 -> 17.39; the 64-tree teams lose by it -- 2048 trees
    // 22.33 -> 22.47, and the three-per-CU form has no five registers to spare -- and keep the bookkeeping inside phase B)
    constexpr bool BDEF = TEAM_DEFER && ENV == AZG_ENV_PENDULUM_V1 && !GMM && TT == 32;
    BDeferred bdef = {};
    // Is the whole team on one XCD?  Every workgroup reports its XCC_ID into two zero-initialised words of the team (max of id
    // and max of 7 - id: they add up to 7 only if all ids are equal) ahead of its first arrival; checked behind the first wait.
    // Until then, and whenever the answer is no, hand-off stores write through to memory (sc1).
    bool wt = true;
    if (tid == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 7u;
        __hip_atomic_fetch_max(cnt + TEAM_CNT_XA * TEAM_CNT_STRIDE, xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_max(cnt + TEAM_CNT_XB * TEAM_CNT_STRIDE, 7u - xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const TileMem<true> parts_mem(L.parts);
    for (int i = tid; i < P.tab_n; i += 256) s_sqrt[i] = P.sqrt_tab[i];
    if (CONT) for (int i = tid; i < P.n_sims + 2; i += 256) s_pw[i] = P.pw_need[i];
    if (tid < 32) s_obs[tid] = 0.0f;
    __syncthreads();
    // The network's first layer of this workgroup's OWN new leaves, all HP units of it, on the vector ALU straight from the
    // observations in LDS (K = obs_dim <= 4: the same fma chain from the bias over k = 0..3 as the MFMA form, bit for bit), written
    // into the team's first activation buffer in the hidden layers' B-operand layout: the observations never travel and the
    // first layer costs no hand-off of its own.  Thread: tree slot tid / (256 / TPW), 8 consecutive units.
    // (Round 5 measured the thread's first-layer weights and biases kept in registers for the whole search, 40 of them, instead of a
    // round trip to L2 in front of every first layer: 12.71-12.75 against 12.71-12.72 ms -- the store drain behind it hides the loads.)
    const TileMem<true> act0(L.act[0], true);
    auto first_layer = [&](bool wt_now) {
        constexpr int TPT = 256 / TPW;                       // threads per tree
        constexpr int UPT = HP / TPT;                        // units per thread
        const int j = tid / TPT, u0 = (tid % TPT) * UPT;
        static_assert(UPT % 4 == 0, "whole float4s of units per thread");
        const int c = us * TPW + j;                          // the tree within the team: group g0 + c / 16, column c % 16
        const float x0 = s_obs[0 * TPW + j], x1 = s_obs[1 * TPW + j], x2 = s_obs[2 * TPW + j], x3 = s_obs[3 * TPW + j];
        TileMem<true> out = act0;
        out.wt = wt_now;
#pragma unroll
        for (int h = 0; h < UPT / 4; ++h) {
            const int u = u0 + 4 * h;
            f32x4 acc = P.b0u[u / 4];
            const f32x4 wa = P.W0u[u], wb = P.W0u[u + 1], wc = P.W0u[u + 2], wd = P.W0u[u + 3];
            acc.x = __builtin_fmaf(wa.x, x0, acc.x); acc.x = __builtin_fmaf(wa.y, x1, acc.x); acc.x = __builtin_fmaf(wa.z, x2, acc.x); acc.x = __builtin_fmaf(wa.w, x3, acc.x);
            acc.y = __builtin_fmaf(wb.x, x0, acc.y); acc.y = __builtin_fmaf(wb.y, x1, acc.y); acc.y = __builtin_fmaf(wb.z, x2, acc.y); acc.y = __builtin_fmaf(wb.w, x3, acc.y);
            acc.z = __builtin_fmaf(wc.x, x0, acc.z); acc.z = __builtin_fmaf(wc.y, x1, acc.z); acc.z = __builtin_fmaf(wc.z, x2, acc.z); acc.z = __builtin_fmaf(wc.w, x3, acc.z);
            acc.w = __builtin_fmaf(wd.x, x0, acc.w); acc.w = __builtin_fmaf(wd.y, x1, acc.w); acc.w = __builtin_fmaf(wd.z, x2, acc.w); acc.w = __builtin_fmaf(wd.w, x3, acc.w);
            // unit u = 16 t + 4 g + r of tree column cc -> float4 ((group * HP/16 + t) * 64 + g * 16 + cc), component r
            out.store4(((size_t)(g0 + c / 16) * (HP / 16) + u / 16) * 64 + ((u % 16) / 4) * 16 + c % 16, act4<true>(P.act, acc));
        }
    };
#ifdef TEAM_STAGGER   /* diagnostic builds: the second half of the grid (every CU's other workgroup) starts TEAM_STAGGER cycles late
                         (round 5, 60k cycles: the same 12.8 ms and the same per-tile times, tools/team_profile.py) */
    if ((blockIdx.x >= gridDim.x / 2) && tid == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)(TEAM_STAGGER)) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
#endif
    if (has_tree) tree_init_root<ENV, TLDS, TPW>(P, st, ts, cold, edge_W, action, tree, live, sub, tj, gtree, s_obs);
    __syncthreads();
    first_layer(true);
    team_arrive(cnt);
#ifdef AZG_STAMPS
    // diagnostic build: cycles of this workgroup (thread 0's clock) in  0 wait for observations | 1..3 tile of layer 1..3 |
    // 4 arrive + wait between layers | 5 wait for the last layer | 6 tree phases | 7 whole loop
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // ... and the tree phases' own parts (tree_phases.cuh: slots 4..15 as in the search kernel's profile; here also 0 head partials into
    // LDS | 1 phase A | 2 phase B | 3 first layer + arrive), thread 0's clock again: a second block of 16 per workgroup behind the first
    unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    TSTAMP(t_begin);
    for (int k = 0; k <= P.n_sims; ++k) {                    // evaluation k follows trace k - 1 (k = 0: the roots)
        const int sim = k - 1;
        TSTAMP(t0);
        if (!team_wait(cnt, (unsigned)(NU * (k + 1)), T, &s_ok)) return;
        TSTAMP(t1);
        TADD(0, t0, t1);
        if (TEAM_SAME_XCD && k == 0) {
            if (tid == 0)
                s_ok = __hip_atomic_load(cnt + TEAM_CNT_XA * TEAM_CNT_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) +
                       __hip_atomic_load(cnt + TEAM_CNT_XB * TEAM_CNT_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 7u;
            __syncthreads();
            wt = s_ok == 0;
            __syncthreads();
        }
        // ---- the hidden layers: this workgroup's 64-unit slice of each, for the team's 32 trees
        for (int l = 1; l <= n_layers; ++l) {
            const int in_buf = (l - 1) & 1;
            TSTAMP(ta);
#if TEAM_WD
            if (l == n_layers) ls_tile_wd<HP, true, TGN, true, KC>(P, L, l, in_buf, us, g0, s_ab, wt);
            else ls_tile_wd<HP, false, TGN, true, KC>(P, L, l, in_buf, us, g0, s_ab, wt);
#else
            if (l == n_layers) ls_tile<HP, true, TGN, 4, true, KC, TEAM_TILE_DBG>(P, L, l, in_buf, us, g0, s_ab, wt);
            else ls_tile<HP, false, TGN, 4, true, KC, TEAM_TILE_DBG>(P, L, l, in_buf, us, g0, s_ab, wt);
#endif
            TSTAMP(tb_);
            team_arrive(cnt + l * TEAM_CNT_STRIDE);
            if (!team_wait(cnt + l * TEAM_CNT_STRIDE, (unsigned)(NU * (k + 1)), T, &s_ok)) return;
            TSTAMP(tc);
            TADD(l < 3 ? l : 3, ta, tb_);
            TADD(l < n_layers ? 4 : 5, tb_, tc);
        }
        // ---- tree phases of this workgroup's trees: the evaluated leaves' values, backup, next trace.
        // Their head partials first (chunk w, output row group q, tree slot j -> s_ab[w * 64 + q * 16 + j]: head_output's layout
        // with the slot as the column)
        TSTAMP(te);
        for (int i = tid; i < NCH * 4 * TPW; i += 256) {
            const int w = i / (4 * TPW), q = (i / TPW) % 4, j = i % TPW, c = us * TPW + j;
            s_ab[w * 64 + q * 16 + j] = parts_mem.load4(((size_t)(g0 + c / 16) * NCH + w) * 64 + q * 16 + c % 16);
        }
        __syncthreads();
        TSTAMP(tp0);
        __builtin_amdgcn_s_setprio(3);   // the walking wave ahead of the other workgroups' MFMA waves on its SIMD (-0.7 % per search)
        if (live) tree_phase_a<ENV, TLDS, GMM, NCH, 64, false, SPEC>(P, st, ts, cold, edge_W, action, tb, sim, sub, tj, gtree, s_ab, P.bhead, s_sqrt STAMP_ARG, s_pw);
        st.need_eval = false;
        TSTAMP(tp1);
        if constexpr (BDEF) { if (k == 0 && live) eps_prepare(P, st, gtree, sub); }   // (no tree_phase_b2 in front of the first trace)
        if (k < P.n_sims) {
            tree_fence();
            if (live) tree_phase_b<ENV, TLDS, GMM, TPW, int, true, false, SPEC, false, BDEF>(P, st, ts, cold, edge_W, action, tb, sub, tj, gtree, s_sqrt, s_pw, s_obs STAMP_ARG, &bdef);
        }
        __builtin_amdgcn_s_setprio(0);   // (after the tree phases on every path, the last step's included)
        TSTAMP(tp2);
#ifdef AZG_STAMPS
        st_acc[0] += tp0 - te; st_acc[1] += tp1 - tp0; st_acc[2] += tp2 - tp1;
#endif
        if (k < P.n_sims) {
            __syncthreads();
            first_layer(wt);
            team_arrive(cnt);
            if constexpr (BDEF) {
                // the rest of the node phase B has just created (tree_phases.cuh: DEFER): behind this workgroup's arrival, in the time the
                // team's other workgroups need to get there
                if (live) tree_phase_b2<ENV, TLDS, SPEC, true>(P, st, ts, cold, edge_W, action, bdef, sub, gtree, s_pw);
                bdef.pending = false;
            }
        }
        TSTAMP(tf);
        TADD(6, te, tf);
#ifdef AZG_STAMPS
        st_acc[3] += tf - tp2;
#endif
    }
    // ---- the trees as the results kernels read them
    if (live) {
        if (sub == 0) P.n_rec[tree] = st.nrec;
        if constexpr (TLDS != TS_GLOBAL) {
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
        if constexpr (LDS_COLD) {
            for (int j = sub; j < st.nrec; j += 16) { P.cold[tb + j] = cold[j]; P.edge_W[tb + j] = edge_W[j]; P.action[tb + j] = action[j]; }
        }
    }
#ifdef AZG_STAMPS
    TSTAMP(t_end);
    TADD(7, t_begin, t_end);
#ifdef TEAM_CENSUS
    // where this workgroup ran: HW_ID (wave, simd, cu, sh, se ...) and XCC_ID, and its block index
    tacc[0] = (unsigned long long)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) << 32) |
              ((unsigned long long)blockIdx.x << 40);
#endif
    if (tid == 0) {
        for (int i = 0; i < 8; ++i) P.stamps[((size_t)tq * NU + us) * 8 + i] = tacc[i];
        for (int i = 0; i < 16; ++i) P.stamps[(size_t)gridDim.x * 8 + ((size_t)tq * NU + us) * 16 + i] = st_acc[i];
    }
#endif
}
