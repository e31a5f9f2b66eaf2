// team.cuh -- wide networks (hidden width >= 512), the whole search as ONE persistent launch of cooperating workgroups.
//
// The lock-step path (lockstep.cuh) pays five kernel boundaries per simulation step and runs its small, latency-bound tree
// kernel (64 workgroups) while the other 192 CUs wait.  Here the batch is cut into TEAMS: 32 trees (two 16-tree groups) and the
// HP/64 workgroups that compute the 64-unit slices of every hidden layer for exactly those trees.  A team never needs data of
// another team, so nothing in the launch is grid-wide: per simulation step a team goes
//     tree phases (its first two workgroups, one per tree group)  ->  hidden layer 1 .. n  (all of its workgroups, one tile each)
// and hands its data from workgroup to workgroup through global memory with one monotonic counter per hand-off (arrive = one
// atomic add per workgroup, wait = one lane polling).  Teams drift apart in time; a CU hosts two workgroups of different teams
// (grid = 2 x CUs at config E), so one team's tree phases and hand-off latencies run under the other's MFMAs.
//
// Visibility (MI355X: a CU's vector L1 is never refreshed by other CUs' stores, the XCD L2s are not coherent with each other):
// every handed-off byte -- activations, observations, head partials -- is stored AND loaded with sc1 buffer instructions
// (TileMem<true>); an arriving workgroup drains its stores (s_waitcnt vmcnt(0) in every wave), meets at its barrier, then one
// lane adds to the counter; a waiting workgroup polls with an sc1 load from one lane and releases the others through its
// barrier.  Placement (blockIdx % 8 = XCD under round-robin dispatch) is used for speed only: a team sits on one XCD.
// Deadlock: every workgroup of the grid has to be resident; the host checks the occupancy before choosing this path, and every
// wait is bounded -- on a time-out the launch raises an abort flag, all workgroups leave, and the host falls back.
// The arithmetic is the lock-step path's (ls_tile, tree_phase_a/b): bit-identical results.
#pragma once
#include "lockstep.cuh"

#define TEAM_CNT_STRIDE 32      // counters 128 bytes apart
#define TEAM_MAX_CNT 8          // counter 0: observations of the step ready; counter l: hidden layer l of the step written
#define TEAM_SPIN_LIMIT (1u << 23)
#ifndef TEAM_STAGGER
#define TEAM_STAGGER 9          // x 8128 cycles: start delay of every second team (about half a simulation step at config E)
#endif
#ifndef TEAM_L0IN
#define TEAM_L0IN 0             // 1: first layer made inside the first hidden layer's staging instead of a team phase of its own
#endif
#define TEAM_CNT_L0 (TEAM_MAX_CNT - 1)   // the first layer's hand-off counter (hidden layers use 1 .. n_hidden - 1)
#ifndef TEAM_MAP
#define TEAM_MAP 0              // 1: consecutive workgroups of an XCD dealt to different teams (census: not how they are placed)
#endif
#ifndef TEAM_GATE
#define TEAM_GATE 1             // the late team of a pair waits for its partner's middle layer before its tree phases
#endif
#define LS_TILE_STAGE_F4 (2 * (4 + 2) * LS_KC * 64)   // float4 entries of the tile routine's two stages (TG = 2, UT = 4)

struct TeamCtl {
    unsigned* cnt;     // [teams][TEAM_MAX_CNT][TEAM_CNT_STRIDE]
    unsigned* abort;   // != 0: a wait timed out, everybody leaves
};

// all of this workgroup's hand-off stores are on their way: drain, meet, one lane arrives
__device__ __forceinline__ void team_arrive(unsigned* c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wait until the counter reaches `target`; false (uniform over the workgroup): aborted
__device__ __forceinline__ bool team_wait(const unsigned* c, unsigned target, unsigned* abort, volatile int* s_ok) {
    if (threadIdx.x == 0) {
        int ok = 1;
        unsigned spins = 0;
        while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 255u) == 0u) {
                if (spins > TEAM_SPIN_LIMIT) __hip_atomic_store(abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
            }
        }
        *s_ok = ok;
    }
    __syncthreads();
    const bool ok = *s_ok != 0;
    __syncthreads();   // everybody has read the verdict before the next wait overwrites it
    return ok;
}

template <int ENV, int HP, bool GMM>
__global__ __launch_bounds__(256, 2) void ls_team_kernel(KParams P, LockStep L, TeamCtl T, int TQ) {
    constexpr bool CONT = (ENV != AZG_ENV_CARTPOLE);
    constexpr int NU = HP / 64, NCH = HP / 64;
    extern __shared__ f32x4 s_ab[];     // the tile routine's two stages (between tiles: the group's head partials, tree workgroups),
                                        // then sqrt_tab [tab_n] and pw_need [n_sims + 2]
    __shared__ float s_obs[64];         // [4][16] observations of the group's new leaves
    __shared__ int s_ok;
    double* s_sqrt = (double*)(s_ab + LS_TILE_STAGE_F4);
    int* s_pw = (int*)(s_sqrt + P.tab_n);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, sub = lane & 15;
    // team and slice of this workgroup: the NU workgroups of a team have equal blockIdx % 8 (one XCD under round-robin placement)
    int tq, us;
    bool late = false;   // this team starts half a step late (see below)
    int partner = -1;    // the team that shares this team's CUs (late teams)
    if (TQ % 8 == 0) {
        const int x = blockIdx.x % 8, j = blockIdx.x / 8, per_x = TQ / 8;   // XCD, index in the XCD's share, teams per XCD
#if TEAM_MAP == 1
        // consecutive workgroups of an XCD land on the same CU (depth-first placement): deal them to different teams
        if (per_x % 2 == 0) {
            const int slot = j & 1, r = j >> 1;
            tq = x * per_x + slot * (per_x / 2) + r / NU; us = r % NU;
            late = slot != 0;
        } else
#endif
        {
            tq = x * per_x + j / NU; us = j % NU;
            // workgroups j and j + (share / 2) of an XCD's share are the two residents of one CU (round-robin, breadth first)
            late = (int)(blockIdx.x / 8) >= (int)(gridDim.x / 16);
            if (late && per_x % 2 == 0) partner = tq - per_x / 2;
        }
    } else { tq = blockIdx.x / NU; us = blockIdx.x % NU; }
#ifdef TEAM_HALF
    if (late) return;   // experiment (timing only, half of the trees are not searched): one workgroup per CU
#endif
    const int G = (P.B + TREES_PER_WG - 1) / TREES_PER_WG;
    const int g0 = 2 * tq;                                   // the team's tree groups g0, g0 + 1
    const int n_tree_wg = G - g0 >= 2 ? 2 : 1;               // (the last team of an odd number of groups has one)
    const bool tree_wg = us < n_tree_wg;
    unsigned* cnt = T.cnt + (size_t)tq * TEAM_MAX_CNT * TEAM_CNT_STRIDE;
    const int n_layers = P.n_hidden - 1;                     // hidden->hidden layers 1 .. n_layers

    // ---- tree workgroups: tables, tree storage, root
    const int tg = g0 + us;                                  // (tree workgroups) the tree group
    const int tl = wave * 4 + (lane >> 4);
    const int tree = tg * TREES_PER_WG + tl;
    const bool live = tree_wg && tree < P.B;
    const unsigned gtree = (unsigned)(P.tree_base + tree);
    const size_t tb = (size_t)(live ? tree : 0) * P.R;
    Cold* cold = P.cold + tb;
    double* edge_W = P.edge_W + tb;
    float* action = P.action + tb;
    TreeStore<false> ts;
    ts.hot = P.hot + tb;
    ts.child = P.child + tb * P.Kp;
    ts.prior = P.prior + tb;
    TreeState st = {};
    const TileMem<true> obs_mem(L.obsT), parts_mem(L.parts);
    if (tree_wg) {
#if TEAM_STAGGER > 0
        // Teams start together and do the same work: left alone they stay in step, all tree phases at once and all layers at
        // once, and nothing overlaps.  An offset between the two teams that share a set of CUs persists (each runs faster while
        // the other is in its tree phase), so one of them starts late: with round-robin dispatch the second half of an XCD's
        // workgroups are the second residents of its CUs (a speed matter only).
        if (late)
            for (int i = 0; i < TEAM_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
#endif
        for (int i = tid; i < P.tab_n; i += 256) s_sqrt[i] = P.sqrt_tab[i];
        if (CONT) for (int i = tid; i < P.n_sims + 2; i += 256) s_pw[i] = P.pw_need[i];
        if (tid < 64) s_obs[tid] = 0.0f;
        __syncthreads();
        tree_init_root<ENV, false>(P, st, ts, cold, edge_W, action, tree, live, sub, tl, gtree, s_obs);
        __syncthreads();
        if (tid < 16) obs_mem.store4((size_t)tg * 16 + tid, ((const f32x4*)s_obs)[tid]);
        team_arrive(cnt);
    }
#ifdef AZG_STAMPS
    // diagnostic build: cycles of this workgroup (thread 0's clock) in  0 wait for observations | 1..3 tile of layer 1..3 |
    // 4 arrive + wait between layers | 5 wait for the last layer (tree workgroups) | 6 tree phases | 7 whole loop
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define TSTAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define TADD(slot, a, b) tacc[slot] += (b) - (a)
#else
#define TSTAMP(v)
#define TADD(slot, a, b)
#endif
    TSTAMP(t_begin);
    for (int k = 0; k <= P.n_sims; ++k) {                    // evaluation k follows trace k - 1 (k = 0: the roots)
        const int sim = k - 1;
        TSTAMP(t0);
        if (!team_wait(cnt, (unsigned)(n_tree_wg * (k + 1)), T.abort, &s_ok)) return;
        TSTAMP(t1);
        TADD(0, t0, t1);
        // ---- the network's first layer (K = obs_dim <= 4: one MFMA k-step per tile): this workgroup's 4 tiles of it, for
        // both tree groups, handed to the team like a hidden layer.  (Made inside the first hidden layer's operand staging
        // instead -- ls_tile's L0IN -- it cost that tile 18k cycles more per step: every slice recomputes all of it.)
        if (!TEAM_L0IN) {
            const int tile = us * 4 + wave;
            const float w0 = P.W0[tile * 64 + lane];
            const f32x4 b0 = P.b0[tile * 64 + lane];
            const TileMem<true> act0(L.act[0]);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float bo = obs_mem.load1((size_t)(g0 + i) * 64 + lane);
                act0.store4(((size_t)(g0 + i) * (HP / 16) + tile) * 64 + lane, act4<true>(P.act, __builtin_amdgcn_mfma_f32_16x16x4f32(w0, bo, b0, 0, 0, 0)));
            }
            team_arrive(cnt + TEAM_CNT_L0 * TEAM_CNT_STRIDE);
            if (!team_wait(cnt + TEAM_CNT_L0 * TEAM_CNT_STRIDE, (unsigned)(NU * (k + 1)), T.abort, &s_ok)) return;
        }
        // ---- the hidden layers: this workgroup's 64-unit slice of each, for the team's 32 trees
        for (int l = 1; l <= n_layers; ++l) {
            const int in_buf = (l - 1) & 1;
            TSTAMP(ta);
            if (l == n_layers) {
                if (l == 1 && TEAM_L0IN) ls_tile<HP, true, 2, 4, true, true>(P, L, l, in_buf, us, g0, s_ab);
                else ls_tile<HP, true, 2, 4, false, true>(P, L, l, in_buf, us, g0, s_ab);
            } else {
                if (l == 1 && TEAM_L0IN) ls_tile<HP, false, 2, 4, true, true>(P, L, l, in_buf, us, g0, s_ab);
                else ls_tile<HP, false, 2, 4, false, true>(P, L, l, in_buf, us, g0, s_ab);
            }
            TSTAMP(tb_);
            team_arrive(cnt + l * TEAM_CNT_STRIDE);
            if (l < n_layers && !team_wait(cnt + l * TEAM_CNT_STRIDE, (unsigned)(NU * (k + 1)), T.abort, &s_ok)) return;
            TSTAMP(tc);
            TADD(l < 3 ? l : 3, ta, tb_);
            TADD(4, tb_, tc);
        }
        if (!tree_wg) continue;
        // ---- tree phases of this workgroup's group: the evaluated leaves' values, backup, next trace
        TSTAMP(td);
#if TEAM_GATE
        // the late team of a CU-sharing pair starts its tree phases when its partner is in the middle of its layers, so that
        // each team's tree phases and hand-off waits fall under the other's MFMAs (the partner never waits for this team)
        if (late && partner >= 0 &&
            !team_wait(T.cnt + ((size_t)partner * TEAM_MAX_CNT + (n_layers + 1) / 2) * TEAM_CNT_STRIDE, (unsigned)(NU * (k + 1)), T.abort, &s_ok)) return;
#endif
        if (!team_wait(cnt + n_layers * TEAM_CNT_STRIDE, (unsigned)(NU * (k + 1)), T.abort, &s_ok)) return;
        TSTAMP(te);
        TADD(5, td, te);
        for (int i = tid; i < NCH * 64; i += 256) s_ab[i] = parts_mem.load4((size_t)tg * NCH * 64 + i);   // the group's head partials
        __syncthreads();
#ifdef AZG_STAMPS
        unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // diagnostic build: discarded here
#endif
        if (live) tree_phase_a<ENV, false, GMM, NCH>(P, st, ts, cold, edge_W, action, tb, sim, sub, tl, gtree, s_ab, P.bhead);
        st.need_eval = false;
        if (k < P.n_sims) {
            __threadfence_block();
            if (live) tree_phase_b<ENV, false, GMM>(P, st, ts, cold, edge_W, action, tb, sub, tl, gtree, s_sqrt, s_pw, s_obs STAMP_ARG);
            __syncthreads();
            if (tid < 16) obs_mem.store4((size_t)tg * 16 + tid, ((const f32x4*)s_obs)[tid]);
            team_arrive(cnt);
        } else if (live && sub == 0) {
            P.n_rec[tree] = st.nrec;
        }
        TSTAMP(tf);
        TADD(6, te, tf);
    }
#ifdef AZG_STAMPS
    TSTAMP(t_end);
    TADD(7, t_begin, t_end);
#ifdef TEAM_CENSUS
    // where this workgroup ran: HW_ID (wave, simd, cu, sh, se ...) and XCC_ID, and its block index
    tacc[0] = (unsigned long long)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) << 32) |
              ((unsigned long long)blockIdx.x << 40);
#endif
    if (tid == 0) for (int i = 0; i < 8; ++i) P.stamps[((size_t)tq * NU + us) * 8 + i] = tacc[i];
#endif
}
