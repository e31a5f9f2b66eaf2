// search_kernel.cuh -- the persistent search kernel: one launch = all n_sims traces of B trees.  A workgroup of NW waves
// owns NG groups of 16 trees (one 16-column MFMA tile each):
//   NW = 4, NG = 1: one wave per SIMD, 4 trees per wave (any network);
//   NW = 8, NG = 1: two waves per SIMD for the network phase, four of the eight walk the trees (4 each): the shape of BASELINE config C;
//   NW = 8, NG = 2: two waves per SIMD, 32 trees = two groups that share every network phase (one barrier per step) and walk together, all
//                   eight waves walking: the second wave of a SIMD fills the other's LDS / memory waits, nothing more -- MFMA and vector
//                   time of a SIMD add up -- (pays off when the batch has more 16-tree groups than the device has CUs).
// NT < 16 ("half-filled tiles"): a group holds only NT = 8 (or 4) trees, the other columns of its MFMA tile carry zeros: more,
// smaller workgroups.  Measured on MI355X (config B's network, ms per search at 2048 / 4096 / 8192 trees): NT = 16: 0.469 / 0.482 /
// 0.626; NT = 8: 0.451 / 0.594 / 1.115; NT = 4: 0.549 / 1.049 / 2.040.  A step of a workgroup takes about as long with 8 trees
// as with 16 (it is a chain of dependent instructions, not a throughput problem), and two workgroups that share a CU slow each
// other down by about 30 %: splitting 4096 trees into 512 half-filled workgroups (two per CU) LOSES 23 % against 256 full ones.
// NT = 8 is used where it does pay: batches of at most 8 trees per CU (one workgroup per CU either way, fewer trees per wave: -4 %).
#pragma once
#include "records.h"
#include "env.cuh"
#include "mlp.cuh"
#include "tree.cuh"
#include "tree_phases.cuh"
#include "results.cuh"
#ifndef AZG_DEFER
#define AZG_DEFER 1   // eight-wave / 16-tree continuous kernels: the new node's bookkeeping behind the barrier, first layer by the non-walking waves (0: A/B builds)
#endif

// Dynamic LDS layout of a workgroup, shared by the kernel and the host's launch planning
struct LdsLayout {
    size_t act_off;    // activation buffers: nbuf x NG groups x HP*64 bytes
    size_t tree_off;   // (TLDS) per-tree regions
    size_t per_tree;   // bytes per tree: R records of 16 B + child-list pool (continuous) or R priors (discrete) (+ env states)
    size_t state_off;  // offset of a tree's env-state slots inside its region (0: none)
    size_t total;
};
// lds_state (discrete LDS trees): n_sims + 1 slots of 4 doubles per tree for the env states of the expanded nodes
__host__ __device__ inline LdsLayout lds_layout(int tab_n, int n_sims, int HP, int NG, int nbuf, int R, bool cont, int tlds, int lds_state = 0, int NT = 16) {
    LdsLayout L;
    L.act_off = ((size_t)tab_n * 8 + (size_t)(n_sims + 2) * 2 + 15) / 16 * 16;
    L.tree_off = L.act_off + (size_t)nbuf * NG * HP * 64;
    L.per_tree = (size_t)R * 16 + (cont ? (size_t)POOL_UNITS(R) * (tlds == TS_LDS9 ? 8 : 4) : (size_t)R * 4);   // 4 ids per pool unit
    L.per_tree = (L.per_tree + 15) / 16 * 16;
    L.state_off = 0;
    if (lds_state && !cont && tlds != TS_GLOBAL) { L.state_off = L.per_tree; L.per_tree += (size_t)(n_sims + 1) * 32; }
    L.total = L.tree_off + (tlds != TS_GLOBAL ? L.per_tree * NT * NG : 0);
    return L;
}
// activation buffers a kernel variant needs: one when a single register-resident hidden layer reads what layer 0 wrote
__host__ __device__ constexpr int act_buffers(int NREG) { return NREG == 1 ? 1 : 2; }

// SPEC: compile-time knowledge about run-time parameters (tree_phases.cuh: Spec<>; 0 = the general code)
template <int ENV, int HP, int NREG, int TLDS, bool GMM, int NW, int NG, int NT = 16, int SPEC = 0>
__global__ __launch_bounds__(64 * NW, NT < 16 ? 2 : 1) void search_kernel(KParams P) {
    constexpr bool CONT = EnvFamily<ENV>::CONT;
    static_assert(NT == 16 || (NG == 1 && NW == 4 && (NT == 8 || NT == 4)), "half-filled tiles: one group, four waves");
    constexpr int TPW = NT * NG;        // trees per workgroup
    // NW = 8, NG = 1 (one 16-tree group, eight waves): all eight waves share the network phase -- a tile fewer each, and the two
    // waves of a SIMD cover each other's LDS round trips and barrier skew there -- but only the first four walk trees (four each,
    // one walking wave per SIMD: waves w and w + 4 share SIMD w).  With all eight walking (two trees each) the SIMT tree phases
    // are issued twice per SIMD and the shape only ties with the 4-wave one; with four walkers it is 4 % faster (config C on
    // MI355X, same box: 1.649 against 1.725 ms per search; all eight walking: 1.707).
    constexpr int NWALK = (NW == 8 && NG == 1) ? 4 : NW;
    constexpr int TPV = TPW / NWALK;    // trees per walking wave (16 lanes each; the wave's other lanes sit the tree phases out)
    typedef typename TreeStore<TLDS>::Rec Rec;
    constexpr int NCH = head_chunks<HP>();   // partial head sums per tree
    constexpr int PSTR = GMM ? 64 : 16;      // entries kept per chunk: all 16 output rows, or rows 0..3 (value + Normal / 2 actions)
    __shared__ f32x4 s_parts[NG * NCH * PSTR];
    // the Acrobot family: six observations -- eight input rows, a second k-step in the network's first layer (KParams::in8 is set)
    constexpr bool IN8 = EnvFamily<ENV>::IN8;
    __shared__ float s_obsT[(IN8 ? 8 : 4) * TPW];
    __shared__ float s_bhead[16];
    __shared__ float s_ln[NREG == 0 ? 2 * 64 : 1];
    __shared__ int s_done;              // discrete mode: trees of this workgroup that have finished their last trace
    // DEFER (eight waves, one 16-tree group, Pendulum family, LDS trees): a simulation step's critical path is network phase -> finish leaf +
    // backup -> descent -> action -> env step -> OBSERVATION.  Everything else phase B does for the node it creates -- edge and node records,
    // the parent's child list, the reward, the cold record -- waits until after the barrier in front of the network phase and is then
    // done by the walking waves (tree_phase_b2) WHILE the other four waves compute the first layer of all sixteen tiles (mlp_forward's
    // SPLIT; a counter in LDS instead of the barrier behind that layer, so that nobody waits for the walking waves there).
    constexpr bool DEFER = AZG_DEFER && NW == 8 && NG == 1 && ENV == AZG_ENV_PENDULUM_V1 && TLDS != TS_GLOBAL && NREG > 0 && !GMM && HP <= 256;
    __shared__ int s_l0;                // (DEFER) first-layer tiles published so far, all network phases of the search
    extern __shared__ double s_dyn[];   // sqrt_tab [tab_n], pw_need [n_sims+2] u16, activation buffers, (TLDS) the trees' hot records

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    // Two waves per SIMD (NW = 8) leave each wave 256 registers, 128 of them weights: nothing that can be re-derived in a few
    // instructions is carried across the network phase there.  The per-tree context (indices, global base pointers, LDS bases)
    // is rebuilt from the thread index at the top of every tree phase, the loop-carried tree state crosses the network phase
    // packed two fields to a register, and the path's rewards / returns are fetched after the network phase instead of during
    // the descent.  With one wave per SIMD (NW = 4: 512 registers) everything stays in registers (measured faster there).
    constexpr bool LEAN = (NW == 8);   // (also for the 16-tree shape: carrying everything instead costs 7 spilled registers and 0.7 % time)
    const LdsLayout L = lds_layout(P.tab_n, P.n_sims, HP, NG, act_buffers(NREG), P.R, CONT, TLDS, P.lds_state, NT);
    double* s_sqrt = s_dyn;
    unsigned short* s_pw = (unsigned short*)(s_dyn + P.tab_n);   // widening thresholds, clamped (a node has < 32768 children)
    f32x4* s_actA = (f32x4*)((char*)s_dyn + L.act_off);
    f32x4* s_actB = act_buffers(NREG) == 2 ? s_actA + NG * (HP / 16 * 64) : s_actA;
    for (int i = tid; i < P.tab_n; i += 64 * NW) s_sqrt[i] = P.sqrt_tab[i];
    if (CONT) for (int i = tid; i < P.n_sims + 2; i += 64 * NW) s_pw[i] = (unsigned short)(P.pw_need[i] < 65535 ? P.pw_need[i] : 65535);
    if (tid < 16) s_bhead[tid] = P.bhead[tid];
    if (tid == 0) { s_done = 0; s_l0 = 0; }

    // register-resident weights
    typedef WRegs<HP, NREG, NW, DEFER> WR;
    WR wr;
    constexpr int NTW = HP / (16 * NW);
    if constexpr (DEFER) {
        // (the first layer belongs to waves NW/2 .. NW-1, 2 * NTW tiles each: mlp_forward's SPLIT)
        if (wave >= NW / 2) {
#pragma unroll
            for (int i = 0; i < 2 * NTW; ++i) {
                wr.w0[i] = P.W0[((wave - NW / 2) * 2 * NTW + i) * 64 + lane];
                wr.b0[i] = P.b0[((wave - NW / 2) * 2 * NTW + i) * 64 + lane];
            }
        }
    } else if constexpr (HP <= 256) {
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            wr.w0[i] = P.W0[(wave * NTW + i) * 64 + lane];
            if constexpr (IN8) { if (P.in8) wr.w0b[i] = P.W0b[(wave * NTW + i) * 64 + lane]; }
            wr.b0[i] = P.b0[(wave * NTW + i) * 64 + lane];
        }
    }
    if (NREG > 0) {
        constexpr int S4 = HP / 16;
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

    // everything a tree phase needs to know about "its" tree, as a function of the thread index
    struct Ctx {
        int sub, tl, tree; bool has_tree, live; unsigned gtree; size_t tb;
        Cold* cold; double* edge_W; float* action; TreeStore<TLDS> ts; const f32x4* my_parts;
    };
    auto make_ctx = [&](int t) {
        Ctx c;
        const int w = t >> 6, ln = t & 63;
        c.sub = ln & 15;
        c.has_tree = w < NWALK && (ln >> 4) < TPV;
        c.tl = c.has_tree ? w * TPV + (ln >> 4) : 0;   // tree within the workgroup
        c.tree = blockIdx.x * TPW + c.tl;
        c.live = c.has_tree && c.tree < P.B;
        c.gtree = (unsigned)(P.tree_base + c.tree);
        c.tb = (size_t)(c.live ? c.tree : 0) * P.R;
        c.cold = P.cold + c.tb;
        c.edge_W = P.edge_W + c.tb;
        c.action = P.action + c.tb;
        if constexpr (TLDS != TS_GLOBAL) {
            // per tree: R records of 16 B, then (continuous) the child-list pool or (discrete) R priors
            char* base = (char*)s_dyn + L.tree_off + L.per_tree * c.tl;
            c.ts.hot = (Rec*)base;
            c.ts.pool = (typename TreeStore<TLDS>::PoolId*)(base + (size_t)P.R * 16);
            c.ts.prior = (float*)(base + (size_t)P.R * 16);
            c.ts.state = L.state_off ? (double*)(base + L.state_off) : nullptr;
        } else {
            c.ts.hot = (Rec*)(P.hot + c.tb);
            c.ts.child = P.child + c.tb * P.Kp;
            c.ts.prior = P.prior + c.tb;
        }
        c.my_parts = s_parts + (c.tl / NT) * NCH * PSTR;   // the head partials of this tree's group
        return c;
    };
    Ctx cx = make_ctx(tid);

#ifdef AZG_STAMPS
    unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    TreeState st = {};
    st.need_eval = false;
    BDeferred bdef = {};   // (DEFER: what the last phase B left for tree_phase_b2)
    if (cx.has_tree) tree_init_root<ENV, TLDS, TPW, IN8>(P, st, cx.ts, cx.cold, cx.edge_W, cx.action, cx.tree, cx.live, cx.sub, cx.tl, cx.gtree, s_obsT);
    __syncthreads();

    // Discrete mode: a tree's traces need a network evaluation only when they create a non-terminal node (CartPole at config B: one
    // trace in five; the others end in a terminal node).  A tree therefore runs up to P.trace_cap traces per step: after backing up
    // a trace that needed no evaluation it starts its next one right away, until a trace does need the network (or the cap is
    // reached: the trees of a wave walk in SIMT lock step, so a tree that keeps going holds up the ones that are waiting for
    // their evaluation).  The order of a tree's traces and everything they compute is unchanged; what changes is how many network
    // phases and barriers a search takes (config B: 101 -> about 40).  Every tree counts its own traces (my_sim); the workgroup
    // leaves when all of its trees are done (s_done, read by everyone right after the step's first barrier).
    // Continuous mode (every trace widens a node and evaluates the new leaf): one trace per step, as before.
    constexpr bool MULTI = !CONT;
    // (the lean packing below does not carry TreeState::chainR / repeat nor the MULTI loop's per-tree trace counter)
    static_assert(!(LEAN && MULTI), "discrete 8-wave shapes: extend the LEAN packing first");
    int my_sim = -1;                    // the trace whose leaf is pending (-1: the root's evaluation); n_sims: the tree is done
    const int n_live = (P.B - (int)blockIdx.x * TPW) < TPW ? (P.B - (int)blockIdx.x * TPW) : TPW;
    // (every unfinished tree completes at least one trace per step, so n_sims + 1 steps always suffice: the bound is a guard, the
    // discrete kernels normally leave through s_done long before)
    for (int sim = -1; sim < P.n_sims; ++sim) {
        // ================= network phase: evaluate the pending leaves =================
        STAMP2(t_a, 0, -1);
        // one barrier: the observations of phase B are visible.  The network runs even if every pending leaf of the workgroup is
        // terminal (rare; its outputs are then ignored): testing for that costs two more barriers per step (__syncthreads_or)
        __syncthreads();
        if constexpr (MULTI) { if (*(volatile int*)&s_done >= n_live) break; }   // (uniform: nobody adds to s_done before the network phase's barriers)
        STAMP2(t_b, 0, 1);
        if constexpr (DEFER) {
            // the rest of the node phase B has just created (the other waves are in the first layer meanwhile)
            STAMP2(t_b2a, 12, -1);
            if (cx.live) tree_phase_b2<ENV, TLDS, SPEC>(P, st, cx.ts, cx.cold, cx.edge_W, cx.action, bdef, cx.sub, cx.gtree, s_pw);
            bdef.pending = false;
            STAMP2(t_b2b, 12, -1);
#ifdef AZG_STAMP_ONLY
            STAMP_ADD(12, t_b2a, t_b2b);   // (single-pair builds only: slot 12 counts descent levels in the full set)
#endif
        }
        unsigned pk0 = 0, pk1 = 0, pk2 = 0, pk3 = 0;
        int tid_o = tid;
        if constexpr (LEAN) {
            // the loop-carried tree state, two fields to a register (record ids, depths, pool units: all < 65536 in LDS trees)
            pk0 = (unsigned)st.nrec | ((unsigned)st.leaf << 16);
            pk1 = (unsigned)st.path_D | ((unsigned)st.kbase << 16);
            pk2 = (unsigned)(st.my_depth + 1) | ((unsigned)st.pid << 16);
            pk3 = (unsigned)st.ptop | ((unsigned)st.need_eval << 16) | ((unsigned)st.resume << 20);
            asm volatile("" : "+v"(pk0), "+v"(pk1), "+v"(pk2), "+v"(pk3));
        }
#ifdef AZG_STAMPS
        mlp_forward<HP, NREG, NW, NG, PSTR, WR, NT, IN8, DEFER>(P, wr, s_obsT, s_actA, s_actB, s_parts, s_ln, wave, lane, st_acc, &s_l0, sim + 2);
#else
        mlp_forward<HP, NREG, NW, NG, PSTR, WR, NT, IN8, DEFER>(P, wr, s_obsT, s_actA, s_actB, s_parts, s_ln, wave, lane, &s_l0, sim + 2);
#endif
        STAMP2(t_c, 1, 2);
        if constexpr (LEAN) {
            // opaque to the optimiser: whatever is derived from these is computed HERE, not kept alive across the network phase
            asm volatile("" : "+v"(pk0), "+v"(pk1), "+v"(pk2), "+v"(pk3), "+v"(tid_o));
            st.nrec = (int)(pk0 & 0xffffu); st.leaf = (int)(pk0 >> 16);
            st.path_D = (int)(pk1 & 0xffffu); st.kbase = (int)(pk1 >> 16);
            st.my_depth = (int)(pk2 & 0xffffu) - 1; st.pid = (int)(pk2 >> 16);
            st.ptop = (int)(pk3 & 0xffffu); st.need_eval = ((pk3 >> 16) & 1u) != 0; st.resume = (int)(pk3 >> 20);
            cx = make_ctx(tid_o);
            // the path's rewards and cumulative returns (the descent did not fetch them: tree_phase_b<..., FETCH = false>)
            st.pr = 0.0; st.pW = 0.0;
            if (cx.live && (MULTI ? my_sim : sim) >= 0 && st.my_depth >= 1) {
                st.pr = CONT ? cx.cold[st.pid].r : discrete_env_reward(Spec<SPEC, ENV>::env(P));
                st.pW = cx.edge_W[st.pid];
            }
        }
        if constexpr (!MULTI) {
            // ================= tree phase A: finish the evaluated leaf, back up =================
            if (cx.live) tree_phase_a<ENV, TLDS, GMM, NCH, PSTR, !CONT, SPEC>(P, st, cx.ts, cx.cold, cx.edge_W, cx.action, cx.tb, sim, cx.sub, cx.tl % NT, cx.gtree, cx.my_parts, s_bhead, s_sqrt STAMP_ARG, s_pw);
            if (sim == P.n_sims - 1) break;
            if constexpr (DEFER) { if (sim < 0 && cx.live) eps_prepare(P, st, cx.gtree, cx.sub); }   // (no tree_phase_b2 in front of the first trace)
            tree_fence();
            STAMP2(t_d, 2, 3);
            // ================= tree phase B: next trace: select down, step the env, expand =================
            st.need_eval = false;
            if (cx.live) tree_phase_b<ENV, TLDS, GMM, TPW, unsigned short, !LEAN, !CONT, SPEC, IN8, DEFER>(P, st, cx.ts, cx.cold, cx.edge_W, cx.action, cx.tb, cx.sub, cx.tl, cx.gtree, s_sqrt, s_pw, s_obsT STAMP_ARG, &bdef);
            tree_fence();
            STAMP2(t_e, 3, -1);
            STAMP_ADD(2, t_c, t_d);   // finish leaf + backup
            STAMP_ADD(3, t_d, t_e);   // select / step / expand
        } else {
            // ================= tree phases A and B, up to trace_cap times: back up, start the next trace, until one needs the network
            bool run = cx.live && my_sim < P.n_sims;
            int k = 0;
            while (run) {
                STAMP2(t_c2, 2, -1);
                tree_phase_a<ENV, TLDS, GMM, NCH, PSTR, !CONT, SPEC>(P, st, cx.ts, cx.cold, cx.edge_W, cx.action, cx.tb, my_sim, cx.sub, cx.tl % NT, cx.gtree, cx.my_parts, s_bhead, s_sqrt STAMP_ARG, s_pw);
                tree_fence();
                STAMP2(t_d, 2, 3);
                st.need_eval = false;
                if (my_sim == P.n_sims - 1) {
                    my_sim = P.n_sims;                          // the tree's last trace is backed up
                    if (cx.sub == 0) atomicAdd(&s_done, 1);
                    run = false;
                } else {
                    my_sim += 1;
                    tree_phase_b<ENV, TLDS, GMM, TPW, unsigned short, !LEAN, !CONT, SPEC, IN8>(P, st, cx.ts, cx.cold, cx.edge_W, cx.action, cx.tb, cx.sub, cx.tl, cx.gtree, s_sqrt, s_pw, s_obsT STAMP_ARG);
                    tree_fence();
                    k += 1;
                    if (st.need_eval || k >= P.trace_cap) run = false;
                }
                STAMP2(t_e, 3, -1);
                STAMP_ADD(2, t_c2, t_d);   // finish leaf + backup
                STAMP_ADD(3, t_d, t_e);    // select / step / expand
            }
        }
        STAMP_ADD(0, t_a, t_b);   // wait at the barrier in front of the network phase
        STAMP_ADD(1, t_b, t_c);   // network phase
    }
    const int nrec = st.nrec;
    const int sub = cx.sub, tree = cx.tree;
    const bool live = cx.live;
    const size_t tb = cx.tb;
    const TreeStore<TLDS>& ts = cx.ts;
#ifdef AZG_STAMPS
    if (lane == 0) for (int i = 0; i < 16; ++i) P.stamps[((size_t)blockIdx.x * NW + wave) * 16 + i] = st_acc[i];
#endif
    // MCTS.return_results (mcts.py:269-307) straight from the trees as they stand (LDS or global): no second launch
    if (live) results_for_tree<CONT, TLDS>(P, ts, cx.cold, cx.action, tb, tree, sub);
    if (live) {
        if (sub == 0) P.n_rec[tree] = nrec;
        if constexpr (TLDS != TS_GLOBAL) {
            // publish the LDS-resident tree in the global format -- only when a dump asked for it (azg_dump_tree): nothing on the
            // product path reads it, return_results was written above straight from LDS
            if (P.publish) {
                RecL* gh = P.hot + tb;
                for (int j = sub; j < nrec; j += 16) {
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
        }
    }
}
