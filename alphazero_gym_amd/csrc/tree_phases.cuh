// tree_phases.cuh -- the two tree phases of one simulation step for one tree (16 lanes), shared by the persistent search
// kernel (search_kernel.cuh) and the lock-step kernels for wide networks (lockstep.cuh).
#pragma once
#ifndef EARLY_COLD
#define EARLY_COLD 1   // tree_phase_b, continuous mode: see there (measured: 2x128 networks -0.7 %, config C within noise)
#endif
#include "records.h"
#include "env.cuh"
#include "mlp.cuh"
#include "tree.cuh"

// What a 16-lane group carries from one phase to the next (group-uniform unless noted)
struct TreeState {
    int nrec;             // records allocated so far
    unsigned eps_draws;   // epsilon-greedy draws consumed
    int leaf;             // record to be evaluated / backed up
    bool need_eval;       // the leaf needs a network evaluation
    int path_D;           // depth of the leaf on the current trace
    int my_depth, pid;    // per lane: slot (depth & 15) of the trace's path: depth and record id ...
    double pr, pW;        // ... and that record's reward and cumulative return W (consumed by backup_path)
    int kbase;            // progressive-widening noise cache: lane `sub` holds the N(0,1) draw of record kbase + sub
    float eps_c;          // per lane
    float eps_next;       // (DEFER) the draw of record nrec, the next one to be created: taken out of the cache behind the barrier (eps_prepare)
    int ptop;             // LDS trees: next free 4-byte unit of the child-list pool
    int resume;           // discrete mode with cached selections: depth at which the next descent leaves the path of this trace (0: the root)
    double chainR;        // per lane: the slot's return of the last backup (reused when the next trace is the same trace again)
    bool repeat;          // the pending trace is the previous one again: same path, same terminal leaf (set by tree_phase_b)
};

// Compile-time knowledge about run-time parameters.  SPEC = 1 (chosen by dispatch.cuh when it holds, for the kernel shapes the BASELINE
// configurations run on; SPEC = 0 is the general code, same results): no epsilon-greedy selection, lowest-index ties; the discrete
// family: CartPole with its two actions and no carried root count beyond the sqrt table; the Pendulum family: Pendulum-v1.  Every test on those parameters then folds away: no scalar compare / branch /
// exec-mask bookkeeping around paths that are never taken, fewer values kept live in scalar registers.
// (ENV: the kernel's env family -- the Acrobot family knows its env and its three actions at compile time whatever SPEC says)
template <int SPEC, int ENV = AZG_ENV_CARTPOLE> struct Spec {
    static __device__ __forceinline__ bool eps0(const KParams& P) { return SPEC ? true : P.epsilon == 0.0; }
    static __device__ __forceinline__ bool tie_random(const KParams& P) { return SPEC ? false : P.tie_random != 0; }
    static __device__ __forceinline__ bool plain(const KParams& P) { return SPEC ? true : (P.epsilon == 0.0 && !P.tie_random); }
    static __device__ __forceinline__ int A(const KParams& P) { return ENV == AZG_ENV_ACROBOT ? 3 : (SPEC ? 2 : P.A); }   // (discrete families)
    static __device__ __forceinline__ int env(const KParams& P) {                                                       // (discrete families)
        return ENV == AZG_ENV_ACROBOT ? (int)AZG_ENV_ACROBOT : (SPEC ? (int)AZG_ENV_CARTPOLE : P.env_id);
    }
    static __device__ __forceinline__ int v1(const KParams& P) { return SPEC ? 1 : P.v1; }                         // (Pendulum family: Pendulum-v1)
    // every node count the search can meet lies inside the host-built sqrt(n + 1) table (no carried root count beyond it)
    static __device__ __forceinline__ bool in_table(const KParams& P, int n) { return SPEC ? true : n < P.tab_n; }
};

// ---- the cached selection of a node ("best"): the child the next descent through the node will take.
// The reference scores a node's children when a trace comes by (selectionUCT).  Between two visits of a node nothing its
// scores depend on changes -- its own count and its children's counts / Q move only when a trace passes through it, and the
// backup of that very trace is the last thing that touches them -- so the arg-max can just as well be taken right after the
// backup and stored with the node: same inputs, same arithmetic, same winner.  The descent then only follows the stored
// children (two LDS round trips per level instead of a dependent chain of child list, child records, division, arg-max), and
// the scoring work moves into the backup, where the nodes of the path are all known at once and are scored side by side
// instead of one after the other -- in discrete mode in ONE pass: a lane per node (refresh_best_own_slot).  Used in discrete mode
// without epsilon-greedy selection.  Continuous mode keeps scoring on the way down: there a trace ends by widening a node, which
// the descent never has to score but the backup would (2.6 scorings per step instead of 1.6), and the register-starved
// kernels do not interleave the scorings: measured 1.74 ms against 1.65 ms at config C.
// Where the field lives: continuous mode: `first` -- the only child while there is one, free from the second child on (LDS
// trees keep longer lists in the pool, global trees in the child table); discrete mode: the winner's INDEX (0 .. A-1) in the
// byte continuous mode uses for `cbase` (RecS / RecM) or in `pad` (RecL).
template <bool CONT> __device__ __forceinline__ int rec_best(const RecS& h) { return CONT ? (int)h.first : (int)h.first + (int)h.cbase; }
template <bool CONT> __device__ __forceinline__ int rec_best(const RecM& h) { return CONT ? (int)h.first : (int)h.first + (int)h.cbase; }
template <bool CONT> __device__ __forceinline__ int rec_best(const RecL& h) { return CONT ? (int)h.first : (int)h.first + (int)h.pad; }
template <bool CONT> __device__ __forceinline__ void set_best(RecS* r, const RecS& h, int c) {
    if (CONT) r->first = (unsigned char)c; else r->cbase = (unsigned char)(c - (int)h.first);
}
template <bool CONT> __device__ __forceinline__ void set_best(RecM* r, const RecM& h, int c) {
    if (CONT) r->first = (unsigned)c; else r->cbase = (unsigned)(c - (int)h.first);
}
template <bool CONT> __device__ __forceinline__ void set_best(RecL* r, const RecL& h, int c) {
    if (CONT) r->first = (unsigned short)c; else r->pad = (unsigned char)(c - (int)h.first);
}

// selectionUCT's scores and arg-max for node p (record hp, K = hp.n_child >= 1 children): the chosen child's record, the same in
// all 16 lanes of the tree.  pick >= 0: that child index instead of the arg-max (epsilon-greedy).
template <int ENV, int TLDS, typename Rec, int SPEC = 0>
__device__ __forceinline__ int select_child(const KParams& P, const TreeStore<TLDS>& ts, int p, const Rec& hp, int sub, const double* s_sqrt,
                                            int pick, unsigned gtree = 0u) {
    constexpr bool CONT = EnvFamily<ENV>::CONT;
    const int K = hp.n_child;
    // sqrt(n + 1): host-built table; a reused root that was searched many times without moving on (discrete mode) can
    // carry a count beyond the table, then the correctly rounded square root is computed in place
    double sq;
    if (CONT || Spec<SPEC, ENV>::in_table(P, (int)hp.node_n)) sq = s_sqrt[hp.node_n];
    else sq = __builtin_sqrt((double)((int)hp.node_n + 1));
    int win_c = 0;
    if (TLDS || K <= 16) {   // (LDS trees have at most 16 children per node)
        // the common case: all children fit one 16-lane row
        const bool valid = sub < K;
        // no branch around the loads: lanes beyond the last child score record 0 and are masked out of the arg-max
        const int si = valid ? sub : 0;
        int c = CONT ? ts.child_at(p, hp, si, P.Kp) : (int)hp.first + si;
        if (!valid) c = 0;
        Rec h = ts.hot[c];
        double ratio = tree_div(sq, (double)((int)h.edge_n + 1));
        double U;
        if (CONT) {
            U = h.Q + P.c_uct * ratio;
        } else {
            float pc = ts.prior[c] * P.c_uct_f;   // float32 product (NumPy >= 2 promotion)
            U = h.Q + (double)pc * ratio;
        }
        if (!SPEC && pick >= 0) win_c = __shfl(c, pick, 16);
        else if (!CONT && K == 2) win_c = argmax2_payload(U, sub, c);
        else win_c = argmax16_payload(U, valid, sub, c);
        if (Spec<SPEC, ENV>::tie_random(P) && pick < 0) {
            // helpers.argmax (helpers.py:46-52): uniform among the children that hold the maximum.  The draw is keyed by the node
            // and its visit count: between two visits of a node nothing its scores depend on changes.
            const double m = rowmax16(U, valid);
            const unsigned row = (unsigned)(__ballot(valid && U == m) >> (threadIdx.x & 48)) & 0xffffu;
            const int cnt = __popc(row);
            if (cnt > 1) {
                const azg_u32x4 b = azg_draw(P.seed, gtree, P.search_idx, ((unsigned)hp.node_n << 16) ^ (unsigned)p, AZG_STREAM_TIE);
                int kth = (int)(b.v[0] % (unsigned)cnt);
                unsigned r2 = row;
                while (kth-- > 0) r2 &= r2 - 1;                // drop the kth lowest tied lanes
                win_c = __shfl(c, __builtin_ctz(r2), 16);
            }
        }
    } else {
        double win_u = 0.0;
        bool have = false;
        for (int base = 0; base < K; base += 16) {   // children are scanned 16 at a time
            const int i = base + sub;
            const bool valid = i < K;
            int c = 0;
            double U = 0.0;
            if (valid) {
                c = CONT ? ts.child_at(p, hp, i, P.Kp) : (int)hp.first + i;
                Rec h = ts.hot[c];
                double ratio = tree_div(sq, (double)((int)h.edge_n + 1));
                if (CONT) {
                    U = h.Q + P.c_uct * ratio;
                } else {
                    float pc = ts.prior[c] * P.c_uct_f;
                    U = h.Q + (double)pc * ratio;
                }
            }
            int w;
            if (pick >= 0) w = (pick >= base && pick < base + 16) ? pick - base : -1;
            else w = argmax16(U, valid, sub);
            if (w >= 0) {
                int wc = __shfl(c, w, 16);
                double wu = __shfl(U, w, 16);
                if (pick >= 0 || !have || wu > win_u) { win_c = wc; win_u = wu; have = true; }
            }
        }
    }
    return win_c;
}

// Re-take the selection of node p after its statistics changed and store it with the node (lane 0 of the tree writes).
template <int ENV, int TLDS, int SPEC = 0>
__device__ __forceinline__ void refresh_best(const KParams& P, const TreeStore<TLDS>& ts, int p, int sub, const double* s_sqrt) {
    constexpr bool CONT = EnvFamily<ENV>::CONT;
    typedef typename TreeStore<TLDS>::Rec Rec;
    const Rec hp = ts.hot[p];
    const int b = select_child<ENV, TLDS, Rec, SPEC>(P, ts, p, hp, sub, s_sqrt, -1);
    if (sub == 0) set_best<CONT>(&ts.hot[p], hp, b);
}

// Discrete mode, two actions: every node of the path at once, each by the lane that holds its path slot (slot d & 15 = depth d,
// record `pid`): the lane reads its node and the node's two children, scores both and stores the winner's index -- no
// cross-lane traffic at all.  `mine`: the lane's slot holds a node of this path above the leaf.
// hp: the lane's own record as backup_path left it (no second read; lanes without a slot hold a zero record, whose "children"
// are records 0 and 1: valid addresses, results unused).
template <int ENV, int TLDS, int SPEC = 0>
__device__ __forceinline__ int refresh_best_own_slot(const KParams& P, const TreeStore<TLDS>& ts, int pid, bool mine, const double* s_sqrt,
                                                     const typename TreeStore<TLDS>::Rec& hp) {
    typedef typename TreeStore<TLDS>::Rec Rec;
    const int p = mine ? pid : 0;
    double sq;
    if (Spec<SPEC, ENV>::in_table(P, (int)hp.node_n)) sq = s_sqrt[hp.node_n];
    else sq = __builtin_sqrt((double)((int)hp.node_n + 1));
    const int c0 = (int)hp.first;
    const Rec h0 = ts.hot[c0], h1 = ts.hot[c0 + 1];
    const float pc0 = ts.prior[c0] * P.c_uct_f, pc1 = ts.prior[c0 + 1] * P.c_uct_f;   // float32 products (NumPy >= 2 promotion)
    const double U0 = h0.Q + (double)pc0 * tree_div(sq, (double)((int)h0.edge_n + 1));
    const double U1 = h1.Q + (double)pc1 * tree_div(sq, (double)((int)h1.edge_n + 1));
    // the first child unless the second is strictly larger (argmax2_payload's rule)
    const int win = c0 + (U0 >= U1 ? 0 : 1);
    if (mine) set_best<false>(&ts.hot[p], hp, win);
    return win;   // (the node's selection: the record the next descent through it takes)
}

// initialize_search + the root's observation (mcts.py:364-383, 589-600); obsT is the [4][TPW] input block of the tree's
// workgroup (TPW trees per workgroup, tl = the tree's index in it)
// OBS8: the workgroup's input block has eight rows (obsT [8][TPW]) and the env is asked for up to eight observations (discrete family,
// general kernels: Acrobot's six)
template <int ENV, int TLDS, int TPW = 16, bool OBS8 = false>
__device__ __forceinline__ void tree_init_root(const KParams& P, TreeState& st, const TreeStore<TLDS>& ts, Cold* cold, double* edge_W,
                                               float* action, int tree, bool live, int sub, int tl, unsigned gtree, float* obsT) {
    constexpr bool CONT = EnvFamily<ENV>::CONT;
    constexpr int S = CONT ? 2 : 4;
    typedef typename TreeStore<TLDS>::Rec Rec;
    st.nrec = 1; st.eps_draws = 0; st.leaf = 0; st.need_eval = live;
    st.path_D = 0; st.my_depth = -1; st.pid = 0; st.pr = 0.0; st.pW = 0.0;
    st.kbase = 1; st.eps_c = 0.0f; st.eps_next = 0.0f; st.ptop = 0; st.resume = 0; st.chainR = 0.0; st.repeat = false;
    if (CONT && live) st.eps_c = azg_normal(P.seed, gtree, P.search_idx, (unsigned)(st.kbase + sub));
        double rs[S], sn;
#pragma unroll
        for (int k = 0; k < S; ++k) rs[k] = (live && k < P.S) ? P.roots[(size_t)tree * P.S + k] : 0.0;   // (P.S: the env's own state width)
        float obs[OBS8 ? 8 : 4];
        if constexpr (OBS8) { sn = 0.0; azg_acrobot_obs(rs, obs); obs[6] = 0.0f; obs[7] = 0.0f; }
        else env_obs<ENV == AZG_ENV_ACROBOT ? AZG_ENV_CARTPOLE : ENV>(rs, obs, &sn);
        if (live && sub == 0) {
            Rec h = make_edge<Rec>(0.0, 0);
            h.node_n = (decltype(h.node_n))P.carry[tree];
            h.flags = FLAG_EXPANDED;
            clear_pad(h);
            if (CONT) set_wnext(h, pw_at(P.pw_need, (int)P.carry[tree], P.n_sims + 1) > 0);   // (no children yet)
            ts.hot[0] = h;
            Cold c;
#pragma unroll
            for (int k = 0; k < 4; ++k) c.s[k] = k < S ? rs[k] : 0.0;
            if (CONT) c.s[2] = sn;
            c.r = 0.0; c.V = 0.0f; c.mu = 0.0f; c.sg = 0.0f; c.pad = 0.0f;
            cold[0] = c;
            edge_W[0] = 0.0;
            if (CONT) action[0] = 0.0f;
            if constexpr (!CONT && TLDS != TS_GLOBAL) {
                if (ts.state) {   // the root will be the first node to be evaluated: slot 0
#pragma unroll
                    for (int k = 0; k < 4; ++k) ts.state[k] = rs[k];
                }
            }
        }
        if constexpr (!OBS8) {
            if (sub < 4) obsT[sub * TPW + tl] = live ? obs[sub] : 0.0f;
        } else if (sub < 8) {   // (eight input rows)
            float v = sub == 0 ? obs[0] : (sub == 1 ? obs[1] : (sub == 2 ? obs[2] : obs[3]));
            if (sub >= 4) v = sub == 4 ? obs[4] : (sub == 5 ? obs[5] : (sub == 6 ? obs[6] : obs[7]));
            obsT[sub * TPW + tl] = live ? v : 0.0f;
        }
}

// Phase A: give the evaluated leaf its value / policy (evaluation, add_value_estimate: mcts.py:385-416, 602-623; the root's first
// action: mcts.py:673), then back the return up (mcts.py:241-267).  `parts` = the NCH partial head sums of the network phase
// for the tree's group of 16 (PSTR entries per chunk), tl = the tree's column in that group.
// RESUME (discrete mode, cached selections): also work out where the next descent leaves this trace's path (st.resume).
// s_pw: continuous mode, the widening thresholds (the backup keeps the path nodes' widens-at-next-visit bits: tree.cuh set_wnext).
template <int ENV, int TLDS, bool GMM, int NCH, int PSTR = 64, bool RESUME = false, int SPEC = 0, typename PW = int>
__device__ __forceinline__ void tree_phase_a(const KParams& P, TreeState& st, const TreeStore<TLDS>& ts, Cold* cold, double* edge_W,
                                             float* action, size_t tb, int sim, int sub, int tl, unsigned gtree, const f32x4* parts,
                                             const float* bhead, const double* s_sqrt STAMP_PARAM_OPT, const PW* s_pw = nullptr) {
    constexpr bool CONT = EnvFamily<ENV>::CONT;
    typedef typename TreeStore<TLDS>::Rec Rec;
    float V = 0.0f;
    STAMP_A(ta0, 4, -1);
    if (st.need_eval) {
        // discrete mode: outputs 0..3 (value, logits) in one pass over the partials; continuous mode keeps three separate
        // sums (measured at config C: the one-pass form is 3 % slower there, at config B 10 % faster)
        f32x4 out4 = {0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (!CONT) { out4 = head_output4<NCH, PSTR>(parts, bhead, tl); V = out4.x; }
        else V = head_output<NCH, PSTR>(parts, bhead, tl, 0);
        if (CONT) {
            float mu, sg;
            float gd[15];
            if constexpr (GMM) {
                gmm_params<NCH, PSTR>(parts, bhead, tl, P.ncomp, P.ls_min, P.ls_max, gd);
                mu = gd[0]; sg = gd[GMM_MAXC];
                float* g = P.gmm + (tb + st.leaf) * 3 * GMM_MAXC;
                if (sub == 0) {
#pragma unroll
                    for (int i = 0; i < 3 * GMM_MAXC; ++i) g[i] = gd[i];
                }
            } else {
                mu = head_output<NCH, PSTR>(parts, bhead, tl, 1);
                float ls = head_output<NCH, PSTR>(parts, bhead, tl, 2);
                ls = ls < P.ls_min ? P.ls_min : (ls > P.ls_max ? P.ls_max : ls);
                sg = azg_expf(ls);
            }
            if (sub == 0) { cold[st.leaf].V = V; cold[st.leaf].mu = mu; cold[st.leaf].sg = sg; }
            if (sim < 0) {
                // add_pw_action(root) before the first trace (mcts.py:673)
                int k = st.nrec++;
                if constexpr (GMM) gmm_pick(gd, P.ncomp, P.seed, gtree, P.search_idx, (unsigned)k, &mu, &sg);
                float eps = __shfl(st.eps_c, k - st.kbase, 16);
                float a = P.bound_f * azg_tanhf(mu + sg * eps);
                if (sub == 0) {
                    Rec h = make_edge<Rec>((double)V, 0);
                    clear_pad(h);
                    ts.hot[k] = h;
                    edge_W[k] = 0.0;
                    action[k] = a;
                }
                ts.child_append(0, make_edge<Rec>(0.0, 0), 0, k, st.ptop, sub == 0, P.Kp,
                                pw_at(P.pw_need, (int)ts.hot[0].node_n, P.n_sims + 1) - 1 > 0);   // (the root with its first child)
            }
        } else {
            // softmax priors + all num_actions edges with Q_init = V (MCTSDiscrete.evaluation, mcts.py:412-416)
            const int A = Spec<SPEC, ENV>::A(P);
            // (logits 1 .. 3 come with out4; further actions, should an environment have them, through head_output)
            // lane a < A works on action a: its logit, its exp; the maximum and the sum are taken in action order (the sum as
            // ((0 + e_0) + e_1) + ..., the reference's order) from the lanes' values
            float my_logit = sub == 0 ? out4.y : (sub == 1 ? out4.z : out4.w);
            if (A > 3 && sub >= 3 && sub < A) my_logit = head_output<NCH, PSTR>(parts, bhead, tl, 1 + sub);
            float mx = out4.y;
            if (A == 2) {
                mx = out4.z > mx ? out4.z : mx;
            } else {
                for (int a = 1; a < A; ++a) { const float v = __shfl(my_logit, a, 16); mx = v > mx ? v : mx; }
            }
            const float e_mine = azg_expf(my_logit - mx);   // (lanes >= A: an unused value)
            float sum;
            if (A == 2) {
                sum = dpp_f32<DPP_QUAD_BCAST0>(e_mine) + dpp_f32<DPP_QUAD_BCAST1>(e_mine);   // 0 + e_0 is e_0 exactly
            } else {
                sum = 0.0f;
                for (int a = 0; a < A; ++a) sum = sum + __shfl(e_mine, a, 16);
            }
            // (two-action form: only the first quad's lanes hold the sum; only lanes 0 .. A-1 use it)
            int k0 = st.nrec;
            st.nrec += A;
            if (sub < A) {
                float prior_a = e_mine / sum;
                Rec h = make_edge<Rec>((double)V, st.leaf);
                clear_pad(h);
                ts.hot[k0 + sub] = h;
                ts.prior[k0 + sub] = prior_a;
                edge_W[k0 + sub] = 0.0;
            }
            if (sub == 0) {
                cold[st.leaf].V = V;
                ts.hot[st.leaf].n_child = (decltype(ts.hot[st.leaf].n_child))A;
                ts.hot[st.leaf].first = (decltype(ts.hot[st.leaf].first))k0;
            }
            if (Spec<SPEC, ENV>::plain(P)) {
                // the new node's own selection (refresh_best): its edges all start at Q = V with no visits, so its scores
                // are V + (prior_a * c_uct as float32) * (sqrt(n + 1) / 1) -- a division by one is exact
                if (A == 2) {
                    const int nn = (int)ts.hot[st.leaf].node_n;   // 0, or the carried count of a reused root
                    double sq;
                    if (Spec<SPEC, ENV>::in_table(P, nn)) sq = s_sqrt[nn];
                    else sq = __builtin_sqrt((double)(nn + 1));
                    float prior_s = 0.0f;
                    if (sub < A) prior_s = e_mine / sum;
                    const float pc = prior_s * P.c_uct_f;
                    const double U = (double)V + (double)pc * sq;
                    const double o = dpp_f64<DPP_QUAD_XOR1>(U);
                    if (sub == 0) set_best<false>(&ts.hot[st.leaf], make_edge<Rec>(0.0, 0), U >= o ? 0 : 1);   // (index relative to first)
                } else {
                    if (!TLDS) tree_fence();
                    refresh_best<ENV, TLDS, SPEC>(P, ts, st.leaf, sub, s_sqrt);
                }
            }
        }
    }
    STAMP_A(ta1, 4, 5);
    STAMP_A_ADD(4, ta0, ta1);   // finish leaf
    if (sim >= 0) {
        if (!TLDS) tree_fence();   // lane 0's partial record stores above must land before the path is re-read
        const bool keep = !CONT && Spec<SPEC, ENV>::plain(P);   // (cached selections: discrete mode, see rec_best)
        Rec myrec;
        backup_path<CONT, TLDS>(ts, cold, edge_W, V, sub, P.gamma_f, P.gamma, st.path_D, st.my_depth, st.pid, st.pr, st.pW,
                                [&](int pn) { if (keep) refresh_best<ENV, TLDS, SPEC>(P, ts, pn, sub, s_sqrt); }, myrec, st.chainR,
                                RESUME && !CONT && st.repeat,    // (only the kernels that resume descents set st.repeat)
                                CONT ? 0.0 : discrete_env_reward(Spec<SPEC, ENV>::env(P)),
                                // (a pending leaf that needs no evaluation is a terminal node: need_eval = !done)
                                CONT ? 0.0 : (st.need_eval ? discrete_env_reward(Spec<SPEC, ENV>::env(P)) : discrete_env_terminal_reward(Spec<SPEC, ENV>::env(P))),
                                s_pw, P.n_sims + 1);
        STAMP_A(ta2, 5, 6);
        STAMP_A_ADD(5, ta1, ta2);   // backup (return chain, record updates)
        if constexpr (!CONT) {
            if (keep) {
                // the selections of the path's nodes, with their new statistics: depths D-1 .. max(0, D-15) are in the lanes' slots
                // (anything deeper than 16 levels was refreshed by backup_from through the callback above)
                const int D = st.path_D, lo = D > 15 ? D - 15 : 0;
                if (!TLDS) tree_fence();
                if (Spec<SPEC, ENV>::A(P) == 2) {
                    // (a slot holds a path node above the leaf: depth in [lo, D - 1]; a root that was never left is slot 0, depth 0)
                    const bool mine = st.my_depth >= lo && st.my_depth < D;
                    const int win = refresh_best_own_slot<ENV, TLDS, SPEC>(P, ts, st.pid, mine, s_sqrt, myrec);
                    if constexpr (RESUME) {
                        // The next trace follows the stored selections from the root: it walks this trace's path for as long as every
                        // node's (re-taken) selection is still the path's next record, i.e. down to the shallowest node whose
                        // selection moved away -- or to the path's end.  Paths of up to 15 levels have depth d in slot d: lane
                        // order = depth order, so the shallowest such node is the row's lowest lane that votes.
                        const int next_pid = dpp_i32<DPP_ROW_ROR15>(st.pid);   // the record in the slot of depth + 1
                        const unsigned long long votes = __ballot(mine && win != next_pid);
                        const unsigned row = (unsigned)(votes >> (threadIdx.x & 48)) & 0xffffu;   // the 16 lanes of this tree
                        st.resume = D < 16 ? (row ? __builtin_ctz(row) : D) : 0;
                    }
                } else {
                    for (int d = D - 1; d >= lo; --d) refresh_best<ENV, TLDS, SPEC>(P, ts, __shfl(st.pid, d & 15, 16), sub, s_sqrt);
                }
            }
        }
        STAMP_A(ta3, 6, -1);
        STAMP_A_ADD(6, ta2, ta3);   // re-scoring of the path's nodes + where the next descent leaves the path
    }
}

// DEFER (the eight-wave / 16-tree continuous kernels, search_kernel.cuh): what phase B leaves for tree_phase_b2 -- everything about the
// new node that the network's next evaluation does not wait for.  The step's critical path ends with the new leaf's observation (the
// barrier in front of the network phase); the new edge's and node's records, the parent's child list, the reward (a function of the
// parent's state and the action only) and the cold record are written after that barrier, while the workgroup's non-walking waves
// compute the first layer.
// (DEFER) the progressive-widening noise of the NEXT record, made ready off the step's critical path: refill the 16-draw cache when the
// next record lies beyond it (Philox + Box-Muller, once per 16 records) and fetch that record's draw from its lane.  Phase B then widens
// with st.eps_next instead of a cross-lane read (and never refills).  Called behind the root's first action and at the end of tree_phase_b2.
__device__ __forceinline__ void eps_prepare(const KParams& P, TreeState& st, unsigned gtree, int sub) {
    if (st.nrec >= st.kbase + 16) {
        st.kbase = st.nrec;
        st.eps_c = azg_normal(P.seed, gtree, P.search_idx, (unsigned)(st.kbase + sub));
    }
    st.eps_next = __shfl(st.eps_c, st.nrec - st.kbase, 16);
}

struct BDeferred {
    bool pending, widen;
    int p, chosen;          // parent node, the new record
    float cact;             // the action of the edge
};

// the deferred half of phase B (Pendulum family: the nodes never end an episode).  The parent's records are read again here (nothing
// has touched them since the descent): carrying them across the barrier would cost the lean walkers registers they do not have.
template <int ENV, int TLDS, int SPEC, bool FETCH = false, typename PW = int>
__device__ __forceinline__ void tree_phase_b2(const KParams& P, TreeState& st, const TreeStore<TLDS>& ts, Cold* cold, double* edge_W, float* action,
                                              const BDeferred& d, int sub, unsigned gtree, const PW* s_pw) {
    typedef typename TreeStore<TLDS>::Rec Rec;
    static_assert(ENV == AZG_ENV_PENDULUM_V1, "deferred expansion: the Pendulum family");
    if (!d.pending) return;
    const double ps0 = cold[d.p].s[0], ps1 = cold[d.p].s[1];   // the parent's env state (reward)
    if (d.widen) {
        // MCTSContinuous.add_pw_action (mcts.py:625-654): the edge, with the node half filled in at once (flags)
        const float V = cold[d.p].V;        // the parent's value: the new edge's Q_init
        const Rec hp = ts.hot[d.p];
        if (sub == 0) {
            Rec h = make_edge<Rec>((double)V, d.p);
            h.flags = FLAG_EXPANDED;
            clear_pad(h);
            set_wnext(h, P.pw0 > 0);        // (a node without visits or children)
            ts.hot[d.chosen] = h;
            edge_W[d.chosen] = 0.0;
            action[d.chosen] = d.cact;
        }
        ts.child_append(d.p, hp, (int)hp.n_child, d.chosen, st.ptop, sub == 0, P.Kp,
                        pw_at(s_pw, (int)hp.node_n, P.n_sims + 1) - ((int)hp.n_child + 1) > 0);
    } else if (sub == 0) {
        Rec h = ts.hot[d.chosen];
        h.flags = FLAG_EXPANDED;
        set_wnext(h, P.pw0 > 0);
        ts.hot[d.chosen] = h;
    }
    // MCTS.expansion (mcts.py:216-238): the reward of the step that led here (mcts.py:687: divided by PENDULUM_R_SCALE) and the rest of the
    // node's cold record (phase B stored the env state)
    const double sp[2] = {ps0, ps1};
    const double r = pendulum_reward(sp, d.cact) / P.reward_scale;
    if constexpr (FETCH) { if (sub == (st.path_D & 15)) st.pr = r; }   // (the leaf's path slot: phase B left 0 there)
    if (sub == 0) {
        float zero = 0.0f;
        asm volatile("" : "+v"(zero));
        Cold* c = cold + d.chosen;
        c->s[3] = 0.0; c->r = r; c->V = zero; c->mu = zero; c->sg = zero; c->pad = zero;
    }
    eps_prepare(P, st, gtree, sub);
}

// Phase B: the next trace: descend by UCT / PUCT (selectionUCT: mcts.py:464-493, 704-741), widen or pick an unexpanded edge,
// step the environment and create the node (expansion: mcts.py:216-238); leaves the new leaf's observation in obsT.
// FETCH: the path's rewards / returns (st.pr, st.pW) are fetched on the way (false: the caller fetches them after the network phase).
// RESUME (discrete mode, cached selections): the descent starts where the last trace's path is left (st.resume, set by
// tree_phase_a<..., RESUME = true>) instead of at the root; the path slots above that depth are still in the lanes.
// DEFER: the new node's records, reward and cold record are left to tree_phase_b2 (`def`), which also hands the leaf's path slot its
// reward (st.pr) when the path's rewards travel in the lanes (FETCH).
template <int ENV, int TLDS, bool GMM, int TPW = 16, typename PW = int, bool FETCH = true, bool RESUME = false, int SPEC = 0, bool OBS8 = false, bool DEFER = false>
__device__ __forceinline__ void tree_phase_b(const KParams& P, TreeState& st, const TreeStore<TLDS>& ts, Cold* cold, double* edge_W,
                                             float* action, size_t tb, int sub, int tl, unsigned gtree, const double* s_sqrt,
                                             const PW* s_pw, float* obsT STAMP_PARAM, BDeferred* def = nullptr) {
    static_assert(!DEFER || (ENV == AZG_ENV_PENDULUM_V1 && !GMM), "deferred expansion: Pendulum family, squashed-Normal head");
    constexpr bool CONT = EnvFamily<ENV>::CONT;
    constexpr int S = CONT ? 2 : 4;
    typedef typename TreeStore<TLDS>::Rec Rec;
    st.need_eval = false;
    if (CONT && !DEFER && st.nrec >= st.kbase + 16) {   // (DEFER: eps_prepare has done it)
        st.kbase = st.nrec;
        st.eps_c = azg_normal(P.seed, gtree, P.search_idx, (unsigned)(st.kbase + sub));
    }
    int p = 0;
    bool resumed = false;
    st.repeat = false;
    if constexpr (RESUME && !CONT) {
        if (Spec<SPEC, ENV>::plain(P) && Spec<SPEC, ENV>::A(P) == 2 && st.resume > 0) {
            // same path as a descent from the root down to depth `resume` (tree_phase_a): go on from that node
            resumed = true;
            p = __shfl(st.pid, st.resume, 16);
            if (st.my_depth > st.resume) st.my_depth = -1;   // the slots below it belong to the old trace
            st.path_D = st.resume;
        }
    }
    // discrete mode: every step of the env pays the same reward, and a path slot that survives from the last trace (depths 1 ..
    // resume) still holds its record's W as the last backup left it: only records that enter the path are fetched
    const double r_step = CONT ? 0.0 : discrete_env_reward(Spec<SPEC, ENV>::env(P));
    const bool keep_slot = !CONT && FETCH && resumed && st.my_depth >= 1;
    Rec hp = ts.hot[p];
    Cold cp;             // cold part of the current node, prefetched one level ahead
    bool from_cold = true;
    if constexpr (!CONT && TLDS != TS_GLOBAL) from_cold = (ts.state == nullptr);   // (LDS-resident env states: read after the descent)
    if (from_cold) cp = cold[p];   // (p: the root, or the node a resumed descent starts from)
    else { cp.s[0] = cp.s[1] = cp.s[2] = cp.s[3] = 0.0; }
    if (!resumed) { st.path_D = 0; st.my_depth = sub == 0 ? 0 : -1; st.pid = 0; }
    if (!keep_slot) { st.pr = 0.0; st.pW = 0.0; }
    int chosen = 0;
    bool widen = false, hit_terminal = false;
    if constexpr (RESUME && !CONT) {
        hit_terminal = resumed && (hp.flags & FLAG_TERMINAL);   // (the old trace ended in a terminal node and nothing moved)
        st.repeat = FETCH && hit_terminal;                       // the same trace again: same slots, same rewards, V = 0 both times
    }
    STAMP2(tb0, 13, -1);
    while (!hit_terminal) {
        STAMP3(tl0, 7, 8, 11);
        const int K = hp.n_child;
        if (CONT) {
            widen = get_wnext(hp);   // NodeContinuous.check_pw (states.py:271-275), evaluated when the node's counts last changed (tree.cuh: set_wnext)
            if (widen) break;
        }
        if (CONT || !Spec<SPEC, ENV>::plain(P)) {
            // scored on the way down (continuous mode; epsilon-greedy selection, MCTS.epsilon_greedy mcts.py:190-195; random ties)
            int pick = -1;
            if (!Spec<SPEC, ENV>::eps0(P)) {
                azg_u32x4 b = azg_draw(P.seed, gtree, P.search_idx, st.eps_draws++, AZG_STREAM_EPS);
                if ((double)azg_u01(b.v[0]) < P.epsilon) pick = (int)(b.v[1] % (unsigned)K);
            }
            chosen = select_child<ENV, TLDS, Rec, SPEC>(P, ts, p, hp, sub, s_sqrt, pick, gtree);
        } else {
            chosen = rec_best<CONT>(hp);   // taken when the node's statistics last changed (refresh_best)
        }
        STAMP2(tl1, 7, -1);
        STAMP_ADD(7, tl0, tl1);    // the level's selection
        STAMP2(tl2, 8, 9);
        // continuous mode: the chosen child's cold record (its cached policy and env state: needed at once if the trace widens there)
        // is requested before the hot record's LDS round trip, not after it (an edge without a child node: a record of the tree all
        // the same, its contents unused)
        Cold cn;
        if constexpr (CONT && EARLY_COLD) cn = cold[chosen];
        Rec hc = ts.hot[chosen];
        if (!(hc.flags & FLAG_EXPANDED)) break;   // an edge without a child node: expand it
        STAMP2(tl3, 9, 10);
        st.path_D += 1;
        p = chosen;
        hp = hc;
        if (sub == (st.path_D & 15)) {   // the level's path slot: its depth, its record, its reward and W (used by backup_path)
            st.my_depth = st.path_D; st.pid = chosen;
            // continuous mode (2-3 levels, a slow scored descent): fetched here, in the shadow of the level's LDS waits;
            // discrete mode (8-9 levels of pointer chasing): all levels at once after the loop (measured both ways)
            if (CONT && FETCH) { st.pr = EARLY_COLD ? cn.r : cold[chosen].r; st.pW = edge_W[chosen]; }
            if (!CONT && FETCH) { st.pr = r_step; st.pW = edge_W[chosen]; }   // (consumed by the backup; requested as the record enters the path)
        }
        if constexpr (EnvFamily<ENV>::TERM) {   // (Pendulum never terminates: no exit, and no exit mask to maintain, in its kernels)
            // `while not node.terminal` (mcts.py:441, 682): the trace ends in this existing terminal node, V = 0, no evaluation
            if (hc.flags & FLAG_TERMINAL) { hit_terminal = true; break; }
        }
        // the node's env state for the step that follows if the trace leaves the tree here: requested at every level, the wave
        // waits only for the last one (Pendulum: the whole cold record, its widening needs the cached policy too)
        if (CONT) { if constexpr (EARLY_COLD) cp = cn; else cp = cold[p]; }
        else if (from_cold) { cp.s[0] = cold[p].s[0]; cp.s[1] = cold[p].s[1]; cp.s[2] = cold[p].s[2]; cp.s[3] = cold[p].s[3]; }
        // (LDS-resident env states: read once, below, for the node the trace leaves the tree from)
        STAMP2(tl4, 10, 11);
        STAMP_ADD(8, tl0, tl2);    // whole selection of a level (scores + arg-max)
        STAMP_ADD(9, tl2, tl3);    // chosen record
        STAMP_ADD(10, tl3, tl4);   // path slot + cold prefetch issue
        STAMP_ADD(11, tl0, tl4);   // full level
#if defined(AZG_STAMPS) && !defined(AZG_STAMP_ONLY)
        st_acc[12] += 1;
#endif
    }
    STAMP2(tb1, 13, 14);
    STAMP_ADD(13, tb0, tb1);       // whole descent
    if (hit_terminal) {
        st.leaf = p;
    } else {
        float cact = 0.0f;
        if (widen) {
            // MCTSContinuous.add_pw_action (mcts.py:625-654)
            const int K = hp.n_child;
            chosen = st.nrec++;
            float eps = DEFER ? st.eps_next : __shfl(st.eps_c, chosen - st.kbase, 16);
            float wmu = cp.mu, wsg = cp.sg;
            if constexpr (GMM) {
                float gd[15];
                const float* g = P.gmm + (tb + p) * 3 * GMM_MAXC;
#pragma unroll
                for (int i = 0; i < 3 * GMM_MAXC; ++i) gd[i] = g[i];
                gmm_pick(gd, P.ncomp, P.seed, gtree, P.search_idx, (unsigned)chosen, &wmu, &wsg);
            }
            cact = P.bound_f * azg_tanhf(wmu + wsg * eps);
            if constexpr (!DEFER) {
                if (sub == 0) {
                    Rec h = make_edge<Rec>((double)cp.V, p);
                    clear_pad(h);
                    ts.hot[chosen] = h;
                    edge_W[chosen] = 0.0;
                    action[chosen] = cact;
                }
                ts.child_append(p, hp, K, chosen, st.ptop, sub == 0, P.Kp, pw_at(s_pw, (int)hp.node_n, P.n_sims + 1) - (K + 1) > 0);
            }
        }
        // MCTS.expansion (mcts.py:216-238): step the env from the parent's cached state
        STAMP2(tw1, 14, 15);
        STAMP_ADD(14, tb1, tw1);   // widening (noise, policy parameters, tanh, edge record, child list)
        st.path_D += 1;
        double ns[S], r, sn;
        int done;
        if (CONT) {
            if (!widen) cact = action[chosen];
            if constexpr (DEFER) {
                // (the new state only: the reward is tree_phase_b2's)
                pendulum_dynamics(Spec<SPEC, ENV>::v1(P), cp.s, cp.s[2], cact, ns);
                r = 0.0; done = 0;
            } else {
                if constexpr (ENV == AZG_ENV_MOUNTAINCAR_CONT) mountaincar_cont_step(cp.s, cact, ns, &r, &done);
                else pendulum_step(Spec<SPEC, ENV>::v1(P), cp.s, cp.s[2], cact, ns, &r, &done);
                r = r / P.reward_scale;   // mcts.py:687 (whatever the env: the reference divides every continuous reward by PENDULUM_R_SCALE)
            }
        } else {
            if constexpr (TLDS != TS_GLOBAL) {
                if (ts.state) {
                    const int slot = Spec<SPEC, ENV>::A(P) == 2 ? ((int)hp.first - 1) >> 1 : ((int)hp.first - 1) / P.A;
                    const double* sp = ts.state + 4 * slot;
                    cp.s[0] = sp[0]; cp.s[1] = sp[1]; cp.s[2] = sp[2]; cp.s[3] = sp[3];
                }
            }
            family_env_step<ENV>(Spec<SPEC, ENV>::env(P), cp.s, chosen - (int)hp.first, ns, &r, &done);
        }
        float obs[OBS8 ? 8 : 4];
        if constexpr (OBS8) { sn = 0.0; azg_acrobot_obs(ns, obs); obs[6] = 0.0f; obs[7] = 0.0f; }
        else env_obs<ENV == AZG_ENV_ACROBOT ? AZG_ENV_CARTPOLE : ENV>(ns, obs, &sn);
        STAMP2(tw2, 15, -1);
        STAMP_ADD(15, tw1, tw2);   // env step + observation
        if constexpr (DEFER) {
            def->pending = true; def->widen = widen; def->p = p; def->chosen = chosen; def->cact = cact;
            if (sub == 0) { cold[chosen].s[0] = ns[0]; cold[chosen].s[1] = ns[1]; cold[chosen].s[2] = sn; }   // (the rest of the record: tree_phase_b2)
        } else if (sub == 0) {
            Cold c;
#pragma unroll
            for (int k = 0; k < 4; ++k) c.s[k] = k < S ? ns[k] : 0.0;
            if (CONT) c.s[2] = sn;
            float zero = 0.0f;
            asm volatile("" : "+v"(zero));   // (made here: a zero vector kept in registers for the whole search costs four of them)
            c.r = r; c.V = zero; c.mu = zero; c.sg = zero; c.pad = zero;
            cold[chosen] = c;
            if constexpr (CONT) {
                Rec hn = ts.hot[chosen];
                hn.flags = (unsigned char)(FLAG_EXPANDED | (done ? FLAG_TERMINAL : 0));
                set_wnext(hn, P.pw0 > 0);   // (a node without visits or children)
                ts.hot[chosen] = hn;
            } else {
                ts.hot[chosen].flags = (unsigned char)(FLAG_EXPANDED | (done ? FLAG_TERMINAL : 0));
            }
            if constexpr (!CONT && TLDS != TS_GLOBAL) {
                if (ts.state && !done) {   // the next node to be evaluated in this tree: its edges will start at record nrec
                    double* sp = ts.state + 4 * (Spec<SPEC, ENV>::A(P) == 2 ? (st.nrec - 1) >> 1 : (st.nrec - 1) / P.A);
#pragma unroll
                    for (int k = 0; k < 4; ++k) sp[k] = ns[k];
                }
            }
        }
        if (sub == (st.path_D & 15)) { st.my_depth = st.path_D; st.pid = chosen; st.pr = r; st.pW = 0.0; }
        st.leaf = chosen;
        st.need_eval = !done;
        if constexpr (!OBS8) {
            if (sub < 4) obsT[sub * TPW + tl] = done ? 0.0f : obs[sub];
        } else if (sub < 8) {
            float v = sub == 0 ? obs[0] : (sub == 1 ? obs[1] : (sub == 2 ? obs[2] : obs[3]));
            if (sub >= 4) v = sub == 4 ? obs[4] : (sub == 5 ? obs[5] : (sub == 6 ? obs[6] : obs[7]));
            obsT[sub * TPW + tl] = done ? 0.0f : v;
        }
    }
}
