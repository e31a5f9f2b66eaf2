// results.cuh -- MCTS.return_results for one tree, shared by the search kernel's epilogue and results_kernel (aux_kernels.cuh).
#pragma once
#include "records.h"
#include "tree.cuh"

// MCTS.return_results (mcts.py:269-307) of one tree by its 16 lanes: lane a = root child a (a + 16, ... for roots with more
// children).  Everything that is written per child is independent across lanes; the root's totals are row reductions that do not
// depend on the order (integer sum, maximum); the on-policy value target, whose float64 sum is order-sensitive, is added up by the
// tree's first lane in the reference's order.  `ts` is the tree wherever it lives: the search kernel's epilogue hands in its
// LDS-resident trees, results_kernel the published global ones (lock-step / team kernels).
template <bool CONT, int TLDS>
__device__ __forceinline__ void results_for_tree(const KParams& P, const TreeStore<TLDS>& ts, const Cold* cold, const float* action, size_t tb,
                                                 int tree, int sub) {
    typedef typename TreeStore<TLDS>::Rec Rec;
    const int Kmax = P.res_Kmax;
    const Rec root = ts.hot[0];
    const int nc = root.n_child;
    int tot_l = 0;
    double qmax_l = -__builtin_huge_val();
    for (int a = sub; a < Kmax; a += 16) {
        const int k = a < nc ? (CONT ? ts.child_at(0, root, a, P.Kp) : (int)root.first + a) : -1;
        const Rec h = ts.hot[k >= 0 ? k : 0];
        const size_t o = (size_t)tree * Kmax + a;
        P.res_actions[o] = k >= 0 ? (CONT ? action[k] : (float)a) : 0.0f;
        P.res_counts[o] = k >= 0 ? (int)h.edge_n : 0;
        P.res_Q[o] = k >= 0 ? h.Q : 0.0;
        const bool ex = k >= 0 && (h.flags & FLAG_EXPANDED);
        P.res_child_n[o] = ex ? (int)h.node_n : -1;
        for (int s2 = 0; s2 < P.S; ++s2) P.res_child_state[o * P.S + s2] = ex ? cold[k].s[s2] : 0.0;
        if (k >= 0) { tot_l += (int)h.edge_n; qmax_l = h.Q > qmax_l ? h.Q : qmax_l; }
    }
    for (int m = 1; m < 16; m <<= 1) {   // row totals (order-independent)
        tot_l += __shfl_xor(tot_l, m, 16);
        const double o = __shfl_xor(qmax_l, m, 16);
        qmax_l = o > qmax_l ? o : qmax_l;
    }
    if (sub == 0) {
        const double qmax = nc > 0 ? qmax_l : 0.0;
        double onp = 0.0;
        if (P.res_v_target == AZG_VT_ON_POLICY) {
            auto kid = [&](int a) { return ts.hot[CONT ? ts.child_at(0, root, a, P.Kp) : (int)root.first + a]; };
            const long tot = tot_l;
            if (!CONT) {
                for (int a = 0; a < nc; ++a) { const Rec h = kid(a); onp += ((double)(int)h.edge_n / (double)tot) * h.Q; }
            } else {
                // reference quirk (mcts.py:111 with Q of shape (K,1)): the K x K outer product is summed
                for (int a = 0; a < nc; ++a) {
                    const double qa = kid(a).Q;
                    for (int b2 = 0; b2 < nc; ++b2) onp += ((double)(int)kid(b2).edge_n / (double)tot) * qa;
                }
            }
        }
        P.res_vt[tree] = P.res_v_target == AZG_VT_ON_POLICY ? onp : qmax;
        P.res_nch[tree] = nc;
        P.res_root_V[tree] = cold[0].V;
    }
    if (CONT && P.ncomp >= 2) {
        for (int i = sub; i < 3 * P.ncomp; i += 16) {
            const int part = i / P.ncomp, c = i % P.ncomp;
            P.res_root_dist[(size_t)tree * 3 * P.ncomp + i] = P.gmm[tb * 3 * GMM_MAXC + part * GMM_MAXC + c];
        }
    } else if (CONT) {
        if (sub == 0) { P.res_root_dist[(size_t)tree * 2] = cold[0].mu; P.res_root_dist[(size_t)tree * 2 + 1] = cold[0].sg; }
    } else {
        for (int d = sub; d < P.nd; d += 16) P.res_root_dist[(size_t)tree * P.nd + d] = ts.prior[(int)root.first + d];
    }
}

