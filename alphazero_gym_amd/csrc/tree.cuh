// tree.cuh -- 16-lane (one DPP row) tree primitives: arg-max, record storage in LDS or global memory, backup.
#pragma once
#include "records.h"

// A tree belongs to ONE wave: its records -- in LDS or in global memory -- are written and read by the lanes of that wave only, and a wave's
// memory operations take effect in program order: what orders lane 0's record stores with the other lanes' later loads is a fence at
// wavefront scope, which only stops the compiler from moving them (LLVM's AMDGPU memory model generates no instruction for it).  The
// workgroup-scope fence that stood here until round 6 made the wave wait for every outstanding global store (s_waitcnt vmcnt(0)) twice per
// simulation step.
__device__ __forceinline__ void tree_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// n / d for the tree statistics (n finite, d a positive count, both far from the ends of the exponent range): the compiler's own
// IEEE float64 division sequence (v_rcp_f64, two Newton steps, quotient, residual correction) without its operand scaling
// and fix-up steps, which are no-ops for such operands -- the same correctly rounded quotient in 8 instead of 11 instructions
__device__ __forceinline__ double tree_div(double n, double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    double q = n * r;
    double rem = __builtin_fma(-d, q, n);
    return __builtin_fma(rem, r, q);
}

// ------------------------------------------------------------------------------------------------ tree walk (16 lanes per tree)

// cross-lane moves inside a 16-lane row (one tree) on the DPP network: no LDS traffic, one VALU op each
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    unsigned long long u = azg_d2u(v);
    unsigned lo = (unsigned)dpp_i32<CTRL>((int)(unsigned)u), hi = (unsigned)dpp_i32<CTRL>((int)(unsigned)(u >> 32));
    return azg_u2d(((unsigned long long)hi << 32) | lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) { return __builtin_bit_cast(float, dpp_i32<CTRL>(__builtin_bit_cast(int, v))); }
#define DPP_QUAD_BCAST1 0x55   // quad_perm:[1,1,1,1]
#define DPP_QUAD_XOR1 0xB1   // quad_perm:[1,0,3,2]
#define DPP_QUAD_XOR2 0x4E   // quad_perm:[2,3,0,1]
#define DPP_ROW_ROR4 0x124
#define DPP_ROW_ROR8 0x128
#define DPP_ROW_ROR15 0x12F   // lane i reads lane (i + 1) & 15

// maximum of u over the row (invalid lanes excluded)
__device__ __forceinline__ double rowmax16(double u, bool valid) {
    double m = valid ? u : -__builtin_huge_val();
    double o;
    o = dpp_f64<DPP_QUAD_XOR1>(m); m = o > m ? o : m;
    o = dpp_f64<DPP_QUAD_XOR2>(m); m = o > m ? o : m;
    o = dpp_f64<DPP_ROW_ROR4>(m); m = o > m ? o : m;
    o = dpp_f64<DPP_ROW_ROR8>(m); m = o > m ? o : m;
    return m;
}
__device__ __forceinline__ int rowmin16(int c) {
    int t;
    t = dpp_i32<DPP_QUAD_XOR1>(c); c = t < c ? t : c;
    t = dpp_i32<DPP_QUAD_XOR2>(c); c = t < c ? t : c;
    t = dpp_i32<DPP_ROW_ROR4>(c); c = t < c ? t : c;
    t = dpp_i32<DPP_ROW_ROR8>(c); c = t < c ? t : c;
    return c;
}
// lane index (0..15) of the maximum over the row, lowest lane on ties (the reference breaks ties randomly, helpers.py:46-52)
__device__ __forceinline__ int argmax16(double u, bool valid, int sub) {
    double m = rowmax16(u, valid);
    return rowmin16((valid && u == m) ? sub : 99);
}
// 32-bit butterflies over the row: the compiler folds each DPP move into the min/max (v_max_i32_dpp ...), one VALU op per stage
__device__ __forceinline__ int rowmax16_i32(int v) {
    int o;
    o = dpp_i32<DPP_QUAD_XOR1>(v); v = o > v ? o : v;
    o = dpp_i32<DPP_QUAD_XOR2>(v); v = o > v ? o : v;
    o = dpp_i32<DPP_ROW_ROR4>(v); v = o > v ? o : v;
    o = dpp_i32<DPP_ROW_ROR8>(v); v = o > v ? o : v;
    return v;
}
__device__ __forceinline__ unsigned rowmax16_u32(unsigned v) {
    unsigned o;
    o = (unsigned)dpp_i32<DPP_QUAD_XOR1>((int)v); v = o > v ? o : v;
    o = (unsigned)dpp_i32<DPP_QUAD_XOR2>((int)v); v = o > v ? o : v;
    o = (unsigned)dpp_i32<DPP_ROW_ROR4>((int)v); v = o > v ? o : v;
    o = (unsigned)dpp_i32<DPP_ROW_ROR8>((int)v); v = o > v ? o : v;
    return v;
}
// arg-max over the row that returns the payload (< 65536) of the winning lane, lowest lane on ties.  The float64 scores are
// compared as order-preserving 64-bit integer keys, high word first (three 4-stage 32-bit butterflies instead of a 64-bit
// compare-and-select per stage).  Scores are finite and never -0 (Q + c * ratio with ratio > 0), so key order == float order.
__device__ __forceinline__ int argmax16_payload(double u, bool valid, int sub, int payload) {
    const long long bits = (long long)azg_d2u(u);
    const long long key = bits ^ ((bits >> 63) & 0x7fffffffffffffffLL);
    const int hi = valid ? (int)(key >> 32) : (int)0x80000000;
    const unsigned lo = (unsigned)key;
    const int mh = rowmax16_i32(hi);
    const bool top = valid && hi == mh;
    const unsigned ml = rowmax16_u32(top ? lo : 0u);
    return rowmin16((top && lo == ml) ? ((sub << 16) | payload) : 0x7fffffff) & 0xffff;
}

// two candidates only (lanes 0 and 1 of the row; CartPole has two actions): one compare instead of a 16-lane butterfly.
// Returns the winner's payload in every lane of the row, lane 0 on ties.
#define DPP_QUAD_BCAST0 0x00   // quad_perm:[0,0,0,0]
__device__ __forceinline__ int argmax2_payload(double u, int sub, int payload) {
    const double o = dpp_f64<DPP_QUAD_XOR1>(u);
    const int po = dpp_i32<DPP_QUAD_XOR1>(payload);
    int w = (sub == 0) ? (u >= o ? payload : po) : 0;   // lane 0 decides: first index unless the second is strictly larger
    w = dpp_i32<DPP_QUAD_BCAST0>(w);                     // lanes 0..3 hold the winner (a record id > 0), the other quads 0
    int t = dpp_i32<DPP_ROW_ROR4>(w); w = t > w ? t : w; // spread by max over the rotations (whichever way they turn)
    t = dpp_i32<DPP_ROW_ROR8>(w); w = t > w ? t : w;
    return w;
}

// storage of the hot part of one tree: LDS (RecS, 8-bit ids, pooled child lists) or global memory (RecL, 16-bit ids,
// a [Kp]-wide child table per record).  child_at / child_append are the continuous-mode child list accessors; the
// arguments of child_append are uniform over the tree's 16 lanes and only `writer` (lane 0) stores.
template <int TLDS> struct TreeStore;
// LDS-resident trees: RecS + byte pool (TS_LDS8) or RecM + 16-bit pool (TS_LDS9), same logic
template <typename RecT, typename IdT> struct TreeStoreLds {
    typedef RecT Rec;
    typedef IdT PoolId;
    Rec* hot; IdT* pool; float* prior;
    // discrete mode, when the CU's LDS has room (search_kernel.cuh: lds_layout): the env state (4 doubles) of every expanded,
    // non-terminal node, slot (first - 1) / A of the node whose child edges start at record `first` -- a node gets its A edges
    // when it is evaluated, so the e-th evaluated node of a tree (the root is the 0-th) owns slot e; a node that is being created
    // while the tree holds nrec records will be the ((nrec - 1) / A)-th.  nullptr: the states are read from the cold records.
    double* state;
    __device__ __forceinline__ int child_at(int, const Rec& hp, int i, int) const {
        return hp.n_child == 1 ? (int)hp.first : (int)pool[4 * (int)hp.cbase + i];
    }
    // (wnext: the parent's widens-at-next-visit bit with its K + 1 children, stored with the new child count)
    __device__ __forceinline__ void child_append(int p, const Rec& hp, int K, int id, int& ptop, bool writer, int, bool wnext = false) const {
        const int cb = hp.cbase;
        if (K == 0) {
            if (writer) hot[p].first = (decltype(hp.first))id;
        } else if (K == 1 || K == 4 || K == 8) {
            const int nb = ptop;
            ptop += K == 1 ? 1 : K / 2;   // blocks of 4, 8, 16 ids
            if (writer) {
                if (K == 1) pool[4 * nb] = (IdT)hp.first;
                else for (int w = 0; w < K; ++w) pool[4 * nb + w] = pool[4 * cb + w];
                pool[4 * nb + K] = (IdT)id;
                hot[p].cbase = (decltype(hp.cbase))nb;
            }
        } else {
            if (writer) pool[4 * cb + K] = (IdT)id;
        }
        if (writer) { hot[p].n_child = (decltype(hp.n_child))(K + 1); hot[p].wnext = wnext ? 1 : 0; }
    }
};
template <> struct TreeStore<TS_LDS8> : TreeStoreLds<RecS, unsigned char> {};
template <> struct TreeStore<TS_LDS9> : TreeStoreLds<RecM, unsigned short> {};
template <> struct TreeStore<TS_GLOBAL> {
    typedef RecL Rec;
    Rec* hot; unsigned short* child; float* prior;
    static constexpr double* state = nullptr;   // (global trees keep env states in their cold records)
    __device__ __forceinline__ int child_at(int p, const Rec&, int i, int Kp) const { return (int)child[p * Kp + i]; }
    __device__ __forceinline__ void child_append(int p, const Rec&, int K, int id, int&, bool writer, int Kp, bool wnext = false) const {
        if (writer) {
            child[p * Kp + K] = (unsigned short)id;
            hot[p].n_child = (unsigned short)(K + 1);
            hot[p].pad = wnext ? 1 : 0;
        }
    }
};

// Continuous mode: "this node widens at its next visit" -- NodeContinuous.check_pw (states.py:271-275: ceil(c_pw (n + 1)^kappa) > number of
// children) evaluated whenever the node's visit count or child count changes (backup, widening, creation) and kept with the node, so that
// the descent reads it with the record instead of looking the threshold up behind it (one LDS round trip per level of the descent).
__device__ __forceinline__ bool get_wnext(const RecS& h) { return h.wnext != 0; }
__device__ __forceinline__ bool get_wnext(const RecM& h) { return h.wnext != 0; }
__device__ __forceinline__ bool get_wnext(const RecL& h) { return (h.pad & 1) != 0; }
__device__ __forceinline__ void set_wnext(RecS& h, bool w) { h.wnext = w ? 1 : 0; }
__device__ __forceinline__ void set_wnext(RecM& h, bool w) { h.wnext = w ? 1 : 0; }
__device__ __forceinline__ void set_wnext(RecL& h, bool w) { h.pad = w ? 1 : 0; }
// the widening threshold of a node with n visits (the host-built table is clamped at n_sims + 1: records.h KParams::pw_need)
template <typename PW>
__device__ __forceinline__ int pw_at(const PW* s_pw, int n, int cap) { return (int)s_pw[n < cap ? n : cap]; }

template <typename Rec>
__device__ __forceinline__ Rec make_edge(double Q, int parent) {
    Rec h;
    h.Q = Q; h.edge_n = 0; h.node_n = 0; h.parent = (decltype(h.parent))parent; h.n_child = 0; h.flags = 0; h.first = 0;
    set_wnext(h, false);
    return h;
}
__device__ __forceinline__ void clear_pad(RecS& h) { h.cbase = 0; }
__device__ __forceinline__ void clear_pad(RecM& h) { h.cbase = 0; }
__device__ __forceinline__ void clear_pad(RecL& h) { h.pad = 0; }

// MCTS.backprop (mcts.py:260-267), generic part: walks parent links from record j to the root, 16 levels at a time
// (lane d = d-th record), fetches rewards / W in parallel, chains the discounted return serially (its rounding order
// is part of the contract), then every lane updates its own record.
// on_node(p): called (by all 16 lanes, p uniform) for every node whose statistics were updated here, after the update is stored
// (the caller of backup_path looks after the nodes of the last 16 levels itself).
// (continuous mode: s_pw / pw_cap = the widening thresholds, for the nodes' widens-at-next-visit bits: set_wnext)
template <bool CONT, int TLDS, typename F, typename PW = int>
__device__ __forceinline__ void backup_from(const TreeStore<TLDS>& ts, const Cold* cold, double* edge_W, int j, float V, int sub,
                                            float gamma_f, double gamma, bool firstlvl, bool at_leaf, double Rv, F&& on_node,
                                            const PW* s_pw = nullptr, int pw_cap = 0) {
    typedef typename TreeStore<TLDS>::Rec Rec;
    while (true) {
        int mine = 0, cnt = 0, jj = j;
        bool hit_root = false;
#pragma unroll 1
        for (int d = 0; d < 16; ++d) {
            if (sub == d) mine = jj;
            cnt = d + 1;
            if (jj == 0) { hit_root = true; break; }
            jj = ts.hot[jj].parent;
        }
        Rec mrec = ts.hot[mine];   // (read once the lane knows its record: a record carried through the loop lives in scratch memory)
        const bool is_edge = (sub < cnt) && (mine != 0);
        double r = 0.0, W = 0.0;
        if (is_edge) { r = cold[mine].r; W = edge_W[mine]; }
        double myR = 0.0;
        const int nedge = hit_root ? cnt - 1 : cnt;
#pragma unroll 1
        for (int d = 0; d < nedge; ++d) {
            double rd = __shfl(r, d, 16);
            double gR;
            if (firstlvl) {
                // continuous: V is a float32 0-d array and gamma a python scalar -> float32 product (NumPy >= 2);
                // discrete: V is a python float -> float64 product
                gR = CONT ? (double)(gamma_f * V) : gamma * (double)V;
                firstlvl = false;
            } else {
                gR = gamma * Rv;
            }
            Rv = rd + gR;
            if (sub == d) myR = Rv;
        }
        if (sub < cnt) {
            if (is_edge) {
                int en = (int)mrec.edge_n + 1;
                double Wn = W + myR;
                mrec.Q = tree_div(Wn, (double)en);
                mrec.edge_n = (decltype(mrec.edge_n))en;
                edge_W[mine] = Wn;
            }
            if (!(at_leaf && sub == 0)) {
                mrec.node_n = (decltype(mrec.node_n))(mrec.node_n + 1);
                if constexpr (CONT) set_wnext(mrec, pw_at(s_pw, (int)mrec.node_n, pw_cap) - (int)mrec.n_child > 0);
            }
            ts.hot[mine] = mrec;
        }
        if (!TLDS) tree_fence();
        for (int d = 0; d < cnt; ++d) {
            const int pn = __shfl(mine, d, 16);
            if (!(at_leaf && d == 0)) on_node(pn);   // (a trace's leaf has no selection to refresh)
        }
        if (hit_root) break;
        j = jj;
        at_leaf = false;
    }
}

// Backup of a trace whose path the descent left in the lanes: slot (depth & 15) holds the record id, its reward and W
// (fetched while descending), so nothing is loaded from global memory here.  Paths deeper than 16 finish in backup_from.
// On return pW holds the slot's updated W and `rec` the slot's updated record (zeros in lanes without a slot); `chainR` is the
// lane's return of this trace.  same_chain: the trace is the previous one again (same path, same terminal leaf, see
// tree_phase_b): every lane's return is the one it had, the serial chain is skipped.  r_uniform (discrete mode): the reward every
// edge of the path carries (env.cuh: discrete_env_reward) except the last one, the edge into the leaf, which carries r_first (the
// same value unless the leaf is terminal and the env pays differently on its last step: Acrobot).
template <bool CONT, int TLDS, typename F, typename PW = int>
__device__ __forceinline__ void backup_path(const TreeStore<TLDS>& ts, const Cold* cold, double* edge_W, float V, int sub, float gamma_f,
                                            double gamma, int D, int my_depth, int pid, double pr, double& pW, F&& on_node,
                                            typename TreeStore<TLDS>::Rec& rec, double& chainR, bool same_chain = false,
                                            double r_uniform = 0.0, double r_first = 0.0, const PW* s_pw = nullptr, int pw_cap = 0) {
    typedef typename TreeStore<TLDS>::Rec Rec;
    const int n0 = D < 16 ? D : 16;
    double Rv = 0.0, myR = 0.0;
    // the return travels down the path one lane per step on the DPP network: slot (depth & 15) reads its deeper neighbour's
    // value (lane sub + 1, the slot of depth + 1) and adds its own reward -- the same serial chain, no LDS shuffles
    if (same_chain) {
        myR = chainR;
    } else if constexpr (!CONT) {
        // discrete mode: every edge of the path carries the same reward (r_uniform), so the chain R_0 = r + gamma V, R_d = r + gamma R_{d-1}
        // needs nothing from the other lanes: every lane runs it by itself (two dependent float64 operations per level instead of
        // a DPP round trip) and keeps the value of its own slot -- the same operations in the same order as the travelling form
        double R = 0.0;
        if (gamma == 1.0) {
            // gamma * x is x itself, bit for bit, for gamma == 1 (the reference's default): one operation per level instead of two
            // ... and with the discrete envs' rewards (+1, -1 or 0 per step) every partial sum V + r_first + r_uniform + ... is exactly
            // representable whenever V is 0 or 2^-25 <= |V| < 2^30 (V is a float32: 24 significant bits, the integers added stay below 2^5),
            // so the chain's value at level d IS V + (r_first + d r_uniform), whatever the order: one addition per lane instead of a
            // serial chain of up to 16 (config B: -300 cycles per trace).  Any other V takes the chain.
            const float av = __builtin_fabsf(V);
            if (av == 0.0f || (av >= 0x1p-25f && av < 0x1p30f)) {
                const int dl = (D - sub) & 15;                  // the level whose return this lane's slot takes
                if (dl < n0) myR = (r_first + (double)dl * r_uniform) + (double)V;
                R = n0 > 0 ? (r_first + (double)(n0 - 1) * r_uniform) + (double)V : (double)V;
            } else {
                R = (double)V;
#pragma unroll 2
                for (int d = 0; d < n0; ++d) {
                    R = (d == 0 ? r_first : r_uniform) + R;
                    if (sub == ((D - d) & 15)) myR = R;
                }
            }
        } else {
#pragma unroll 2
            for (int d = 0; d < n0; ++d) {
                const double gR = d == 0 ? gamma * (double)V : gamma * R;
                R = (d == 0 ? r_first : r_uniform) + gR;
                if (sub == ((D - d) & 15)) myR = R;
            }
        }
        Rv = R;            // (D >= 16: the return of the shallowest of the 16 levels, handed to backup_from)
        chainR = myR;
    } else {
#pragma unroll 1
        for (int d = 0; d < n0; ++d) {
            const int src = (D - d) & 15;
            const double nb = dpp_f64<DPP_ROW_ROR15>(myR);
            const double gR = d == 0 ? (CONT ? (double)(gamma_f * V) : gamma * (double)V) : gamma * nb;
            if (sub == src) myR = pr + gR;
        }
        chainR = myR;
    }
    if (CONT && D >= 16) Rv = __shfl(myR, (D - 15) & 15, 16);
    const bool valid = my_depth >= 0 && my_depth > D - 16;
    int par = 0;
    rec = make_edge<Rec>(0.0, 0);
    clear_pad(rec);
    if (valid) {
        rec = ts.hot[pid];
        // (continuous mode: the node's widening threshold at its new count, requested ahead of the division below)
        int need = 0;
        if constexpr (CONT) { if (my_depth < D) need = pw_at(s_pw, (int)rec.node_n + 1, pw_cap); }
        par = rec.parent;
        if (my_depth >= 1) {
            int en = (int)rec.edge_n + 1;
            double Wn = pW + myR;
            rec.Q = tree_div(Wn, (double)en);
            rec.edge_n = (decltype(rec.edge_n))en;
            edge_W[pid] = Wn;
            pW = Wn;
        }
        if (my_depth < D) {
            rec.node_n = (decltype(rec.node_n))(rec.node_n + 1);
            if constexpr (CONT) set_wnext(rec, need - (int)rec.n_child > 0);
        }
        ts.hot[pid] = rec;
    }
    if (D >= 16) {
        int j = __shfl(par, (D - 15) & 15, 16);   // parent of the shallowest record handled above
        backup_from<CONT, TLDS>(ts, cold, edge_W, j, V, sub, gamma_f, gamma, false, false, Rv, on_node, s_pw, pw_cap);
    }
}
