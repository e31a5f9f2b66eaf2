// aux_kernels.cuh -- result gathering, the device-resident self-play step, and the math / MFMA self tests.
#pragma once
#include "records.h"
#include "env.cuh"
#include "tree.cuh"
#include "results.cuh"

// ------------------------------------------------------------------------------------------------ result gathering

// The root's child edges of one tree, staged per thread: every record (and action) is fetched by an independent load in an
// unrolled loop -- one global round trip for all of them instead of one per child -- and then kept in LDS, where the
// order-sensitive sums below can index it dynamically.  Trees with more than RK_MAX root children read global memory directly.
#define RK_MAX 16
#define RK_THREADS 64
struct RootKids {
    RecL rec[RK_MAX];
    float act[RK_MAX];
    int id[RK_MAX];
};
__device__ __forceinline__ void root_kids_load(const KParams& P, size_t tb, const RecL& root, bool cont, RootKids* k) {
    const RecL* hot = P.hot + tb;
    const unsigned short* child = P.child + tb * P.Kp;
    const int nc = root.n_child;
    int id[RK_MAX];
#pragma unroll
    for (int a = 0; a < RK_MAX; ++a) id[a] = a < nc ? (cont ? (int)child[a] : (int)root.first + a) : 0;
    RecL r[RK_MAX];
    float ac[RK_MAX];
#pragma unroll
    for (int a = 0; a < RK_MAX; ++a) { r[a] = hot[id[a]]; ac[a] = cont ? P.action[tb + id[a]] : (float)a; }
#pragma unroll
    for (int a = 0; a < RK_MAX; ++a) { k->rec[a] = r[a]; k->act[a] = ac[a]; k->id[a] = a < nc ? id[a] : -1; }
}
// child a of the root: from the staged copy, or from global memory for roots with more than RK_MAX children
struct RootView {
    const KParams& P; size_t tb; const RecL& root; bool cont; const RootKids* k; bool staged;
    __device__ __forceinline__ int id(int a) const {
        if (staged) return k->id[a];
        return a < (int)root.n_child ? (cont ? (int)P.child[tb * P.Kp + a] : (int)root.first + a) : -1;
    }
    __device__ __forceinline__ RecL rec(int a) const { if (staged) return k->rec[a]; int i = id(a); return P.hot[tb + (i >= 0 ? i : 0)]; }
    __device__ __forceinline__ float act(int a) const { if (staged) return k->act[a]; int i = id(a); return cont ? P.action[tb + (i >= 0 ? i : 0)] : (float)a; }
};

// The same for all trees from their published (global) form: 16 trees per workgroup.
#define RS_TREES 16
__global__ __launch_bounds__(16 * RS_TREES) void results_kernel(KParams P) {
    const int sub = threadIdx.x & 15;
    const int tree = blockIdx.x * RS_TREES + (threadIdx.x >> 4);
    if (tree >= P.B) return;
    const size_t tb = (size_t)tree * P.R;
    TreeStore<TS_GLOBAL> ts;
    ts.hot = P.hot + tb;
    ts.child = P.child + tb * P.Kp;
    ts.prior = P.prior + tb;
    if (P.mode == AZG_MODE_CONTINUOUS) results_for_tree<true, TS_GLOBAL>(P, ts, P.cold + tb, P.action + tb, tb, tree, sub);
    else results_for_tree<false, TS_GLOBAL>(P, ts, P.cold + tb, P.action + tb, tb, tree, sub);
}

// One self-play step after a search, one thread per game: replay row, the agent's final action rule, the real env step,
// episode bookkeeping and the next search's root (the CPU oracle restates the same arithmetic for the parity tests).
struct SelfPlay {
    int max_len, deterministic;
    int final_selection;      // AZG_FS_*
    double agent_eps;         // ContinuousAgent.epsilon
    const double* ctab;       // (c / m)^temperature at [m (m + 1) / 2 + c], 0 <= c <= m <= n_sims, built by the host (NULL: temperature 1)
    unsigned step_idx;
    int* t; int* episode; int* fcnt;
    double* ret; double* fsum;
    float* rows;          // this step's block [B][row_len]
    double* roots; int* carry;
};

// The common case (at most 16 root children: every LDS-tree configuration): 16 lanes per game, lane a = root child a.  The replay
// row is written by the lanes side by side; totals are row reductions that do not depend on the order (integer sum, maximum,
// first index of the largest count); everything whose float64 order matters (on-policy target, normaliser sums, the inverse-CDF
// walk of numpy's random.choice) is added up by the game's first lane in the reference's order, reading the lanes' values by
// shuffles; that lane also steps the env and keeps the episode's books.  Same arithmetic as selfplay_kernel below (roots with
// more children) and as the oracle.
#define SP_TREES 16
__global__ __launch_bounds__(16 * SP_TREES) void selfplay_kernel16(KParams P, SelfPlay sp, int Kmax, int v_target, int env_id, int S_obs) {
    const int sub = threadIdx.x & 15;
    const int tree = blockIdx.x * SP_TREES + (threadIdx.x >> 4);
    if (tree >= P.B) return;
    const bool cont = P.mode == AZG_MODE_CONTINUOUS;
    const unsigned gtree = (unsigned)(P.tree_base + tree);
    const int S = P.S, K = Kmax, RL = S_obs + 3 * Kmax + 1;
    float* row = sp.rows + (size_t)tree * RL;
    // The root's children as MCTS.return_results left them (written by the search kernel's epilogue from its LDS-resident trees, or
    // by results_kernel after the lock-step / team kernels): no tree record is read here, so a search need not publish its trees.
    const int nc = P.res_nch[tree];
    // lane a: root child a
    const bool has = sub < nc;
    struct { int edge_n, node_n, flags; double Q; } h;
    {
        const size_t o = (size_t)tree * Kmax + (sub < Kmax ? sub : 0);
        const int cn = P.res_child_n[o];                   // the child node's visit count, -1: the edge has no child node yet
        h.edge_n = has ? P.res_counts[o] : 0;
        h.Q = has ? P.res_Q[o] : 0.0;
        h.node_n = (has && cn >= 0) ? cn : 0;
        h.flags = (has && cn >= 0) ? FLAG_EXPANDED : 0;
    }
    const float act = has ? (cont ? P.res_actions[(size_t)tree * Kmax + sub] : (float)sub) : 0.0f;
    if (sub < K) {
        row[S_obs + sub] = act;
        row[S_obs + K + sub] = has ? (float)h.edge_n : 0.0f;
        row[S_obs + 2 * K + sub] = has ? (float)h.Q : 0.0f;
    }
    int tot = has ? h.edge_n : 0;
    double qmax = has ? h.Q : -__builtin_huge_val();
    int ckey = has ? ((h.edge_n << 4) | (15 - sub)) : -1;          // largest count, lowest index on ties (counts < 2^27)
    for (int m = 1; m < 16; m <<= 1) {
        tot += __shfl_xor(tot, m, 16);
        const double o = __shfl_xor(qmax, m, 16);
        qmax = o > qmax ? o : qmax;
        const int ok = __shfl_xor(ckey, m, 16);
        ckey = ok > ckey ? ok : ckey;
    }
    if (nc == 0) qmax = 0.0;
    const int cmax = ckey >> 4, amax = 15 - (ckey & 15);
    // discrete: the lane's unnormalised pi entry (stable_normalizer's x / max(x), to the temperature)
    double x = 0.0;
    if (!cont && has) {
        if (sp.final_selection == AZG_FS_MAX_VALUE) x = h.Q / qmax;
        else x = sp.ctab ? sp.ctab[(size_t)cmax * (cmax + 1) / 2 + h.edge_n] : (double)h.edge_n / (double)cmax;
    }
    // ---- the game's first lane: order-sensitive sums, the final action, the env step, the books (the shuffles are executed by
    // the whole row: loop bounds are row-uniform)
    double onp = 0.0;
    if (v_target == AZG_VT_ON_POLICY) {
        if (!cont) {
            for (int a = 0; a < nc; ++a) onp += ((double)__shfl(h.edge_n, a, 16) / (double)tot) * __shfl(h.Q, a, 16);
        } else {
            for (int a = 0; a < nc; ++a) {
                const double qa = __shfl(h.Q, a, 16);
                for (int b2 = 0; b2 < nc; ++b2) onp += ((double)__shfl(h.edge_n, b2, 16) / (double)tot) * qa;
            }
        }
    }
    int pick = 0;
    if (cont) {
        // ContinuousAgent.act (agents.py:524-535): actions[Qs.argmax()] / actions[counts.argmax()], first index on ties
        pick = amax;
        if (sp.final_selection == AZG_FS_MAX_VALUE) {
            double qb = 0.0;
            for (int a = 0; a < nc; ++a) { const double q = __shfl(h.Q, a, 16); if (a == 0 || q > qb) { qb = q; pick = a; } }
        }
        if (sp.agent_eps != 0.0) {
            // epsilon_greedy (agents.py:471-490): random.random() < epsilon -> np.random.choice(actions)
            azg_u32x4 b = azg_draw(P.seed, gtree, sp.step_idx, 0u, AZG_STREAM_ACT);
            if ((double)azg_u01(b.v[0]) < sp.agent_eps) pick = (int)(b.v[1] % (unsigned)nc);
        }
    } else {
        // DiscreteAgent.act (agents.py:294-301): pi = stable_normalizer(Qs | counts, temperature) (helpers.py:26-27), then
        // pi.argmax() or np.random.choice(len(pi), p=pi) (cdf = cumsum(pi); cdf /= cdf[-1]; first index with u < cdf)
        double sum = 0.0;
        for (int a = 0; a < nc; ++a) sum = sum + __shfl(x, a, 16);
        const double pi = has ? __builtin_fabs(x / sum) : 0.0;
        double best = 0.0, cum = 0.0;
        for (int a = 0; a < nc; ++a) {
            const double pa = __shfl(pi, a, 16);
            if (a == 0 || pa > best) { best = pa; pick = a; }
            cum = cum + pa;
        }
        if (!sp.deterministic) {
            azg_u32x4 b = azg_draw(P.seed, gtree, sp.step_idx, 0u, AZG_STREAM_ACT);
            const double u = ((double)b.v[0] + 0.5) * (1.0 / 4294967296.0);
            const double last = cum;
            double c = 0.0;
            pick = nc - 1;
            bool found = false;
            for (int a = 0; a < nc; ++a) {
                const double pa = __shfl(pi, a, 16);
                if (!found) {
                    c = c + pa;
                    if (u < c / last) { pick = a; found = true; }
                }
            }
        }
    }
    const float pact = __shfl(act, pick, 16);
    const int pnode_n = __shfl(h.node_n, pick, 16), pflags = __shfl((int)h.flags, pick, 16);
    if (sub != 0) return;
    double root[4] = {0.0, 0.0, 0.0, 0.0};
    for (int k = 0; k < S; ++k) root[k] = sp.roots[(size_t)tree * S + k];
    float obs[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    double sn = 0.0;
    // (the MountainCars' observation is (position, velocity): the discrete family's state as float32 with S_obs = 2; Acrobot: six)
    if (!cont) discrete_env_obs(env_id, root, obs);
    else if (env_id == AZG_ENV_MOUNTAINCAR_CONT) env_obs<AZG_ENV_CARTPOLE>(root, obs, &sn);
    else env_obs<AZG_ENV_PENDULUM_V1>(root, obs, &sn);
    for (int k = 0; k < 8; ++k) if (k < S_obs) row[k] = obs[k];
    row[S_obs + 3 * K] = (float)(v_target == AZG_VT_ON_POLICY ? onp : qmax);
    double ns[4] = {0.0, 0.0, 0.0, 0.0}, r;
    int done;
    if (!cont && env_id == AZG_ENV_ACROBOT) azg_acrobot_step(root, pick, ns, &r, &done);
    else if (!cont) discrete_env_step(env_id, root, pick, ns, &r, &done);
    else if (env_id == AZG_ENV_MOUNTAINCAR_CONT) mountaincar_cont_step(root, pact, ns, &r, &done);
    else pendulum_step(env_id == AZG_ENV_PENDULUM_V1, root, sn, pact, ns, &r, &done);
    double ret = sp.ret[tree] + r;
    int t = sp.t[tree] + 1;
    if (done || t >= sp.max_len) {
        sp.fsum[tree] = sp.fsum[tree] + ret;
        sp.fcnt[tree] += 1;
        ret = 0.0;
        t = 0;
        int ep = sp.episode[tree] + 1;
        sp.episode[tree] = ep;
        azg_reset_state(P.seed, gtree, (unsigned)ep, azg_reset_kind(env_id), ns);
        sp.carry[tree] = 0;
    } else {
        sp.carry[tree] = (!cont && (pflags & FLAG_EXPANDED)) ? pnode_n : 0;
    }
    sp.ret[tree] = ret;
    sp.t[tree] = t;
    for (int k = 0; k < S; ++k) sp.roots[(size_t)tree * S + k] = ns[k];
}

// Roots with more than 16 children (global trees of long continuous searches): one thread per game.
__global__ __launch_bounds__(RK_THREADS) void selfplay_kernel(KParams P, SelfPlay sp, int Kmax, int v_target, int env_id, int S_obs) {
    __shared__ RootKids s_kids[RK_THREADS];
    int tree = blockIdx.x * blockDim.x + threadIdx.x;
    if (tree >= P.B) return;
    const size_t tb = (size_t)tree * P.R;
    const RecL* hot = P.hot + tb;
    const bool cont = P.mode == AZG_MODE_CONTINUOUS;
    const unsigned gtree = (unsigned)(P.tree_base + tree);
    const int S = P.S, K = Kmax, RL = S_obs + 3 * Kmax + 1;
    double root[4] = {0.0, 0.0, 0.0, 0.0};
    for (int k = 0; k < S; ++k) root[k] = sp.roots[(size_t)tree * S + k];
    float* row = sp.rows + (size_t)tree * RL;
    const RecL r0 = hot[0];
    const int nc = r0.n_child;
    const bool staged = Kmax <= RK_MAX;
    if (staged) root_kids_load(P, tb, r0, cont, &s_kids[threadIdx.x]);
    const RootView rv{P, tb, r0, cont, &s_kids[threadIdx.x], staged};
    float obs[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    double sn = 0.0;
    if (!cont) discrete_env_obs(env_id, root, obs);
    else if (env_id == AZG_ENV_MOUNTAINCAR_CONT) env_obs<AZG_ENV_CARTPOLE>(root, obs, &sn);
    else env_obs<AZG_ENV_PENDULUM_V1>(root, obs, &sn);
    for (int k = 0; k < 8; ++k) if (k < S_obs) row[k] = obs[k];
    double qmax = 0.0, onp = 0.0;
    long tot = 0;
    int cmax = 0, amax = 0;
    for (int a = 0; a < nc; ++a) tot += rv.rec(a).edge_n;
    for (int a = 0; a < K; ++a) {
        int k = a < nc ? rv.id(a) : -1;
        RecL h = rv.rec(a < nc ? a : 0);
        row[S_obs + a] = k >= 0 ? rv.act(a) : 0.0f;
        row[S_obs + K + a] = k >= 0 ? (float)h.edge_n : 0.0f;
        row[S_obs + 2 * K + a] = k >= 0 ? (float)h.Q : 0.0f;
        if (k >= 0) {
            if (a == 0 || h.Q > qmax) qmax = h.Q;
            if (!cont && v_target == AZG_VT_ON_POLICY) onp += ((double)h.edge_n / (double)tot) * h.Q;
            if (a == 0 || h.edge_n > cmax) { cmax = h.edge_n; amax = a; }
        }
    }
    if (cont && v_target == AZG_VT_ON_POLICY)
        for (int a = 0; a < nc; ++a)
            for (int b = 0; b < nc; ++b) onp += ((double)rv.rec(b).edge_n / (double)tot) * rv.rec(a).Q;
    row[S_obs + 3 * K] = (float)(v_target == AZG_VT_ON_POLICY ? onp : qmax);
    int pick = 0;
    if (cont) {
        // ContinuousAgent.act (agents.py:524-535): actions[Qs.argmax()] / actions[counts.argmax()], first index on ties
        pick = amax;
        if (sp.final_selection == AZG_FS_MAX_VALUE) {
            double qb = 0.0;
            for (int a = 0; a < nc; ++a) { const double q = rv.rec(a).Q; if (a == 0 || q > qb) { qb = q; pick = a; } }
        }
        if (sp.agent_eps != 0.0) {
            // epsilon_greedy (agents.py:471-490): random.random() < epsilon -> np.random.choice(actions)
            azg_u32x4 b = azg_draw(P.seed, gtree, sp.step_idx, 0u, AZG_STREAM_ACT);
            if ((double)azg_u01(b.v[0]) < sp.agent_eps) pick = (int)(b.v[1] % (unsigned)nc);
        }
    } else {
        // DiscreteAgent.act (agents.py:294-301): pi = stable_normalizer(Qs | counts, temperature) (helpers.py:26-27), then
        // pi.argmax() or np.random.choice(len(pi), p=pi) (cdf = cumsum(pi); cdf /= cdf[-1]; first index with u < cdf)
        double x[RK_MAX];
        double sum = 0.0;
        for (int a = 0; a < RK_MAX; ++a) {
            x[a] = 0.0;
            if (a < nc) {
                const RecL h = rv.rec(a);
                if (sp.final_selection == AZG_FS_MAX_VALUE) x[a] = h.Q / qmax;
                else x[a] = sp.ctab ? sp.ctab[(size_t)cmax * (cmax + 1) / 2 + h.edge_n] : (double)h.edge_n / (double)cmax;
                sum = sum + x[a];
            }
        }
        double best = 0.0, cum = 0.0;
        for (int a = 0; a < RK_MAX; ++a)
            if (a < nc) {
                x[a] = __builtin_fabs(x[a] / sum);
                if (a == 0 || x[a] > best) { best = x[a]; pick = a; }
                cum = cum + x[a];
            }
        if (!sp.deterministic) {
            azg_u32x4 b = azg_draw(P.seed, gtree, sp.step_idx, 0u, AZG_STREAM_ACT);
            const double u = ((double)b.v[0] + 0.5) * (1.0 / 4294967296.0);
            const double last = cum;
            double c = 0.0;
            pick = nc - 1;
            bool found = false;
            for (int a = 0; a < RK_MAX; ++a)
                if (a < nc && !found) {
                    c = c + x[a];
                    if (u < c / last) { pick = a; found = true; }
                }
        }
    }
    double ns[4] = {0.0, 0.0, 0.0, 0.0}, r;
    int done;
    if (!cont && env_id == AZG_ENV_ACROBOT) {
        azg_acrobot_step(root, pick, ns, &r, &done);
    } else if (!cont) {
        discrete_env_step(env_id, root, pick, ns, &r, &done);
    } else if (env_id == AZG_ENV_MOUNTAINCAR_CONT) {
        mountaincar_cont_step(root, rv.act(pick), ns, &r, &done);
    } else {
        double s1, c1;
        azg_sincos(root[0], &s1, &c1);
        pendulum_step(env_id == AZG_ENV_PENDULUM_V1, root, s1, rv.act(pick), ns, &r, &done);
    }
    double ret = sp.ret[tree] + r;
    int t = sp.t[tree] + 1;
    if (done || t >= sp.max_len) {
        sp.fsum[tree] = sp.fsum[tree] + ret;
        sp.fcnt[tree] += 1;
        ret = 0.0;
        t = 0;
        int ep = sp.episode[tree] + 1;
        sp.episode[tree] = ep;
        azg_reset_state(P.seed, gtree, (unsigned)ep, azg_reset_kind(env_id), ns);
        sp.carry[tree] = 0;
    } else {
        RecL hk = rv.rec(pick);
        sp.carry[tree] = (!cont && (hk.flags & FLAG_EXPANDED)) ? hk.node_n : 0;
    }
    sp.ret[tree] = ret;
    sp.t[tree] = t;
    for (int k = 0; k < S; ++k) sp.roots[(size_t)tree * S + k] = ns[k];
}

__global__ void math_selftest_kernel(int fn_id, const double* in, double* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = in[i], s, c;
    switch (fn_id) {
        case 0: out[i] = (double)azg_expf((float)x); break;
        case 1: out[i] = (double)azg_expm1f((float)x); break;
        case 2: out[i] = (double)azg_tanhf((float)x); break;
        case 3: out[i] = (double)azg_logf((float)x); break;
        case 4: out[i] = (double)azg_cos2pif((float)x); break;
        case 5: azg_sincos(x, &s, &c); out[i] = s; break;
        case 6: azg_sincos(x, &s, &c); out[i] = c; break;
        case 7: out[i] = azg_pymod(x, 2.0 * 3.141592653589793, 0.15915494309189535); break;
        case 8: out[i] = (double)azg_normal(34u, (uint32_t)x, 0u, (uint32_t)(x * 7.0)); break;
        case 9: out[i] = (double)((float)x / 3.0f); break;
        case 10: out[i] = (double)__builtin_sqrtf((float)x); break;
        case 11: out[i] = x / 3.0; break;
        case 12: out[i] = __builtin_sqrt(x); break;
        default: out[i] = 0.0;
    }
}

// fn_id 100: one 16x16x4 MFMA chain over n/… ; in = [a0..a(K-1), b0..b(K-1), c], out[0] = D[0][0]; probes the accumulation order
__global__ void mfma_probe_kernel(const double* in, double* out, int K) {
    int lane = threadIdx.x;
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    float c = (float)in[2 * K];
    acc.x = acc.y = acc.z = acc.w = c;
    for (int s = 0; s < K / 4; ++s) {
        int k = 4 * s + (lane >> 4);
        float a = (float)in[k];
        float b = (float)in[K + k];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    if (lane == 0) out[0] = (double)acc.x;
}

// fn_id 101: matrix-pipe rate probe (diagnostic): every wave issues iters x 16 register-only fp32 MFMAs (4 independent chains);
// out[0..1] of wave 0: shader-clock cycles (s_memtime) and constant 100 MHz ticks (s_memrealtime) across the loop.
__global__ __launch_bounds__(256) void mfma_rate_kernel(double* out, int iters) {
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float a = 1.0f + 1e-7f * threadIdx.x, b = 1.0f - 1e-7f * threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = acc[0].x + acc[1].y + acc[2].z + acc[3].w;
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = (double)(t1 - t0); out[1] = (double)(r1 - r0); }
    if (sum == 123.456f) out[2] = sum;   // keeps the chains alive
}

// fn_id 102: latency probe (diagnostic): one wave, dependent chains of N operations each; out[i] = shader cycles per operation.
//   0 v_fma_f64   1 v_fma_f32   2 v_mul_f64 + v_add_f64   3 float64 division (tree_div)   4 IEEE float64 division
//   5 LDS round trip (ds_read_b32, dependent address)   6 LDS 16-byte round trip   7 DPP move (dependent)   8 ds_bpermute
//   9 global load round trip (dependent address, L2-resident)   10 azg_sincos   11 v_mfma_f32_16x16x4 dependent chain
__global__ __launch_bounds__(64) void latency_probe_kernel(double* out, int* chase, int n) {
    __shared__ int s_chase[1024];
    __shared__ f32x4 s_wide[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) s_chase[i] = (i * 17 + 5) & 1023;
    for (int i = lane; i < 256; i += 64) s_wide[i] = f32x4{(float)((i * 7 + 3) & 255), 0.0f, 0.0f, 0.0f};
    __syncthreads();
    unsigned long long t0, t1;
    double res[12];
#define PROBE(idx, init, body, sink)                                              \
    {                                                                             \
        init;                                                                     \
        __builtin_amdgcn_s_waitcnt(0);                                            \
        t0 = __builtin_amdgcn_s_memtime();                                        \
        __builtin_amdgcn_sched_barrier(0);                                        \
        _Pragma("unroll 1") for (int it = 0; it < n; it += 8) { body; body; body; body; body; body; body; body; }   \
        sink;                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                        \
        t1 = __builtin_amdgcn_s_memtime();                                        \
        res[idx] = (double)(t1 - t0) / n;                                         \
    }
    double xd = 1.0 + 1e-9 * lane, ad = 0.999999, bd = 1e-7, accd = 0.0;
    float xf = 1.0f + 1e-6f * lane;
    PROBE(0, , xd = __builtin_fma(xd, ad, bd), accd += xd)
    PROBE(1, , xf = __builtin_fmaf(xf, 0.99999f, 1e-6f), accd += xf)
    PROBE(2, , xd = xd * ad + bd, accd += xd)
    PROBE(3, , xd = tree_div(xd + 2.0, 1.5), accd += xd)
    PROBE(4, , xd = (xd + 2.0) / 1.5, accd += xd)
    int p = lane;
    PROBE(5, , p = s_chase[p], accd += p)
    int pw = lane;
    PROBE(6, , pw = (int)s_wide[pw & 255].x, accd += pw)
    int dv = lane;
    PROBE(7, , dv = __builtin_amdgcn_update_dpp(0, dv, 0x121, 0xf, 0xf, false) + 1, accd += dv)
    int bv = lane;
    PROBE(8, , bv = __builtin_amdgcn_ds_bpermute(((bv + 1) & 63) << 2, bv), accd += bv)
    int g = lane;
    PROBE(9, , g = chase[g], accd += g)
    double sn = 0.1 * lane, cs = 0.0;
    PROBE(10, , azg_sincos(sn + 0.5, &sn, &cs), accd += sn + cs)
    f32x4 ma = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    PROBE(11, , ma = __builtin_amdgcn_mfma_f32_16x16x4f32(xf, 1.0f, ma, 0, 0, 0), accd += ma.x)
#undef PROBE
    if (lane == 0) for (int i = 0; i < 12; ++i) out[i] = res[i];
    if (accd == 1.2345) out[12] = accd;
}
