// azg_engine.hip -- MI355X (gfx950) batched MCTS engine behind the C ABI of include/azgym.h.
//
// One persistent kernel launch runs a whole MCTS search (all n_sims simulations) for B independent trees.
//   * workgroup = 256 threads = 4 waves = one group of 16 trees (= one 16-row MFMA tile of leaf evaluations);
//     a workgroup never talks to another one, so there is no grid-wide synchronisation anywhere.
//   * tree phase: a 16-lane sub-wave owns one tree.  Lanes scan the <=16 children of a node in parallel
//     (PUCT / progressive-widening UCT, float64), arg-max by a 4-step butterfly, step the closed-form
//     environment, expand, and back the return up the path.
//   * network phase: the policy/value MLP for the 16 new leaves on v_mfma_f32_16x16x4_f32.  Activations are
//     kept transposed ([unit][tree]) so that an MFMA's D registers are the next layer's B operand as they
//     stand; hidden->hidden weights can live in the 512-entry VGPR/AGPR file for the whole search (NREG>0).
// Reference semantics: alphazero/search/mcts.py (search 418-462 / 656-702, selectionUCT 464-493 / 704-741,
// backprop 241-267, return_results 269-307), alphazero/search/states.py, alphazero/network/policies.py.
// The arithmetic (operation order, float32/float64 placement) is specified by oracle/azg_oracle.c, which is pinned
// to the reference by tests/golden; this file must agree with it bit for bit.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/azg_math.h"
#include "../../include/azgym.h"

#define FLAG_EXPANDED 1
#define FLAG_TERMINAL 2
#define TREES_PER_WG 16
#define MAX_STREAM_LAYERS 8

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Record j of a tree = edge j + (once expanded) the node that edge leads to; record 0 is the root.
// "Hot" part: everything selection and the count/Q side of backup touch.  Two encodings:
//   RecS (16 B) lives in LDS for the whole search when the tree fits (<= 255 records, counts < 65536),
//   RecL (24 B) lives in global memory (any size); it is also the format trees are published in at the end of a search.
struct __attribute__((aligned(16))) RecS {
    double Q;                // edge action value (Q_init = parent V)
    unsigned short edge_n;   // edge visit count
    unsigned short node_n;   // node visit count
    unsigned char parent;    // record of the parent node
    unsigned char n_child;   // node: number of child edges
    unsigned char flags;     // FLAG_EXPANDED | FLAG_TERMINAL
    unsigned char first;     // discrete: record of child edge 0 (children are contiguous)
};
struct __attribute__((aligned(8))) RecL {
    double Q;
    int edge_n;
    int node_n;
    short parent;
    unsigned short n_child;
    unsigned short first;
    unsigned char flags;
    unsigned char pad;
};
static_assert(sizeof(RecS) == 16, "RecS must be 16 bytes");
static_assert(sizeof(RecL) == 24, "RecL must be 24 bytes");

// "Cold" part of a node (global memory): read once when a child is expanded from it or an action is sampled at it
struct __attribute__((aligned(16))) Cold {
    double s[4];   // env state (Pendulum: theta, theta_dot, sin(theta) cached, unused)
    double r;      // reward on arriving here (already divided by reward_scale in continuous mode)
    float V;       // value estimate
    float mu;      // continuous: cached squashed-Normal mean
    float sg;      //             and standard deviation
    float pad;
};
static_assert(sizeof(Cold) == 64, "Cold must be 64 bytes");

struct KParams {
    int B, n_sims, R, Kp, A, nd, n_out, n_hidden, act, v1, tree_base, mode;
    double c_uct, gamma, epsilon, reward_scale;
    float c_uct_f, gamma_f, bound_f, ls_min, ls_max;
    unsigned long long seed;
    unsigned search_idx;
    int S;                   // env state dim
    int tab_n;               // entries in sqrt_tab
    const double* roots;     // [B][S]
    const int* carry;        // [B]
    RecL* hot;               // [B][R]      published trees (and working storage when the tree does not fit LDS)
    Cold* cold;              // [B][R]
    double* edge_W;          // [B][R]      edge cumulative return
    float* action;           // [B][R]      continuous: edge action
    float* prior;            // [B][R]      discrete: edge prior
    float* gmm;              // [B][R][15]  continuous mixture head: mu[5] | sigma[5] | cumulative mixture probability[5]
    int ncomp;               // C (0: squashed Normal)
    unsigned short* child;   // [B][R][Kp]  continuous: child record ids of a node, in creation order
    int* n_rec;              // [B]
    const int* pw_need;      // [n_sims+2]
    const double* sqrt_tab;  // [tab_n]  sqrt(n+1)
    const float* W0;         // [HP/16][64]
    const f32x4* b0;         // [HP/16][64]
    const f32x4* Wl[MAX_STREAM_LAYERS]; // hidden->hidden layer l (1-based index l-1): [HP/16 tiles][HP/16 s4][64]
    const f32x4* bl[MAX_STREAM_LAYERS]; // [HP/16][64]
    const f32x4* Whead;      // [HP/16 s4][64]
    const float* bhead;      // [16]
    unsigned long long* stamps; // diagnostic build only (-DAZG_STAMPS): [grid][8] cycle sums per phase
};

#ifdef AZG_STAMPS
#define STAMP(var) unsigned long long var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0)
#define STAMP_ADD(slot, t0, t1) st_acc[slot] += (t1) - (t0)
#else
#define STAMP(var)
#define STAMP_ADD(slot, t0, t1)
#endif

// ------------------------------------------------------------------------------------------------ environments

// observation of a state; Pendulum also returns sin(theta) so that the node can cache it for its children's dynamics
template <int ENV>
__device__ __forceinline__ void env_obs(const double* s, float* obs, double* sn_out) {
    if (ENV == AZG_ENV_CARTPOLE) {
        obs[0] = (float)s[0]; obs[1] = (float)s[1]; obs[2] = (float)s[2]; obs[3] = (float)s[3];
        *sn_out = 0.0;
    } else {
        double sn, cs;
        azg_sincos(s[0], &sn, &cs);
        obs[0] = (float)cs; obs[1] = (float)sn; obs[2] = (float)s[1]; obs[3] = 0.0f;
        *sn_out = sn;
    }
}

// gym CartPoleEnv.step (explicit Euler); same operation order as oracle/azg_oracle.c cartpole_step
__device__ __forceinline__ void cartpole_step(const double* s, int action, double* o, double* reward, int* done) {
    const double gravity = 9.8, masspole = 0.1, total_mass = 0.1 + 1.0, length = 0.5;
    const double polemass_length = 0.1 * 0.5, force_mag = 10.0, tau = 0.02;
    const double theta_thr = 12.0 * 2.0 * 3.141592653589793 / 360.0, x_thr = 2.4;
    double x = s[0], x_dot = s[1], theta = s[2], theta_dot = s[3];
    double force = action == 1 ? force_mag : -force_mag;
    double sintheta, costheta;
    azg_sincos(theta, &sintheta, &costheta);
    double temp = (force + (polemass_length * (theta_dot * theta_dot)) * sintheta) / total_mass;
    double thetaacc = (gravity * sintheta - costheta * temp) /
                      (length * (4.0 / 3.0 - (masspole * (costheta * costheta)) / total_mass));
    double xacc = temp - ((polemass_length * thetaacc) * costheta) / total_mass;
    x = x + tau * x_dot;
    x_dot = x_dot + tau * xacc;
    theta = theta + tau * theta_dot;
    theta_dot = theta_dot + tau * thetaacc;
    o[0] = x; o[1] = x_dot; o[2] = theta; o[3] = theta_dot;
    *done = (x < -x_thr) || (x > x_thr) || (theta < -theta_thr) || (theta > theta_thr);
    *reward = 1.0;
}

// gym PendulumEnv.step; v1: speed clipped before integrating theta, v0: after.  sn_th = sin(theta), cached in the node
__device__ __forceinline__ void pendulum_step(int v1, const double* s, double sn_th, float action, double* o, double* reward, int* done) {
    const double max_speed = 8.0, dt = 0.05, pi = 3.141592653589793;
    const float max_torque = 2.0f;
    double th = s[0], thdot = s[1];
    float uc = action < -max_torque ? -max_torque : (action > max_torque ? max_torque : action);
    double u = (double)uc;
    double an = azg_pymod(th + pi, 2.0 * pi, 0.15915494309189535) - pi;
    double costs = (an * an + 0.1 * (thdot * thdot)) + 0.001 * (u * u);
    double newth, newthdot, sn, cs;
    if (v1) {
        newthdot = thdot + (15.0 * sn_th + 3.0 * u) * dt;
        newthdot = newthdot < -max_speed ? -max_speed : (newthdot > max_speed ? max_speed : newthdot);
        newth = th + newthdot * dt;
    } else {
        azg_sincos(th + pi, &sn, &cs);
        newthdot = thdot + (-15.0 * sn + 3.0 * u) * dt;
        newth = th + newthdot * dt;
        newthdot = newthdot < -max_speed ? -max_speed : (newthdot > max_speed ? max_speed : newthdot);
    }
    o[0] = newth; o[1] = newthdot;
    *reward = -costs;
    *done = 0;
}

// ------------------------------------------------------------------------------------------------ MLP on MFMA

typedef int i32x4 __attribute__((ext_vector_type(4)));

// azg_expm1f on four values at once: the same operations in the same order per component (bit-identical), written
// component-parallel so that the four dependent fma chains interleave (and pack into v_pk_fma_f32)
__device__ __forceinline__ f32x4 expm1f4_nonpos(f32x4 x) {
    // x <= 0 here, so only the lower clamp of azg_expm1f can trigger
    const f32x4 lo = {-87.0f, -87.0f, -87.0f, -87.0f};
    f32x4 xc = __builtin_elementwise_max(x, lo);
    const f32x4 magic = {12582912.0f, 12582912.0f, 12582912.0f, 12582912.0f};
    const f32x4 l2e = {1.44269504088896341f, 1.44269504088896341f, 1.44269504088896341f, 1.44269504088896341f};
    const f32x4 ln2h = {0.693145751953125f, 0.693145751953125f, 0.693145751953125f, 0.693145751953125f};
    const f32x4 ln2l = {1.42860682030941723212e-6f, 1.42860682030941723212e-6f, 1.42860682030941723212e-6f, 1.42860682030941723212e-6f};
    f32x4 kf = __builtin_elementwise_fma(xc, l2e, magic);
    kf = kf - magic;
    f32x4 r = __builtin_elementwise_fma(-kf, ln2h, xc);
    r = __builtin_elementwise_fma(-kf, ln2l, r);
    f32x4 p = {1.98412698412698413e-4f, 1.98412698412698413e-4f, 1.98412698412698413e-4f, 1.98412698412698413e-4f};
    const f32x4 c5 = {1.38888888888888894e-3f, 1.38888888888888894e-3f, 1.38888888888888894e-3f, 1.38888888888888894e-3f};
    const f32x4 c4 = {8.33333333333333322e-3f, 8.33333333333333322e-3f, 8.33333333333333322e-3f, 8.33333333333333322e-3f};
    const f32x4 c3 = {4.16666666666666644e-2f, 4.16666666666666644e-2f, 4.16666666666666644e-2f, 4.16666666666666644e-2f};
    const f32x4 c2 = {1.66666666666666657e-1f, 1.66666666666666657e-1f, 1.66666666666666657e-1f, 1.66666666666666657e-1f};
    const f32x4 half = {0.5f, 0.5f, 0.5f, 0.5f}, one = {1.0f, 1.0f, 1.0f, 1.0f};
    p = __builtin_elementwise_fma(p, r, c5);
    p = __builtin_elementwise_fma(p, r, c4);
    p = __builtin_elementwise_fma(p, r, c3);
    p = __builtin_elementwise_fma(p, r, c2);
    p = __builtin_elementwise_fma(p, r, half);
    f32x4 em1 = __builtin_elementwise_fma(p * r, r, r);
    i32x4 k = __builtin_convertvector(kf, i32x4);
    f32x4 sc = (f32x4)((k + 127) << 23);
    return __builtin_elementwise_fma(sc, em1, sc - one);
}

// ELU without a branch: max(x,0) + expm1(min(x,0)); expm1(+-0) == +0 exactly, so this equals `x > 0 ? x : expm1(x)` bit
// for bit (the sign of a zero that v_max/v_min may pick differently from the host's select vanishes in the sum)
__device__ __forceinline__ f32x4 act4(int act, f32x4 v) {
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 pos = __builtin_elementwise_max(v, zero);
    if (act == AZG_ACT_ELU) {
        f32x4 neg = __builtin_elementwise_min(v, zero);
        return pos + expm1f4_nonpos(neg);
    }
    return pos;
}

__device__ __forceinline__ f32x4 mfma4(f32x4 a, f32x4 b, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
    return acc;
}

// Register-resident hidden->hidden weights: wave w owns output tiles [w*NTW, (w+1)*NTW) of each layer.
template <int HP, int NREG>
struct WRegs {
    static constexpr int NTW = HP / 64;
    static constexpr int S4 = HP / 16;
    f32x4 w[NREG > 0 ? NREG : 1][NTW][S4];
    f32x4 b[NREG > 0 ? NREG : 1][NTW];
    f32x4 wh[NTW];   // head weights of this wave's K-chunk
    float w0[NTW];   // first layer (K <= 4: one k-step per tile)
    f32x4 b0[NTW];
};

// The MLP for the workgroup's 16 leaves.  obsT: [4][16] (input feature k, tree).  Result: parts[4 waves][64 lanes] = every
// wave's partial head sums (head_output() combines them).  Activations cross waves through the two act buffers
// (HP/16 tiles x 64 lanes x float4 = the D registers of each 16x16 output tile as they stand); a layer's output stays in
// registers until the next layer publishes it, and the last layer's output feeds the head MFMAs directly.
template <int HP, int NREG>
__device__ __forceinline__ void mlp_forward(const KParams& P, const WRegs<HP, NREG>& wr, const float* obsT, f32x4* actA, f32x4* actB,
                                            f32x4* parts, int wave, int lane
#ifdef AZG_STAMPS
                                            , unsigned long long* st_acc
#endif
                                            ) {
    STAMP(m0);
    constexpr int NTW = HP / 64;   // output tiles per wave
    constexpr int S4 = HP / 16;    // groups of 4 MFMA k-steps over a hidden vector
    f32x4 h[NTW];                  // this wave's tiles of the latest layer, after the activation
    // layer 0: K = in_dim <= 4 -> one k-step
    {
        float b = obsT[lane];
#pragma unroll
        for (int i = 0; i < NTW; ++i) h[i] = act4(P.act, __builtin_amdgcn_mfma_f32_16x16x4f32(wr.w0[i], b, wr.b0[i], 0, 0, 0));
    }
    f32x4* buf = actA;
    f32x4* other = actB;
    // hidden->hidden layers held in registers
    if (NREG > 0) {
#pragma unroll
        for (int l = 0; l < NREG; ++l) {
#pragma unroll
            for (int i = 0; i < NTW; ++i) buf[(wave * NTW + i) * 64 + lane] = h[i];
            __syncthreads();
            STAMP(m1);
            f32x4 acc[NTW];
#pragma unroll
            for (int i = 0; i < NTW; ++i) acc[i] = wr.b[l][i];
            f32x4 bcur = buf[lane];
#pragma unroll
            for (int s4 = 0; s4 < S4; ++s4) {
                f32x4 bnext = bcur;
                if (s4 + 1 < S4) bnext = buf[(s4 + 1) * 64 + lane];   // prefetch the next 4 k-steps' B operand
                __builtin_amdgcn_sched_barrier(0);                    // keep the ds_read above this block's MFMAs
                // k-step outer, tile inner: consecutive MFMAs are independent chains (40-cycle dependent latency)
#pragma unroll
                for (int i = 0; i < NTW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr.w[l][i][s4].x, bcur.x, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < NTW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr.w[l][i][s4].y, bcur.y, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < NTW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr.w[l][i][s4].z, bcur.z, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < NTW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr.w[l][i][s4].w, bcur.w, acc[i], 0, 0, 0);
                bcur = bnext;
            }
            STAMP(m2);
#pragma unroll
            for (int i = 0; i < NTW; ++i) h[i] = act4(P.act, acc[i]);
            STAMP(m2b);
            if (l == 0) { STAMP_ADD(4, m0, m1); }
            STAMP_ADD(5, m1, m2);
            STAMP_ADD(6, m2, m2b);
            f32x4* t = buf; buf = other; other = t;
        }
    } else {
        // weights streamed from global memory (L2-resident), any number of layers
        for (int l = 1; l < P.n_hidden; ++l) {
#pragma unroll
            for (int i = 0; i < NTW; ++i) buf[(wave * NTW + i) * 64 + lane] = h[i];
            __syncthreads();
            const f32x4* W = P.Wl[l - 1];
            const f32x4* bb = P.bl[l - 1];
            f32x4 acc[NTW];
#pragma unroll
            for (int i = 0; i < NTW; ++i) acc[i] = bb[(wave * NTW + i) * 64 + lane];
#pragma unroll 2
            for (int s4 = 0; s4 < S4; ++s4) {
                f32x4 b = buf[s4 * 64 + lane];
                f32x4 a[NTW];
#pragma unroll
                for (int i = 0; i < NTW; ++i) a[i] = W[((wave * NTW + i) * S4 + s4) * 64 + lane];
#pragma unroll
                for (int i = 0; i < NTW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b.x, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < NTW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b.y, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < NTW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b.z, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < NTW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b.w, acc[i], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NTW; ++i) h[i] = act4(P.act, acc[i]);
            f32x4* t = buf; buf = other; other = t;
        }
    }
    // heads: wave w sums its quarter of the hidden units (chain from 0) straight from its registers
    {
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            f32x4 a = (NREG > 0) ? wr.wh[i] : P.Whead[(wave * NTW + i) * 64 + lane];
            acc = mfma4(a, h[i], acc);
        }
        parts[wave * 64 + lane] = acc;
    }
    __syncthreads();
}

// network output o of tree tl: bias + the four waves' partial sums, added in wave order (the oracle's summation order)
__device__ __forceinline__ float head_output(const f32x4* parts, const float* s_bhead, int tl, int o) {
    float total = s_bhead[o];
    const int idx = (o >> 2) * 16 + tl;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        f32x4 pv = parts[w * 64 + idx];
        float p = (o & 3) == 0 ? pv.x : ((o & 3) == 1 ? pv.y : ((o & 3) == 2 ? pv.z : pv.w));
        total = total + p;
    }
    return total;
}

#define GMM_MAXC 5
// DiagonalGMMPolicy head (policies.py:544-560) of one node from the raw network outputs: mu_c, sigma_c = exp(clamp(log_std_c)),
// cumulative softmax(log_coeff) in component order.  d[15] = mu[5] | sigma[5] | cum[5] (fixed stride so that every index
// below is a compile-time constant and the arrays stay in registers).
__device__ __forceinline__ void gmm_params(const f32x4* parts, const float* s_bhead, int tl, int C, float ls_min, float ls_max, float* d) {
    float mx = head_output(parts, s_bhead, tl, 1 + 2 * C);
#pragma unroll
    for (int c = 1; c < GMM_MAXC; ++c)
        if (c < C) { float v = head_output(parts, s_bhead, tl, 1 + 2 * C + c); mx = v > mx ? v : mx; }
    float ex[GMM_MAXC], sum = 0.0f, cum = 0.0f;
#pragma unroll
    for (int c = 0; c < GMM_MAXC; ++c) {
        ex[c] = 0.0f;
        if (c < C) { ex[c] = azg_expf(head_output(parts, s_bhead, tl, 1 + 2 * C + c) - mx); sum = sum + ex[c]; }
    }
#pragma unroll
    for (int c = 0; c < GMM_MAXC; ++c) {
        d[c] = 0.0f; d[GMM_MAXC + c] = 0.0f; d[2 * GMM_MAXC + c] = 2.0f;
        if (c < C) {
            float ls = head_output(parts, s_bhead, tl, 1 + C + c);
            ls = ls < ls_min ? ls_min : (ls > ls_max ? ls_max : ls);
            d[c] = head_output(parts, s_bhead, tl, 1 + c);
            d[GMM_MAXC + c] = azg_expf(ls);
            cum = cum + ex[c] / sum;
            d[2 * GMM_MAXC + c] = cum;
        }
    }
}

// MixtureSameFamily.sample (policies.py:656-668): component by inverse CDF with the third word of the widening draw
__device__ __forceinline__ void gmm_pick(const float* d, int C, unsigned long long seed, unsigned gtree, unsigned search, unsigned k,
                                         float* mu, float* sg) {
    azg_u32x4 b = azg_draw(seed, gtree, search, k, AZG_STREAM_PW);
    float u = azg_u01(b.v[2]);
    float m = 0.0f, s = 0.0f;
    bool found = false;
#pragma unroll
    for (int i = 0; i < GMM_MAXC; ++i) {
        bool last = (i == C - 1);
        if (i < C && !found && (u < d[2 * GMM_MAXC + i] || last)) { m = d[i]; s = d[GMM_MAXC + i]; found = true; }
    }
    *mu = m;
    *sg = s;
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
#define DPP_QUAD_XOR1 0xB1   // quad_perm:[1,0,3,2]
#define DPP_QUAD_XOR2 0x4E   // quad_perm:[2,3,0,1]
#define DPP_ROW_ROR4 0x124
#define DPP_ROW_ROR8 0x128

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
// same, but returns the payload (< 65536) of the winning lane: no second cross-lane round trip
__device__ __forceinline__ int argmax16_payload(double u, bool valid, int sub, int payload) {
    double m = rowmax16(u, valid);
    return rowmin16((valid && u == m) ? ((sub << 16) | payload) : 0x7fffffff) & 0xffff;
}

// storage of the hot part of one tree: LDS (RecS, 8-bit ids) or global memory (RecL, 16-bit ids)
template <bool TLDS> struct TreeStore;
template <> struct TreeStore<true> {
    typedef RecS Rec;
    typedef unsigned char Id;
    Rec* hot; Id* child; float* prior;
};
template <> struct TreeStore<false> {
    typedef RecL Rec;
    typedef unsigned short Id;
    Rec* hot; Id* child; float* prior;
};

template <typename Rec>
__device__ __forceinline__ Rec make_edge(double Q, int parent) {
    Rec h;
    h.Q = Q; h.edge_n = 0; h.node_n = 0; h.parent = (decltype(h.parent))parent; h.n_child = 0; h.flags = 0; h.first = 0;
    return h;
}
__device__ __forceinline__ void clear_pad(RecS&) {}
__device__ __forceinline__ void clear_pad(RecL& h) { h.pad = 0; }

// MCTS.backprop (mcts.py:260-267), generic part: walks parent links from record j to the root, 16 levels at a time
// (lane d = d-th record), fetches rewards / W in parallel, chains the discounted return serially (its rounding order
// is part of the contract), then every lane updates its own record.
template <bool CONT, bool TLDS>
__device__ __forceinline__ void backup_from(const TreeStore<TLDS>& ts, const Cold* cold, double* edge_W, int j, float V, int sub,
                                            float gamma_f, double gamma, bool firstlvl, bool at_leaf, double Rv) {
    typedef typename TreeStore<TLDS>::Rec Rec;
    while (true) {
        int mine = 0, cnt = 0, jj = j;
        bool hit_root = false;
        Rec mrec = ts.hot[jj];
#pragma unroll 1
        for (int d = 0; d < 16; ++d) {
            Rec rr = ts.hot[jj];
            if (sub == d) { mine = jj; mrec = rr; }
            cnt = d + 1;
            if (jj == 0) { hit_root = true; break; }
            jj = rr.parent;
        }
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
                mrec.Q = Wn / (double)en;
                mrec.edge_n = (decltype(mrec.edge_n))en;
                edge_W[mine] = Wn;
            }
            if (!(at_leaf && sub == 0)) mrec.node_n = (decltype(mrec.node_n))(mrec.node_n + 1);
            ts.hot[mine] = mrec;
        }
        if (hit_root) break;
        j = jj;
        at_leaf = false;
    }
}

// Backup of a trace whose path the descent left in the lanes: slot (depth & 15) holds the record id, its reward and W
// (fetched while descending), so nothing is loaded from global memory here.  Paths deeper than 16 finish in backup_from.
template <bool CONT, bool TLDS>
__device__ __forceinline__ void backup_path(const TreeStore<TLDS>& ts, const Cold* cold, double* edge_W, float V, int sub, float gamma_f,
                                            double gamma, int D, int my_depth, int pid, double pr, double pW) {
    typedef typename TreeStore<TLDS>::Rec Rec;
    const int n0 = D < 16 ? D : 16;
    double Rv = 0.0, myR = 0.0;
#pragma unroll 1
    for (int d = 0; d < n0; ++d) {
        const int src = (D - d) & 15;
        double rd = __shfl(pr, src, 16);
        double gR = d == 0 ? (CONT ? (double)(gamma_f * V) : gamma * (double)V) : gamma * Rv;
        Rv = rd + gR;
        if (sub == src) myR = Rv;
    }
    const bool valid = my_depth >= 0 && my_depth > D - 16;
    int par = 0;
    if (valid) {
        Rec rec = ts.hot[pid];
        par = rec.parent;
        if (my_depth >= 1) {
            int en = (int)rec.edge_n + 1;
            double Wn = pW + myR;
            rec.Q = Wn / (double)en;
            rec.edge_n = (decltype(rec.edge_n))en;
            edge_W[pid] = Wn;
        }
        if (my_depth < D) rec.node_n = (decltype(rec.node_n))(rec.node_n + 1);
        ts.hot[pid] = rec;
    }
    if (D >= 16) {
        int j = __shfl(par, (D - 15) & 15, 16);   // parent of the shallowest record handled above
        backup_from<CONT, TLDS>(ts, cold, edge_W, j, V, sub, gamma_f, gamma, false, false, Rv);
    }
}

template <int ENV, int HP, int NREG, bool TLDS, bool GMM>
__global__ __launch_bounds__(256, 1) void search_kernel(KParams P) {
    constexpr bool CONT = (ENV != AZG_ENV_CARTPOLE);
    constexpr int S = CONT ? 2 : 4;
    typedef typename TreeStore<TLDS>::Rec Rec;
    typedef typename TreeStore<TLDS>::Id Id;
    __shared__ f32x4 s_parts[4 * 64];
    __shared__ float s_obsT[4 * 16];
    __shared__ float s_bhead[16];
    extern __shared__ double s_dyn[];   // sqrt_tab [tab_n], pw_need [n_sims+2] ints, two activation buffers, (TLDS) the 16 trees' hot records

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int sub = lane & 15;
    const int tl = wave * 4 + (lane >> 4);          // tree within the workgroup
    const int tree = blockIdx.x * TREES_PER_WG + tl;
    const bool live = tree < P.B;
    const unsigned gtree = (unsigned)(P.tree_base + tree);

    double* s_sqrt = s_dyn;
    int* s_pw = (int*)(s_dyn + P.tab_n);
    const size_t act_off = ((size_t)P.tab_n * 8 + (size_t)(P.n_sims + 2) * 4 + 15) / 16 * 16;
    f32x4* s_actA = (f32x4*)((char*)s_dyn + act_off);
    f32x4* s_actB = s_actA + HP / 16 * 64;
    for (int i = tid; i < P.tab_n; i += 256) s_sqrt[i] = P.sqrt_tab[i];
    if (CONT) for (int i = tid; i < P.n_sims + 2; i += 256) s_pw[i] = P.pw_need[i];
    if (tid < 16) s_bhead[tid] = P.bhead[tid];

    // register-resident weights
    WRegs<HP, NREG> wr;
    {
        constexpr int NTW0 = HP / 64;
#pragma unroll
        for (int i = 0; i < NTW0; ++i) {
            wr.w0[i] = P.W0[(wave * NTW0 + i) * 64 + lane];
            wr.b0[i] = P.b0[(wave * NTW0 + i) * 64 + lane];
        }
    }
    if (NREG > 0) {
        constexpr int NTW = HP / 64, S4 = HP / 16;
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

    const size_t tb = (size_t)(live ? tree : 0) * P.R;
    Cold* cold = P.cold + tb;
    double* edge_W = P.edge_W + tb;
    float* action = P.action + tb;
    TreeStore<TLDS> ts;
    if (TLDS) {
        // per tree: R records of 16 B, then (continuous) R x Kp child ids or (discrete) R priors
        size_t off = act_off + (size_t)2 * HP * 64;
        size_t per = (size_t)P.R * 16 + (CONT ? (size_t)P.R * P.Kp : (size_t)P.R * 4);
        per = (per + 15) / 16 * 16;
        char* base = (char*)s_dyn + off + per * tl;
        ts.hot = (Rec*)base;
        ts.child = (Id*)(base + (size_t)P.R * 16);
        ts.prior = (float*)(base + (size_t)P.R * 16);
    } else {
        ts.hot = (Rec*)(P.hot + tb);
        ts.child = (Id*)(P.child + tb * P.Kp);
        ts.prior = P.prior + tb;
    }

#ifdef AZG_STAMPS
    unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    int nrec = 1;
    unsigned eps_draws = 0;
    int leaf = 0;
    bool need_eval = live;
    // the current trace's path, one record per lane (slot = depth & 15): id, reward, W -- consumed by backup_path
    int path_D = 0, my_depth = -1, pid = 0;
    double pr = 0.0, pW = 0.0;
    // progressive-widening noise: lane `sub` holds the N(0,1) draw for record kbase + sub
    int kbase = 1;
    float eps_c = 0.0f;
    if (CONT && live) eps_c = azg_normal(P.seed, gtree, P.search_idx, (unsigned)(kbase + sub));

    // ---- root (initialize_search + evaluation / add_value_estimate: mcts.py:364-383, 437; 589-600, 672)
    {
        double rs[S], sn;
#pragma unroll
        for (int k = 0; k < S; ++k) rs[k] = live ? P.roots[(size_t)tree * S + k] : 0.0;
        float obs[4];
        env_obs<ENV>(rs, obs, &sn);
        if (live && sub == 0) {
            Rec h = make_edge<Rec>(0.0, 0);
            h.node_n = (decltype(h.node_n))P.carry[tree];
            h.flags = FLAG_EXPANDED;
            clear_pad(h);
            ts.hot[0] = h;
            Cold c;
#pragma unroll
            for (int k = 0; k < 4; ++k) c.s[k] = k < S ? rs[k] : 0.0;
            if (CONT) c.s[2] = sn;
            c.r = 0.0; c.V = 0.0f; c.mu = 0.0f; c.sg = 0.0f; c.pad = 0.0f;
            cold[0] = c;
            edge_W[0] = 0.0;
            if (CONT) action[0] = 0.0f;
        }
        if (sub < 4) s_obsT[sub * 16 + tl] = live ? obs[sub] : 0.0f;
    }
    __syncthreads();

    for (int sim = -1; sim < P.n_sims; ++sim) {
        // ================= network phase: evaluate the 16 pending leaves =================
        STAMP(t_a);
        int any = __syncthreads_or(need_eval ? 1 : 0);
        STAMP(t_b);
#ifdef AZG_STAMPS
        if (any) mlp_forward<HP, NREG>(P, wr, s_obsT, s_actA, s_actB, s_parts, wave, lane, st_acc);
#else
        if (any) mlp_forward<HP, NREG>(P, wr, s_obsT, s_actA, s_actB, s_parts, wave, lane);
#endif
        STAMP(t_c);

        // ================= tree phase A: finish the evaluated leaf, back up =================
        if (live) {
            float V = 0.0f;
            if (need_eval) {
                V = head_output(s_parts, s_bhead, tl, 0);
                if (CONT) {
                    float mu, sg;
                    float gd[15];
                    if constexpr (GMM) {
                        gmm_params(s_parts, s_bhead, tl, P.ncomp, P.ls_min, P.ls_max, gd);
                        mu = gd[0]; sg = gd[GMM_MAXC];
                        float* g = P.gmm + (tb + leaf) * 3 * GMM_MAXC;
                        if (sub == 0) {
#pragma unroll
                            for (int i = 0; i < 3 * GMM_MAXC; ++i) g[i] = gd[i];
                        }
                    } else {
                        mu = head_output(s_parts, s_bhead, tl, 1);
                        float ls = head_output(s_parts, s_bhead, tl, 2);
                        ls = ls < P.ls_min ? P.ls_min : (ls > P.ls_max ? P.ls_max : ls);
                        sg = azg_expf(ls);
                    }
                    if (sub == 0) { cold[leaf].V = V; cold[leaf].mu = mu; cold[leaf].sg = sg; }
                    if (sim < 0) {
                        // add_pw_action(root) before the first trace (mcts.py:673)
                        int k = nrec++;
                        if constexpr (GMM) gmm_pick(gd, P.ncomp, P.seed, gtree, P.search_idx, (unsigned)k, &mu, &sg);
                        float eps = __shfl(eps_c, k - kbase, 16);
                        float a = P.bound_f * azg_tanhf(mu + sg * eps);
                        if (sub == 0) {
                            Rec h = make_edge<Rec>((double)V, 0);
                            clear_pad(h);
                            ts.hot[k] = h;
                            edge_W[k] = 0.0;
                            action[k] = a;
                            ts.child[0] = (Id)k;
                            ts.hot[0].n_child = 1;
                        }
                    }
                } else {
                    // softmax priors + all num_actions edges with Q_init = V (MCTSDiscrete.evaluation, mcts.py:412-416)
                    const int A = P.A;
                    float mx = head_output(s_parts, s_bhead, tl, 1);
                    for (int a = 1; a < A; ++a) { float v = head_output(s_parts, s_bhead, tl, 1 + a); mx = v > mx ? v : mx; }
                    float sum = 0.0f;
                    for (int a = 0; a < A; ++a) sum = sum + azg_expf(head_output(s_parts, s_bhead, tl, 1 + a) - mx);
                    int k0 = nrec;
                    nrec += A;
                    if (sub < A) {
                        float pr = azg_expf(head_output(s_parts, s_bhead, tl, 1 + sub) - mx) / sum;
                        Rec h = make_edge<Rec>((double)V, leaf);
                        clear_pad(h);
                        ts.hot[k0 + sub] = h;
                        ts.prior[k0 + sub] = pr;
                        edge_W[k0 + sub] = 0.0;
                    }
                    if (sub == 0) {
                        cold[leaf].V = V;
                        ts.hot[leaf].n_child = (decltype(ts.hot[leaf].n_child))A;
                        ts.hot[leaf].first = (decltype(ts.hot[leaf].first))k0;
                    }
                }
            }
            STAMP(t_c2);
            STAMP_ADD(8, t_c, t_c2);    // finish leaf (before backup)
            if (sim >= 0) {
                if (!TLDS) __threadfence_block();   // lane 0's partial record stores above must land before the path is re-read
                backup_path<CONT, TLDS>(ts, cold, edge_W, V, sub, P.gamma_f, P.gamma, path_D, my_depth, pid, pr, pW);
            }
        }
        if (sim == P.n_sims - 1) break;
        __threadfence_block();
        STAMP(t_d);

        // ================= tree phase B: next trace: select down, step the env, expand =================
        // The descent loop contains only UCT levels, so the four trees of a wave run the same code and differ only in
        // trip count; widening and expansion happen once, after the loop, for all four trees together.
        need_eval = false;
        if (live) {
            if (CONT && nrec >= kbase + 16) {
                kbase = nrec;
                eps_c = azg_normal(P.seed, gtree, P.search_idx, (unsigned)(kbase + sub));
            }
            int p = 0;
            Rec hp = ts.hot[0];
            Cold cp = cold[0];   // cold part of the current node, prefetched one level ahead
            path_D = 0; my_depth = sub == 0 ? 0 : -1; pid = 0; pr = 0.0; pW = 0.0;
            int chosen = 0;
            bool widen = false, hit_terminal = false;
            while (true) {
                const int K = hp.n_child;
                if (CONT) {
                    int nn = (int)hp.node_n < P.n_sims + 1 ? (int)hp.node_n : P.n_sims + 1;
                    widen = s_pw[nn] - K > 0;   // NodeContinuous.check_pw (states.py:271-275)
                    if (widen) break;
                }
                STAMP(t_l0);
                int pick = -1;
                if (P.epsilon != 0.0) {
                    // MCTS.epsilon_greedy (mcts.py:190-195)
                    azg_u32x4 b = azg_draw(P.seed, gtree, P.search_idx, eps_draws++, AZG_STREAM_EPS);
                    if ((double)azg_u01(b.v[0]) < P.epsilon) pick = (int)(b.v[1] % (unsigned)K);
                }
                const double sq = s_sqrt[hp.node_n];
                int win_c = 0;
                if (K <= 16) {
                    // the common case: all children fit one 16-lane row
                    const bool valid = sub < K;
                    int c = 0;
                    double U = 0.0;
                    if (valid) {
                        c = CONT ? (int)ts.child[p * P.Kp + sub] : (int)hp.first + sub;
                        Rec h = ts.hot[c];
                        double ratio = sq / (double)((int)h.edge_n + 1);
                        if (CONT) {
                            U = h.Q + P.c_uct * ratio;
                        } else {
                            float pc = ts.prior[c] * P.c_uct_f;   // float32 product (NumPy >= 2 promotion)
                            U = h.Q + (double)pc * ratio;
                        }
                    }
                    if (pick >= 0) win_c = __shfl(c, pick, 16);
                    else win_c = argmax16_payload(U, valid, sub, c);
                } else {
                    double win_u = 0.0;
                    bool have = false;
                    for (int base = 0; base < K; base += 16) {   // children are scanned 16 at a time
                        const int i = base + sub;
                        const bool valid = i < K;
                        int c = 0;
                        double U = 0.0;
                        if (valid) {
                            c = CONT ? (int)ts.child[p * P.Kp + i] : (int)hp.first + i;
                            Rec h = ts.hot[c];
                            double ratio = sq / (double)((int)h.edge_n + 1);
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
                chosen = win_c;
                Rec hc = ts.hot[chosen];
                STAMP(t_l1);
#ifdef AZG_STAMPS
                st_acc[11] += t_l1 - t_l0; st_acc[12] += 1;
#endif
                if (!(hc.flags & FLAG_EXPANDED)) break;   // an edge without a child node: expand it
                path_D += 1;
                p = chosen;
                hp = hc;
                if (sub == (path_D & 15)) {   // only the slot's lane fetches the level's reward and W (used by backup_path)
                    my_depth = path_D; pid = chosen;
                    pr = cold[chosen].r; pW = edge_W[chosen];
                }
                if (hc.flags & FLAG_TERMINAL) { hit_terminal = true; break; }
                cp = cold[p];
            }
            STAMP(t_x);
            STAMP_ADD(9, t_d, t_x);    // descent until the expansion point
            if (hit_terminal) {
                leaf = p;
            } else {
                float cact = 0.0f;
                if (widen) {
                    // MCTSContinuous.add_pw_action (mcts.py:625-654)
                    const int K = hp.n_child;
                    chosen = nrec++;
                    float eps = __shfl(eps_c, chosen - kbase, 16);
                    float wmu = cp.mu, wsg = cp.sg;
                    if constexpr (GMM) {
                        float gd[15];
                        const float* g = P.gmm + (tb + p) * 3 * GMM_MAXC;
#pragma unroll
                        for (int i = 0; i < 3 * GMM_MAXC; ++i) gd[i] = g[i];
                        gmm_pick(gd, P.ncomp, P.seed, gtree, P.search_idx, (unsigned)chosen, &wmu, &wsg);
                    }
                    cact = P.bound_f * azg_tanhf(wmu + wsg * eps);
                    if (sub == 0) {
                        Rec h = make_edge<Rec>((double)cp.V, p);
                        clear_pad(h);
                        ts.hot[chosen] = h;
                        edge_W[chosen] = 0.0;
                        action[chosen] = cact;
                        ts.child[p * P.Kp + K] = (Id)chosen;
                        ts.hot[p].n_child = (decltype(hp.n_child))(K + 1);
                    }
                }
                // MCTS.expansion (mcts.py:216-238): step the env from the parent's cached state
                path_D += 1;
                double ns[S], r, sn;
                int done;
                if (CONT) {
                    if (!widen) cact = action[chosen];
                    pendulum_step(P.v1, cp.s, cp.s[2], cact, ns, &r, &done);
                    r = r / P.reward_scale;   // mcts.py:687
                } else {
                    cartpole_step(cp.s, chosen - (int)hp.first, ns, &r, &done);
                }
                float obs[4];
                env_obs<ENV>(ns, obs, &sn);
                if (sub == 0) {
                    Cold c;
#pragma unroll
                    for (int k = 0; k < 4; ++k) c.s[k] = k < S ? ns[k] : 0.0;
                    if (CONT) c.s[2] = sn;
                    c.r = r; c.V = 0.0f; c.mu = 0.0f; c.sg = 0.0f; c.pad = 0.0f;
                    cold[chosen] = c;
                    ts.hot[chosen].flags = (unsigned char)(FLAG_EXPANDED | (done ? FLAG_TERMINAL : 0));
                }
                if (sub == (path_D & 15)) { my_depth = path_D; pid = chosen; pr = r; pW = 0.0; }
                leaf = chosen;
                need_eval = !done;
                if (sub < 4) s_obsT[sub * 16 + tl] = done ? 0.0f : obs[sub];
            }
            STAMP(t_y);
            STAMP_ADD(10, t_x, t_y);   // widen + env step + node creation
        }
        __threadfence_block();
        STAMP(t_e);
        STAMP_ADD(0, t_a, t_b);   // wait at the barrier in front of the network phase
        STAMP_ADD(1, t_b, t_c);   // network phase
        STAMP_ADD(2, t_c, t_d);   // finish leaf + backup
        STAMP_ADD(3, t_d, t_e);   // select / step / expand
    }
#ifdef AZG_STAMPS
    if (lane == 0) for (int i = 0; i < 16; ++i) P.stamps[((size_t)blockIdx.x * 4 + wave) * 16 + i] = st_acc[i];
#endif
    if (live) {
        if (sub == 0) P.n_rec[tree] = nrec;
        if (TLDS) {
            // publish the LDS-resident tree in the global format
            RecL* gh = P.hot + tb;
            for (int j = sub; j < nrec; j += 16) {
                Rec h = ts.hot[j];
                RecL o;
                o.Q = h.Q; o.edge_n = h.edge_n; o.node_n = h.node_n; o.parent = (short)h.parent; o.n_child = h.n_child;
                o.first = h.first; o.flags = h.flags; o.pad = 0;
                gh[j] = o;
                if (CONT) {
                    for (int i = 0; i < (int)h.n_child; ++i) P.child[(tb + j) * P.Kp + i] = ts.child[j * P.Kp + i];
                } else {
                    P.prior[tb + j] = ts.prior[j];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ result gathering

// MCTS.return_results (mcts.py:269-307): one thread per tree, from the published (global) trees
__global__ void results_kernel(KParams P, int Kmax, int v_target, float* actions, int* counts, double* Q, double* vt, int* nch,
                               int* child_n, double* child_state, float* root_V, float* root_dist) {
    int tree = blockIdx.x * blockDim.x + threadIdx.x;
    if (tree >= P.B) return;
    size_t tb = (size_t)tree * P.R;
    const RecL* hot = P.hot + tb;
    const unsigned short* child = P.child + tb * P.Kp;
    const bool cont = P.mode == AZG_MODE_CONTINUOUS;
    const RecL root = hot[0];
    int nc = root.n_child;
    long tot = 0;
    for (int a = 0; a < nc; ++a) tot += hot[cont ? child[a] : root.first + a].edge_n;
    double qmax = 0.0, onp = 0.0;
    for (int a = 0; a < Kmax; ++a) {
        int k = a < nc ? (cont ? (int)child[a] : (int)root.first + a) : -1;
        RecL h = hot[k >= 0 ? k : 0];
        actions[(size_t)tree * Kmax + a] = k >= 0 ? (cont ? P.action[tb + k] : (float)a) : 0.0f;
        counts[(size_t)tree * Kmax + a] = k >= 0 ? h.edge_n : 0;
        Q[(size_t)tree * Kmax + a] = k >= 0 ? h.Q : 0.0;
        bool ex = k >= 0 && (h.flags & FLAG_EXPANDED);
        child_n[(size_t)tree * Kmax + a] = ex ? h.node_n : -1;
        for (int s = 0; s < P.S; ++s) child_state[((size_t)tree * Kmax + a) * P.S + s] = ex ? P.cold[tb + k].s[s] : 0.0;
        if (k >= 0) {
            if (a == 0 || h.Q > qmax) qmax = h.Q;
            if (!cont) onp += ((double)h.edge_n / (double)tot) * h.Q;
        }
    }
    if (cont) {
        // reference quirk (mcts.py:111 with Q of shape (K,1)): the K x K outer product is summed
        for (int a = 0; a < nc; ++a)
            for (int b = 0; b < nc; ++b) onp += ((double)hot[child[b]].edge_n / (double)tot) * hot[child[a]].Q;
    }
    vt[tree] = v_target == AZG_VT_ON_POLICY ? onp : qmax;
    nch[tree] = nc;
    root_V[tree] = P.cold[tb].V;
    if (cont && P.ncomp >= 2) {
        for (int part = 0; part < 3; ++part)
            for (int c = 0; c < P.ncomp; ++c)
                root_dist[(size_t)tree * 3 * P.ncomp + part * P.ncomp + c] = P.gmm[tb * 3 * GMM_MAXC + part * GMM_MAXC + c];
    } else if (cont) {
        root_dist[(size_t)tree * 2] = P.cold[tb].mu;
        root_dist[(size_t)tree * 2 + 1] = P.cold[tb].sg;
    } else {
        for (int d = 0; d < P.nd; ++d) root_dist[(size_t)tree * P.nd + d] = P.prior[tb + root.first + d];
    }
}

// One self-play step after a search, one thread per game: replay row, the agent's final action rule, the real env step,
// episode bookkeeping and the next search's root (the CPU oracle restates the same arithmetic for the parity tests).
struct SelfPlay {
    int max_len, deterministic;
    unsigned step_idx;
    int* t; int* episode; int* fcnt;
    double* ret; double* fsum;
    float* rows;          // this step's block [B][row_len]
    double* roots; int* carry;
};

__global__ void selfplay_kernel(KParams P, SelfPlay sp, int Kmax, int v_target, int env_id, int S_obs) {
    int tree = blockIdx.x * blockDim.x + threadIdx.x;
    if (tree >= P.B) return;
    const size_t tb = (size_t)tree * P.R;
    const RecL* hot = P.hot + tb;
    const unsigned short* child = P.child + tb * P.Kp;
    const bool cont = P.mode == AZG_MODE_CONTINUOUS;
    const unsigned gtree = (unsigned)(P.tree_base + tree);
    const int S = P.S, K = Kmax, RL = S_obs + 3 * Kmax + 1;
    double root[4] = {0.0, 0.0, 0.0, 0.0};
    for (int k = 0; k < S; ++k) root[k] = sp.roots[(size_t)tree * S + k];
    float* row = sp.rows + (size_t)tree * RL;
    const RecL r0 = hot[0];
    const int nc = r0.n_child;
    float obs[4];
    double sn;
    if (env_id == AZG_ENV_CARTPOLE) env_obs<AZG_ENV_CARTPOLE>(root, obs, &sn); else env_obs<AZG_ENV_PENDULUM_V1>(root, obs, &sn);
    for (int k = 0; k < S_obs; ++k) row[k] = obs[k];
    double qmax = 0.0, onp = 0.0;
    long tot = 0;
    int cmax = 0, amax = 0;
    for (int a = 0; a < nc; ++a) tot += hot[cont ? (int)child[a] : (int)r0.first + a].edge_n;
    for (int a = 0; a < K; ++a) {
        int k = a < nc ? (cont ? (int)child[a] : (int)r0.first + a) : -1;
        RecL h = hot[k >= 0 ? k : 0];
        row[S_obs + a] = k >= 0 ? (cont ? P.action[tb + k] : (float)a) : 0.0f;
        row[S_obs + K + a] = k >= 0 ? (float)h.edge_n : 0.0f;
        row[S_obs + 2 * K + a] = k >= 0 ? (float)h.Q : 0.0f;
        if (k >= 0) {
            if (a == 0 || h.Q > qmax) qmax = h.Q;
            if (!cont) onp += ((double)h.edge_n / (double)tot) * h.Q;
            if (a == 0 || h.edge_n > cmax) { cmax = h.edge_n; amax = a; }
        }
    }
    if (cont)
        for (int a = 0; a < nc; ++a)
            for (int b = 0; b < nc; ++b) onp += ((double)hot[child[b]].edge_n / (double)tot) * hot[child[a]].Q;
    row[S_obs + 3 * K] = (float)(v_target == AZG_VT_ON_POLICY ? onp : qmax);
    int pick = amax;
    if (!cont && !sp.deterministic) {
        azg_u32x4 b = azg_draw(P.seed, gtree, sp.step_idx, 0u, AZG_STREAM_ACT);
        double u = ((double)b.v[0] + 0.5) * (1.0 / 4294967296.0);
        double sum = 0.0;
        for (int a = 0; a < nc; ++a) sum = sum + (double)hot[(int)r0.first + a].edge_n / (double)cmax;
        double cum = 0.0;
        pick = nc - 1;
        for (int a = 0; a < nc; ++a) {
            cum = cum + ((double)hot[(int)r0.first + a].edge_n / (double)cmax) / sum;
            if (u < cum) { pick = a; break; }
        }
    }
    const int krec = cont ? (int)child[pick] : (int)r0.first + pick;
    double ns[4] = {0.0, 0.0, 0.0, 0.0}, r;
    int done;
    if (env_id == AZG_ENV_CARTPOLE) {
        cartpole_step(root, pick, ns, &r, &done);
    } else {
        double s1, c1;
        azg_sincos(root[0], &s1, &c1);
        pendulum_step(env_id == AZG_ENV_PENDULUM_V1, root, s1, P.action[tb + krec], ns, &r, &done);
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
        azg_reset_state(P.seed, gtree, (unsigned)ep, env_id == AZG_ENV_CARTPOLE, ns);
        sp.carry[tree] = 0;
    } else {
        RecL hk = hot[krec];
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

// ------------------------------------------------------------------------------------------------ host side

struct azg_engine {
    azg_config cfg;
    int S_env, S_obs, Kmax, Kp, R, nd, tab_n;
    int mlp_ready, HP, n_hidden, n_out, act, nreg;
    int tree_lds;            // 1: hot records live in LDS during the search
    size_t dyn_lds;          // dynamic LDS bytes per workgroup
    float ls_min, ls_max;
    hipStream_t stream;
    hipEvent_t ev0, ev1;
    KParams P;
    std::vector<void*> dev_allocs;
    std::vector<void*> weight_allocs;
    // results staging
    float* d_actions; int* d_counts; double* d_Q; double* d_vt; int* d_nch; int* d_child_n; double* d_child_state;
    float* d_rootV; float* d_rootdist;
    double* d_roots; int* d_carry;
    uint32_t search_idx;
    int sp_on, sp_max_len, sp_det, sp_cap, sp_steps, sp_row;
    uint32_t sp_step_idx;
    int* d_sp_t; int* d_sp_episode; int* d_sp_fcnt; double* d_sp_ret; double* d_sp_fsum; float* d_sp_rows;
    std::vector<void*> sp_allocs;
    int searched, results_valid;
    float last_ms;
    std::string err;
};

static std::string g_create_err;

static int fail(azg_engine* e, int code, const std::string& msg) {
    if (e) e->err = msg; else g_create_err = msg;
    return code;
}

#define HIPCHK(e, call)                                                                                  \
    do {                                                                                                 \
        hipError_t _rc = (call);                                                                         \
        if (_rc != hipSuccess) return fail(e, AZG_E_DEVICE, std::string(#call) + ": " + hipGetErrorString(_rc)); \
    } while (0)

template <typename T>
static int dalloc(azg_engine* e, T** p, size_t n, std::vector<void*>& reg) {
    void* q = nullptr;
    hipError_t rc = hipMalloc(&q, n * sizeof(T) > 0 ? n * sizeof(T) : 16);
    if (rc != hipSuccess) return fail(e, AZG_E_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(rc));
    reg.push_back(q);
    *p = (T*)q;
    return AZG_OK;
}

template <int ENV, int HP, int NREG, bool TLDS, bool GMM>
static hipError_t launch_g(azg_engine* e) {
    dim3 grid((e->cfg.n_trees + TREES_PER_WG - 1) / TREES_PER_WG), block(256);
    auto kern = search_kernel<ENV, HP, NREG, TLDS, GMM>;
    if (e->dyn_lds > 48 * 1024) {
        hipError_t rc = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->dyn_lds);
        if (rc != hipSuccess) return rc;
    }
    hipLaunchKernelGGL(kern, grid, block, e->dyn_lds, e->stream, e->P);
    return hipGetLastError();
}

// LDS plan of one launch: tables + two activation buffers (+ the 16 trees' hot records when they fit: 8-bit record ids,
// 16-bit counts, <= 16 children per node, and the 160 KB of a CU)
static void plan_lds(azg_engine* e) {
    const int ns = e->cfg.n_sims;
    const bool cont = e->cfg.mode == AZG_MODE_CONTINUOUS;
    size_t off = ((size_t)e->tab_n * 8 + (size_t)(ns + 2) * 4 + 15) / 16 * 16 + (size_t)2 * e->HP * 64;
    size_t per = (size_t)e->R * 16 + (cont ? (size_t)e->R * e->Kp : (size_t)e->R * 4);
    per = (per + 15) / 16 * 16;
    const size_t static_lds = 4096 + 1024 + 256 + 64 + 64;   // head partials, outputs, observations, head bias (+ slack)
    bool fits = e->R <= 255 && e->Kp == 16 && 4 * ns + 4 < 65536 && off + per * TREES_PER_WG + static_lds <= 160 * 1024;
    const char* force = getenv("AZG_FORCE_GLOBAL_TREE");
    if (force && force[0] == '1') fits = false;
    e->tree_lds = fits ? 1 : 0;
    e->dyn_lds = fits ? off + per * TREES_PER_WG : off;
}

template <int ENV, int HP, int NREG, bool TLDS>
static hipError_t launch_t(azg_engine* e) {
    if constexpr (ENV != AZG_ENV_CARTPOLE) {
        if (e->P.ncomp >= 2) return launch_g<ENV, HP, NREG, TLDS, true>(e);
    }
    return launch_g<ENV, HP, NREG, TLDS, false>(e);
}

template <int ENV, int HP, int NREG>
static hipError_t launch(azg_engine* e) {
    plan_lds(e);
    return e->tree_lds ? launch_t<ENV, HP, NREG, true>(e) : launch_t<ENV, HP, NREG, false>(e);
}

template <int ENV>
static hipError_t dispatch(azg_engine* e) {
    const int HP = e->HP, NR = e->nreg;
    if (HP == 64) {
        if (NR == 1) return launch<ENV, 64, 1>(e);
        if (NR == 2) return launch<ENV, 64, 2>(e);
        if (NR == 3) return launch<ENV, 64, 3>(e);
        return launch<ENV, 64, 0>(e);
    }
    if (HP == 128) {
        if (NR == 1) return launch<ENV, 128, 1>(e);
        if (NR == 2) return launch<ENV, 128, 2>(e);
        if (NR == 3) return launch<ENV, 128, 3>(e);
        return launch<ENV, 128, 0>(e);
    }
    if (HP == 256) {
        if (NR == 1) return launch<ENV, 256, 1>(e);
        return launch<ENV, 256, 0>(e);
    }
    if (HP == 512) return launch<ENV, 512, 0>(e);
    if (HP == 1024) return launch<ENV, 1024, 0>(e);
    return hipErrorInvalidValue;
}

extern "C" {

int azg_abi_version(void) { return AZG_ABI_VERSION; }

const char* azg_last_error(const azg_engine* e) { return e ? e->err.c_str() : g_create_err.c_str(); }

void azg_engine_destroy(azg_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->cfg.device_id);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (void* p : e->dev_allocs) (void)hipFree(p);
    for (void* p : e->weight_allocs) (void)hipFree(p);
    for (void* p : e->sp_allocs) (void)hipFree(p);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

int azg_engine_create(const azg_config* cfg, azg_engine** out) {
    if (!cfg || !out) return fail(nullptr, AZG_E_INVALID, "null argument");
    if (cfg->struct_size != (int32_t)sizeof(azg_config)) return fail(nullptr, AZG_E_INVALID, "azg_config size mismatch");
    if (cfg->n_trees < 1 || cfg->n_sims < 1) return fail(nullptr, AZG_E_INVALID, "n_trees and n_sims must be >= 1");
    if (cfg->env_id < 0 || cfg->env_id > 2) return fail(nullptr, AZG_E_INVALID, "unknown env_id");
    if (cfg->mode == AZG_MODE_DISCRETE && cfg->env_id != AZG_ENV_CARTPOLE)
        return fail(nullptr, AZG_E_UNSUPPORTED, "discrete mode requires a discrete-action env (CartPole)");
    if (cfg->mode == AZG_MODE_CONTINUOUS && cfg->env_id == AZG_ENV_CARTPOLE)
        return fail(nullptr, AZG_E_UNSUPPORTED, "continuous mode requires a continuous-action env (Pendulum)");
    if (cfg->mode == AZG_MODE_DISCRETE && cfg->num_actions != 2) return fail(nullptr, AZG_E_INVALID, "CartPole has num_actions == 2");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(nullptr, AZG_E_DEVICE, "no HIP device available");
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(nullptr, AZG_E_DEVICE, "device_id out of range");
    azg_engine* e = new azg_engine();
    e->cfg = *cfg;
    e->stream = nullptr; e->ev0 = e->ev1 = nullptr;
    e->mlp_ready = 0; e->searched = 0; e->results_valid = 0; e->search_idx = 0; e->last_ms = 0.0f; e->sp_on = 0;
    e->S_env = cfg->env_id == AZG_ENV_CARTPOLE ? 4 : 2;
    e->S_obs = cfg->env_id == AZG_ENV_CARTPOLE ? 4 : 3;
    const int ns = cfg->n_sims;
    std::vector<int> pw(ns + 2, 0);
    if (cfg->mode == AZG_MODE_CONTINUOUS) {
        int kmax = 1;
        for (int n = 0; n < ns + 2; ++n) {
            // NodeContinuous.check_pw (states.py:271-273): python float pow + math.ceil, evaluated on the host with libm
            double v = std::ceil(cfg->c_pw * std::pow((double)(n + 1), cfg->kappa));
            if (v > 1e6) v = 1e6;
            pw[n] = (int)v;
            if (n < ns && pw[n] > kmax) kmax = pw[n];
        }
        e->Kmax = kmax;
        e->R = ns + 2;
        e->nd = 2;
    } else {
        e->Kmax = cfg->num_actions;
        e->R = 1 + cfg->num_actions * (ns + 1);
        e->nd = cfg->num_actions;
    }
    if (e->R > 32767) { delete e; return fail(nullptr, AZG_E_UNSUPPORTED, "tree too large: records per tree must be < 32768"); }
    e->Kp = (e->Kmax + 15) / 16 * 16;
    // sqrt(n+1) table: node visit counts reach n_sims (+ the carried root count in discrete mode, <= 3 n_sims)
    e->tab_n = cfg->mode == AZG_MODE_CONTINUOUS ? ns + 2 : 4 * ns + 4;
    e->tree_lds = 0;
    e->dyn_lds = 0;
    if (hipSetDevice(cfg->device_id) != hipSuccess) { delete e; return fail(nullptr, AZG_E_DEVICE, "hipSetDevice failed"); }
#define CK(x) do { int _r = (x); if (_r != AZG_OK) { g_create_err = e->err; azg_engine_destroy(e); return _r; } } while (0)
#define HK(call) do { hipError_t _rc = (call); if (_rc != hipSuccess) { g_create_err = std::string(#call) + ": " + hipGetErrorString(_rc); azg_engine_destroy(e); return AZG_E_DEVICE; } } while (0)
    HK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    HK(hipEventCreate(&e->ev0));
    HK(hipEventCreate(&e->ev1));
    KParams& P = e->P;
    memset(&P, 0, sizeof(P));
    const size_t B = (size_t)cfg->n_trees, R = (size_t)e->R;
    RecL* hot; Cold* cold; double* edge_W; float* action; float* prior; unsigned short* child; int* n_rec;
    int* d_pw; double* d_sq;
    CK(dalloc(e, &hot, B * R, e->dev_allocs));
    CK(dalloc(e, &cold, B * R, e->dev_allocs));
    CK(dalloc(e, &edge_W, B * R, e->dev_allocs));
    CK(dalloc(e, &action, B * R, e->dev_allocs));
    CK(dalloc(e, &prior, B * R, e->dev_allocs));
    CK(dalloc(e, &child, B * R * e->Kp, e->dev_allocs));
    CK(dalloc(e, &n_rec, B, e->dev_allocs));
    CK(dalloc(e, &d_pw, (size_t)ns + 2, e->dev_allocs));
    CK(dalloc(e, &d_sq, (size_t)e->tab_n, e->dev_allocs));
    CK(dalloc(e, &e->d_roots, B * e->S_env, e->dev_allocs));
    CK(dalloc(e, &e->d_carry, B, e->dev_allocs));
    const size_t K = (size_t)e->Kmax;
    CK(dalloc(e, &e->d_actions, B * K, e->dev_allocs));
    CK(dalloc(e, &e->d_counts, B * K, e->dev_allocs));
    CK(dalloc(e, &e->d_Q, B * K, e->dev_allocs));
    CK(dalloc(e, &e->d_vt, B, e->dev_allocs));
    CK(dalloc(e, &e->d_nch, B, e->dev_allocs));
    CK(dalloc(e, &e->d_child_n, B * K, e->dev_allocs));
    CK(dalloc(e, &e->d_child_state, B * K * e->S_env, e->dev_allocs));
    CK(dalloc(e, &e->d_rootV, B, e->dev_allocs));
    CK(dalloc(e, &e->d_rootdist, B * e->nd, e->dev_allocs));
    {
        unsigned long long* st;
        CK(dalloc(e, &st, ((B + TREES_PER_WG - 1) / TREES_PER_WG) * 4 * 16, e->dev_allocs));
        e->P.stamps = st;
    }
    std::vector<double> sq(e->tab_n);
    for (int n = 0; n < e->tab_n; ++n) sq[n] = std::sqrt((double)(n + 1));
    HK(hipMemcpy(d_pw, pw.data(), sizeof(int) * (ns + 2), hipMemcpyHostToDevice));
    HK(hipMemcpy(d_sq, sq.data(), sizeof(double) * e->tab_n, hipMemcpyHostToDevice));
    HK(hipMemset(hot, 0, B * R * sizeof(RecL)));
    HK(hipMemset(e->d_carry, 0, B * sizeof(int)));
    P.B = cfg->n_trees; P.n_sims = ns; P.R = e->R; P.Kp = e->Kp; P.A = cfg->num_actions; P.nd = e->nd;
    P.v1 = cfg->env_id == AZG_ENV_PENDULUM_V1; P.tree_base = cfg->tree_id_base; P.mode = cfg->mode;
    P.c_uct = cfg->c_uct; P.gamma = cfg->gamma; P.epsilon = cfg->epsilon; P.reward_scale = cfg->reward_scale;
    P.c_uct_f = (float)cfg->c_uct; P.gamma_f = (float)cfg->gamma; P.bound_f = (float)cfg->action_bound;
    P.seed = cfg->seed; P.S = e->S_env; P.tab_n = e->tab_n;
    P.roots = e->d_roots; P.carry = e->d_carry;
    P.hot = hot; P.cold = cold; P.edge_W = edge_W; P.action = action; P.prior = prior; P.child = child;
    P.n_rec = n_rec; P.pw_need = d_pw; P.sqrt_tab = d_sq;
    *out = e;
    return AZG_OK;
#undef CK
#undef HK
}

static int pad64(int n) { return (n + 63) / 64 * 64; }
static inline int unit_of(int i) { int t = i >> 4, r = (i >> 2) & 3, g = i & 3; return 16 * t + 4 * g + r; }

int azg_set_weights(azg_engine* e, const azg_mlp_desc* d, const float* blob, size_t n_floats) {
    if (!e || !d || !blob) return AZG_E_INVALID;
    if (d->struct_size != (int32_t)sizeof(azg_mlp_desc)) return fail(e, AZG_E_INVALID, "azg_mlp_desc size mismatch");
    if (d->n_hidden < 1 || d->n_hidden > AZG_MAX_HIDDEN_LAYERS) return fail(e, AZG_E_INVALID, "n_hidden out of range");
    if (d->in_dim != e->S_obs) return fail(e, AZG_E_INVALID, "in_dim does not match the env observation");
    int ncomp = 0;
    if (e->cfg.mode == AZG_MODE_CONTINUOUS) {
        ncomp = d->num_components >= 2 ? d->num_components : 0;
        if (ncomp > 5) return fail(e, AZG_E_UNSUPPORTED, "at most 5 mixture components");
        if (d->n_dist != (ncomp ? 3 * ncomp : 2)) return fail(e, AZG_E_INVALID, "n_dist does not match num_components");
    } else if (d->n_dist != e->nd) return fail(e, AZG_E_INVALID, "n_dist does not match the engine mode");
    if (1 + d->n_dist > 16) return fail(e, AZG_E_UNSUPPORTED, "at most 15 distribution outputs");
    size_t need = 0;
    int k = d->in_dim, hmax = 0;
    for (int l = 0; l < d->n_hidden; ++l) {
        if (d->hidden[l] < 1 || d->hidden[l] > 4096) return fail(e, AZG_E_INVALID, "hidden width out of range");
        need += (size_t)d->hidden[l] * k + d->hidden[l];
        k = d->hidden[l];
        if (k > hmax) hmax = k;
    }
    need += (size_t)(1 + d->n_dist) * k + (1 + d->n_dist);
    if (need != n_floats) return fail(e, AZG_E_INVALID, "weight blob size mismatch");
    const int HP = pad64(hmax);
    if (HP != 64 && HP != 128 && HP != 256 && HP != 512 && HP != 1024)
        return fail(e, AZG_E_UNSUPPORTED, "hidden width (padded to a multiple of 64) must be one of 64,128,256,512,1024");
    HIPCHK(e, hipSetDevice(e->cfg.device_id));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    for (void* p : e->weight_allocs) (void)hipFree(p);
    e->weight_allocs.clear();
    const int NT = HP / 16, S4 = HP / 16;
    // unpack the torch-layout blob into zero-padded [HP][Kp] matrices
    std::vector<std::vector<float>> Wd(d->n_hidden), bd(d->n_hidden);
    const float* p = blob;
    int kt = d->in_dim, kp = 4;
    for (int l = 0; l < d->n_hidden; ++l) {
        int h = d->hidden[l];
        Wd[l].assign((size_t)HP * kp, 0.0f);
        bd[l].assign(HP, 0.0f);
        for (int n = 0; n < h; ++n)
            for (int kk = 0; kk < kt; ++kk) Wd[l][(size_t)n * kp + kk] = p[(size_t)n * kt + kk];
        p += (size_t)h * kt;
        for (int n = 0; n < h; ++n) bd[l][n] = p[n];
        p += h;
        kt = h; kp = HP;
    }
    const int n_out = 1 + d->n_dist;
    std::vector<float> Wh((size_t)16 * HP, 0.0f), bh(16, 0.0f);
    for (int kk = 0; kk < kt; ++kk) Wh[kk] = p[kk];
    p += kt;
    bh[0] = *p++;
    for (int o = 0; o < d->n_dist; ++o)
        for (int kk = 0; kk < kt; ++kk) Wh[(size_t)(1 + o) * HP + kk] = p[(size_t)o * kt + kk];
    p += (size_t)d->n_dist * kt;
    for (int o = 0; o < d->n_dist; ++o) bh[1 + o] = p[o];
    // MFMA operand layouts (lane l: row/col = l & 15, k-slot g = l >> 4; D register r of tile t = unit 16t + 4g + r)
    std::vector<float> W0s((size_t)NT * 64), b0s((size_t)NT * 64 * 4);
    for (int t = 0; t < NT; ++t)
        for (int l = 0; l < 64; ++l) {
            int row = 16 * t + (l & 15), g = l >> 4;
            W0s[(size_t)t * 64 + l] = Wd[0][(size_t)row * 4 + g];
            for (int r = 0; r < 4; ++r) b0s[((size_t)t * 64 + l) * 4 + r] = bd[0][16 * t + 4 * g + r];
        }
    float *dW0, *db0, *dWh, *dbh;
    if (dalloc(e, &dW0, W0s.size(), e->weight_allocs)) return AZG_E_DEVICE;
    if (dalloc(e, &db0, b0s.size(), e->weight_allocs)) return AZG_E_DEVICE;
    HIPCHK(e, hipMemcpy(dW0, W0s.data(), W0s.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(e, hipMemcpy(db0, b0s.data(), b0s.size() * 4, hipMemcpyHostToDevice));
    e->P.W0 = dW0;
    e->P.b0 = (const f32x4*)db0;
    for (int l = 1; l < d->n_hidden; ++l) {
        std::vector<float> Ws((size_t)NT * S4 * 64 * 4), bs((size_t)NT * 64 * 4);
        for (int t = 0; t < NT; ++t)
            for (int l64 = 0; l64 < 64; ++l64) {
                int row = 16 * t + (l64 & 15), g = l64 >> 4;
                for (int s4 = 0; s4 < S4; ++s4)
                    for (int j = 0; j < 4; ++j) {
                        int i = 4 * (4 * s4 + j) + g;   // canonical position consumed by k-slot g of step 4*s4+j
                        Ws[(((size_t)t * S4 + s4) * 64 + l64) * 4 + j] = Wd[l][(size_t)row * HP + unit_of(i)];
                    }
                for (int r = 0; r < 4; ++r) bs[((size_t)t * 64 + l64) * 4 + r] = bd[l][16 * t + 4 * g + r];
            }
        float *dW, *db;
        if (dalloc(e, &dW, Ws.size(), e->weight_allocs)) return AZG_E_DEVICE;
        if (dalloc(e, &db, bs.size(), e->weight_allocs)) return AZG_E_DEVICE;
        HIPCHK(e, hipMemcpy(dW, Ws.data(), Ws.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(e, hipMemcpy(db, bs.data(), bs.size() * 4, hipMemcpyHostToDevice));
        e->P.Wl[l - 1] = (const f32x4*)dW;
        e->P.bl[l - 1] = (const f32x4*)db;
    }
    std::vector<float> Whs((size_t)S4 * 64 * 4);
    for (int s4 = 0; s4 < S4; ++s4)
        for (int l64 = 0; l64 < 64; ++l64) {
            int o = l64 & 15, g = l64 >> 4;
            for (int j = 0; j < 4; ++j) {
                int i = 4 * (4 * s4 + j) + g;
                Whs[((size_t)s4 * 64 + l64) * 4 + j] = Wh[(size_t)o * HP + unit_of(i)];
            }
        }
    if (dalloc(e, &dWh, Whs.size(), e->weight_allocs)) return AZG_E_DEVICE;
    if (dalloc(e, &dbh, (size_t)16, e->weight_allocs)) return AZG_E_DEVICE;
    HIPCHK(e, hipMemcpy(dWh, Whs.data(), Whs.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(e, hipMemcpy(dbh, bh.data(), 16 * 4, hipMemcpyHostToDevice));
    e->P.Whead = (const f32x4*)dWh;
    e->P.bhead = dbh;
    if (e->cfg.mode == AZG_MODE_CONTINUOUS) {
        // per-node mixture cache and the root-distribution staging buffer are sized by n_dist (the weight set owns them)
        float* g = nullptr;
        if (ncomp) { if (dalloc(e, &g, (size_t)e->cfg.n_trees * e->R * 3 * GMM_MAXC, e->weight_allocs)) return AZG_E_DEVICE; }
        float* rd = nullptr;
        if (dalloc(e, &rd, (size_t)e->cfg.n_trees * d->n_dist, e->weight_allocs)) return AZG_E_DEVICE;
        e->P.gmm = g; e->P.ncomp = ncomp; e->d_rootdist = rd; e->nd = d->n_dist; e->P.nd = d->n_dist;
    }
    e->HP = HP; e->n_hidden = d->n_hidden; e->n_out = n_out; e->act = d->activation;
    e->P.n_hidden = d->n_hidden; e->P.n_out = n_out; e->P.act = d->activation; e->P.ls_min = d->log_std_min; e->P.ls_max = d->log_std_max;
    // hidden->hidden layers that fit the register file stay there for the whole search
    int nhh = d->n_hidden - 1;
    int regs = nhh * (HP * HP / 256);   // VGPRs per lane: each of the 4 waves holds a quarter of every layer
    e->nreg = (nhh >= 1 && nhh <= 3 && regs <= 288) ? nhh : 0;
    const char* force = getenv("AZG_FORCE_STREAM_WEIGHTS");
    if (force && force[0] == '1') e->nreg = 0;
    e->mlp_ready = 1;
    return AZG_OK;
}

int azg_set_search_index(azg_engine* e, uint32_t idx) { if (!e) return AZG_E_INVALID; e->search_idx = idx; return AZG_OK; }

int azg_upload_roots(azg_engine* e, const double* roots, const int32_t* carry) {
    if (!e || !roots) return AZG_E_INVALID;
    const int B = e->cfg.n_trees, S = e->S_env;
    if (e->cfg.env_id == AZG_ENV_CARTPOLE) {
        const double theta_thr = 12.0 * 2.0 * 3.141592653589793 / 360.0, x_thr = 2.4;
        for (int i = 0; i < B; ++i) {
            const double* s = roots + (size_t)i * S;
            if ((s[0] < -x_thr) || (s[0] > x_thr) || (s[2] < -theta_thr) || (s[2] > theta_thr))
                return fail(e, AZG_E_TERMINAL_ROOT, "Can't do tree search from a terminal node");
        }
    }
    if (carry)
        for (int i = 0; i < B; ++i)
            if (carry[i] < 0 || carry[i] > 3 * e->cfg.n_sims) return fail(e, AZG_E_INVALID, "root_n_carry out of range");
    HIPCHK(e, hipSetDevice(e->cfg.device_id));
    HIPCHK(e, hipMemcpyAsync(e->d_roots, roots, sizeof(double) * (size_t)B * S, hipMemcpyHostToDevice, e->stream));
    if (carry) HIPCHK(e, hipMemcpyAsync(e->d_carry, carry, sizeof(int) * (size_t)B, hipMemcpyHostToDevice, e->stream));
    else HIPCHK(e, hipMemsetAsync(e->d_carry, 0, sizeof(int) * (size_t)B, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return AZG_OK;
}

int azg_search_resident(azg_engine* e) {
    if (!e) return AZG_E_INVALID;
    if (!e->mlp_ready) return fail(e, AZG_E_STATE, "azg_set_weights has not been called");
    HIPCHK(e, hipSetDevice(e->cfg.device_id));
    e->P.search_idx = e->search_idx;
    HIPCHK(e, hipEventRecord(e->ev0, e->stream));
    hipError_t rc = e->cfg.env_id == AZG_ENV_CARTPOLE ? dispatch<AZG_ENV_CARTPOLE>(e) : dispatch<AZG_ENV_PENDULUM_V1>(e);
    if (rc != hipSuccess) return fail(e, AZG_E_DEVICE, std::string("search kernel launch: ") + hipGetErrorString(rc));
    HIPCHK(e, hipEventRecord(e->ev1, e->stream));
    e->search_idx += 1;
    e->searched = 1;
    e->results_valid = 0;
    return AZG_OK;
}

int azg_sync(azg_engine* e) {
    if (!e) return AZG_E_INVALID;
    HIPCHK(e, hipSetDevice(e->cfg.device_id));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return AZG_OK;
}

int azg_search(azg_engine* e, const double* roots, const int32_t* carry) {
    int rc = azg_upload_roots(e, roots, carry);
    if (rc) return rc;
    rc = azg_search_resident(e);
    if (rc) return rc;
    return azg_sync(e);
}

int azg_last_search_ms(azg_engine* e, float* ms) {
    if (!e || !ms) return AZG_E_INVALID;
    if (!e->searched) return fail(e, AZG_E_STATE, "no search has run");
    HIPCHK(e, hipSetDevice(e->cfg.device_id));
    HIPCHK(e, hipEventSynchronize(e->ev1));
    HIPCHK(e, hipEventElapsedTime(ms, e->ev0, e->ev1));
    return AZG_OK;
}

static int gather_results(azg_engine* e) {
    if (!e->searched) return fail(e, AZG_E_STATE, "no search has run");
    if (e->results_valid) return AZG_OK;
    HIPCHK(e, hipSetDevice(e->cfg.device_id));
    int B = e->cfg.n_trees;
    hipLaunchKernelGGL(results_kernel, dim3((B + 127) / 128), dim3(128), 0, e->stream, e->P, e->Kmax, e->cfg.v_target, e->d_actions,
                       e->d_counts, e->d_Q, e->d_vt, e->d_nch, e->d_child_n, e->d_child_state, e->d_rootV, e->d_rootdist);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipStreamSynchronize(e->stream));
    e->results_valid = 1;
    return AZG_OK;
}

#define D2H(dst, src, bytes) do { if (dst) HIPCHK(e, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); } while (0)

int azg_results(azg_engine* e, float* actions, int32_t* counts, double* Q, double* v_target, int32_t* n_children) {
    if (!e) return AZG_E_INVALID;
    int rc = gather_results(e);
    if (rc) return rc;
    size_t B = e->cfg.n_trees, K = e->Kmax;
    D2H(actions, e->d_actions, B * K * 4);
    D2H(counts, e->d_counts, B * K * 4);
    D2H(Q, e->d_Q, B * K * 8);
    D2H(v_target, e->d_vt, B * 8);
    D2H(n_children, e->d_nch, B * 4);
    return AZG_OK;
}

int azg_root_children(azg_engine* e, int32_t* child_n, double* child_state) {
    if (!e) return AZG_E_INVALID;
    int rc = gather_results(e);
    if (rc) return rc;
    size_t B = e->cfg.n_trees, K = e->Kmax;
    D2H(child_n, e->d_child_n, B * K * 4);
    D2H(child_state, e->d_child_state, B * K * e->S_env * 8);
    return AZG_OK;
}

int azg_root_eval(azg_engine* e, float* value, float* dist) {
    if (!e) return AZG_E_INVALID;
    int rc = gather_results(e);
    if (rc) return rc;
    size_t B = e->cfg.n_trees;
    D2H(value, e->d_rootV, B * 4);
    D2H(dist, e->d_rootdist, B * e->nd * 4);
    return AZG_OK;
}

int azg_dump_tree(azg_engine* e, int32_t* n_records, int32_t* parent, int32_t* edge_n, double* edge_W, double* edge_Q,
                  float* edge_action, int32_t* node_n, double* node_r, float* node_V, uint8_t* node_flags) {
    if (!e) return AZG_E_INVALID;
    if (!e->searched) return fail(e, AZG_E_STATE, "no search has run");
    HIPCHK(e, hipSetDevice(e->cfg.device_id));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    size_t B = e->cfg.n_trees, R = e->R;
    std::vector<RecL> hot(B * R);
    std::vector<Cold> cold(B * R);
    std::vector<int> nrec(B);
    std::vector<double> ew(B * R);
    std::vector<float> ac(B * R);
    HIPCHK(e, hipMemcpy(hot.data(), e->P.hot, B * R * sizeof(RecL), hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemcpy(cold.data(), e->P.cold, B * R * sizeof(Cold), hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemcpy(nrec.data(), e->P.n_rec, B * 4, hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemcpy(ew.data(), e->P.edge_W, B * R * 8, hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemcpy(ac.data(), e->P.action, B * R * 4, hipMemcpyDeviceToHost));
    const bool cont = e->cfg.mode == AZG_MODE_CONTINUOUS;
    const int A = e->cfg.num_actions;
    for (size_t i = 0; i < B; ++i) {
        if (n_records) n_records[i] = nrec[i];
        for (size_t j = 0; j < R; ++j) {
            size_t o = i * R + j;
            bool in = (int)j < nrec[i];
            const RecL& h = hot[o];
            bool ex = in && (h.flags & FLAG_EXPANDED);
            if (parent) parent[o] = in ? (j == 0 ? -1 : h.parent) : 0;
            if (edge_n) edge_n[o] = in ? h.edge_n : 0;
            if (edge_W) edge_W[o] = in ? ew[o] : 0.0;
            if (edge_Q) edge_Q[o] = in ? h.Q : 0.0;
            if (edge_action) edge_action[o] = in ? (cont ? ac[o] : (j == 0 ? 0.0f : (float)((j - 1) % A))) : 0.0f;
            if (node_n) node_n[o] = in ? h.node_n : 0;
            if (node_r) node_r[o] = ex ? cold[o].r : 0.0;
            if (node_V) node_V[o] = ex ? cold[o].V : 0.0f;
            if (node_flags) node_flags[o] = in ? h.flags : 0;
        }
    }
    return AZG_OK;
}

// diagnostic (-DAZG_STAMPS builds): per-wave cycle sums [n_workgroups*4][16]; returns the number of rows
int azg_debug_stamps(azg_engine* e, unsigned long long* out, size_t max_rows) {
    if (!e || !out) return AZG_E_INVALID;
    size_t rows = (size_t)((e->cfg.n_trees + TREES_PER_WG - 1) / TREES_PER_WG) * 4;
    if (rows > max_rows) rows = max_rows;
    HIPCHK(e, hipSetDevice(e->cfg.device_id));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipMemcpy(out, e->P.stamps, rows * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return (int)rows;
}

int azg_max_children(const azg_engine* e) { return e ? e->Kmax : AZG_E_INVALID; }
int azg_max_records(const azg_engine* e) { return e ? e->R : AZG_E_INVALID; }
int azg_env_state_dim(const azg_engine* e) { return e ? e->S_env : AZG_E_INVALID; }
int azg_obs_dim(const azg_engine* e) { return e ? e->S_obs : AZG_E_INVALID; }

int azg_synthetic_roots(azg_engine* e, double* roots) {
    if (!e || !roots) return AZG_E_INVALID;
    for (int i = 0; i < e->cfg.n_trees; ++i)
        azg_reset_state(e->cfg.seed, (uint32_t)(e->cfg.tree_id_base + i), 0u, e->cfg.env_id == AZG_ENV_CARTPOLE, roots + (size_t)i * e->S_env);
    return AZG_OK;
}

int azg_selfplay_row_len(const azg_engine* e) { return e ? e->S_obs + 3 * e->Kmax + 1 : AZG_E_INVALID; }

int azg_selfplay_begin(azg_engine* e, int32_t max_episode_length, int32_t deterministic, int32_t capacity_steps) {
    if (!e || max_episode_length < 1 || capacity_steps < 1) return AZG_E_INVALID;
    HIPCHK(e, hipSetDevice(e->cfg.device_id));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    for (void* p : e->sp_allocs) (void)hipFree(p);
    e->sp_allocs.clear();
    const size_t B = e->cfg.n_trees;
    e->sp_row = e->S_obs + 3 * e->Kmax + 1;
    if (dalloc(e, &e->d_sp_t, B, e->sp_allocs) || dalloc(e, &e->d_sp_episode, B, e->sp_allocs) || dalloc(e, &e->d_sp_fcnt, B, e->sp_allocs) ||
        dalloc(e, &e->d_sp_ret, B, e->sp_allocs) || dalloc(e, &e->d_sp_fsum, B, e->sp_allocs) ||
        dalloc(e, &e->d_sp_rows, (size_t)capacity_steps * B * e->sp_row, e->sp_allocs))
        return AZG_E_DEVICE;
    HIPCHK(e, hipMemset(e->d_sp_t, 0, B * 4));
    HIPCHK(e, hipMemset(e->d_sp_episode, 0, B * 4));
    HIPCHK(e, hipMemset(e->d_sp_fcnt, 0, B * 4));
    HIPCHK(e, hipMemset(e->d_sp_ret, 0, B * 8));
    HIPCHK(e, hipMemset(e->d_sp_fsum, 0, B * 8));
    std::vector<double> roots(B * e->S_env);
    azg_synthetic_roots(e, roots.data());
    HIPCHK(e, hipMemcpy(e->d_roots, roots.data(), roots.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(e, hipMemset(e->d_carry, 0, B * 4));
    e->sp_on = 1; e->sp_max_len = max_episode_length; e->sp_det = deterministic; e->sp_cap = capacity_steps; e->sp_steps = 0;
    e->sp_step_idx = 0;
    return AZG_OK;
}

int azg_selfplay_step(azg_engine* e) {
    if (!e) return AZG_E_INVALID;
    if (!e->sp_on) return fail(e, AZG_E_STATE, "azg_selfplay_begin has not been called");
    if (e->sp_steps >= e->sp_cap) return fail(e, AZG_E_STATE, "replay ring is full: download and clear the rows");
    int rc = azg_search_resident(e);
    if (rc) return rc;
    SelfPlay sp;
    sp.max_len = e->sp_max_len; sp.deterministic = e->sp_det; sp.step_idx = e->sp_step_idx;
    sp.t = e->d_sp_t; sp.episode = e->d_sp_episode; sp.fcnt = e->d_sp_fcnt; sp.ret = e->d_sp_ret; sp.fsum = e->d_sp_fsum;
    sp.rows = e->d_sp_rows + (size_t)e->sp_steps * e->cfg.n_trees * e->sp_row;
    sp.roots = e->d_roots; sp.carry = e->d_carry;
    const int B = e->cfg.n_trees;
    hipLaunchKernelGGL(selfplay_kernel, dim3((B + 127) / 128), dim3(128), 0, e->stream, e->P, sp, e->Kmax, e->cfg.v_target, e->cfg.env_id, e->S_obs);
    HIPCHK(e, hipGetLastError());
    e->sp_steps += 1;
    e->sp_step_idx += 1;
    return AZG_OK;
}

int azg_selfplay_rows(azg_engine* e, float* rows, size_t max_rows, int32_t clear) {
    if (!e) return AZG_E_INVALID;
    if (!e->sp_on) return fail(e, AZG_E_STATE, "azg_selfplay_begin has not been called");
    HIPCHK(e, hipSetDevice(e->cfg.device_id));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    size_t n = (size_t)e->sp_steps * e->cfg.n_trees;
    if (n > max_rows) n = max_rows;
    if (rows && n) HIPCHK(e, hipMemcpy(rows, e->d_sp_rows, n * e->sp_row * 4, hipMemcpyDeviceToHost));
    if (clear) e->sp_steps = 0;
    return (int)n;
}

int azg_selfplay_stats(azg_engine* e, double* fsum, int32_t* fcnt, double* env_state) {
    if (!e) return AZG_E_INVALID;
    if (!e->sp_on) return fail(e, AZG_E_STATE, "azg_selfplay_begin has not been called");
    HIPCHK(e, hipSetDevice(e->cfg.device_id));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    size_t B = e->cfg.n_trees;
    D2H(fsum, e->d_sp_fsum, B * 8);
    D2H(fcnt, e->d_sp_fcnt, B * 4);
    D2H(env_state, e->d_roots, B * e->S_env * 8);
    return AZG_OK;
}

int azg_math_selftest(int device_id, int fn_id, const double* in, double* out, size_t n) {
    if (!in || !out || n == 0) return AZG_E_INVALID;
    if (hipSetDevice(device_id) != hipSuccess) return AZG_E_DEVICE;
    double *di = nullptr, *dout = nullptr;
    if (hipMalloc((void**)&di, n * 8) != hipSuccess) return AZG_E_DEVICE;
    if (hipMalloc((void**)&dout, n * 8) != hipSuccess) { (void)hipFree(di); return AZG_E_DEVICE; }
    int rc = AZG_OK;
    if (hipMemcpy(di, in, n * 8, hipMemcpyHostToDevice) != hipSuccess) rc = AZG_E_DEVICE;
    if (rc == AZG_OK) {
        if (fn_id == 100) {
            int K = (int)((n - 1) / 2);
            hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, 0, di, dout, K);
        } else {
            hipLaunchKernelGGL(math_selftest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, fn_id, di, dout, n);
        }
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = AZG_E_DEVICE;
    }
    if (rc == AZG_OK && hipMemcpy(out, dout, n * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = AZG_E_DEVICE;
    (void)hipFree(di);
    (void)hipFree(dout);
    return rc;
}

}  // extern "C"
