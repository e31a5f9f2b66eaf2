// mlp.cuh -- the policy/value MLP of the 16 leaves of a workgroup on v_mfma_f32_16x16x4_f32 (+ ELU, heads, mixture head).
#pragma once
#include "records.h"

// ------------------------------------------------------------------------------------------------ MLP on MFMA

typedef int i32x4 __attribute__((ext_vector_type(4)));


// azg_expm1f on four values at once: the same operations in the same order per component (bit-identical), written
// component-parallel so that the four dependent fma chains interleave (and pack into v_pk_fma_f32)
__device__ __forceinline__ f32x4 expm1f4_nonpos(f32x4 xc) {
    // -87 <= xc <= 0 here (act4 clamps), so neither clamp of azg_expm1f can trigger
    const f32x4 magic = {12582912.0f, 12582912.0f, 12582912.0f, 12582912.0f};
    const f32x4 l2e = {1.44269504088896341f, 1.44269504088896341f, 1.44269504088896341f, 1.44269504088896341f};
    const f32x4 ln2h = {0.693145751953125f, 0.693145751953125f, 0.693145751953125f, 0.693145751953125f};
    const f32x4 ln2l = {1.42860682030941723212e-6f, 1.42860682030941723212e-6f, 1.42860682030941723212e-6f, 1.42860682030941723212e-6f};
    const f32x4 kb = __builtin_elementwise_fma(xc, l2e, magic);   // 1.5 * 2^23 + k: the integer k sits in the low mantissa bits
    const f32x4 kf = kb - magic;
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
    // 2^k = float bits (k + 127) << 23; the bits of kb are 0x4B400000 + k and 0x4B400000 << 23 == 0 (mod 2^32): one shift-add
    const i32x4 scb = (((i32x4)kb) << 23) + 0x3F800000;
    const f32x4 sc = (f32x4)scb;
    return __builtin_elementwise_fma(sc, em1, sc - one);
}

// ELU without a branch: max(x,0) + expm1(clamp(x, -87, 0)); expm1(+-0) == +0 exactly, so this equals `x > 0 ? x : expm1(x)` bit
// for bit (the sign of a zero that v_max / v_med3 may pick differently from the host's select vanishes in the sum)
template <bool RARE = true>
__device__ __forceinline__ f32x4 act4(int act, f32x4 v) {
    // max(x, 0) in one instruction per value (the max builtin first quiets a possible signalling NaN with a second v_max; so does
    // fmed3(x, 0, inf), which the compiler turns into the same pair)
    f32x4 pos;
    asm("v_max_f32 %0, 0, %1" : "=v"(pos.x) : "v"(v.x)); asm("v_max_f32 %0, 0, %1" : "=v"(pos.y) : "v"(v.y));
    asm("v_max_f32 %0, 0, %1" : "=v"(pos.z) : "v"(v.z)); asm("v_max_f32 %0, 0, %1" : "=v"(pos.w) : "v"(v.w));
    if (act == AZG_ACT_ELU) {
        f32x4 neg;
        neg.x = __builtin_amdgcn_fmed3f(v.x, -87.0f, 0.0f); neg.y = __builtin_amdgcn_fmed3f(v.y, -87.0f, 0.0f);
        neg.z = __builtin_amdgcn_fmed3f(v.z, -87.0f, 0.0f); neg.w = __builtin_amdgcn_fmed3f(v.w, -87.0f, 0.0f);
        return pos + expm1f4_nonpos(neg);
    }
    if (!RARE || act == AZG_ACT_RELU) return pos;
    // the remaining activations (leakyrelu, relu6, silu, hardswish) are compiled into the weight-streaming kernels only
    // (the host selects those kernels for them): scalar, shared with the host
    f32x4 r;
    r.x = azg_activation(act, v.x); r.y = azg_activation(act, v.y); r.z = azg_activation(act, v.z); r.w = azg_activation(act, v.w);
    return r;
}

__device__ __forceinline__ f32x4 mfma4(f32x4 a, f32x4 b, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
    return acc;
}

// Head outputs are summed in chunks of the hidden vector (chains from 0), the chunk partials then added in order:
// HP <= 256: 8 chunks of HP/8 positions; wider layers: chunks of 64 positions (4 tiles), HP/64 of them.
template <int HP>
__host__ __device__ constexpr int head_chunks() { return HP <= 256 ? 8 : HP / 64; }

// Register-resident hidden->hidden weights.  A workgroup has NW waves (4: 16 trees, 8: 32 trees = two groups of 16);
// wave w owns output tiles [w*NTW, (w+1)*NTW) of each layer, for every tree group.
// L0H: the first layer is computed by the workgroup's upper half of waves only (mlp_forward's SPLIT): they hold 2 * NTW tiles of it each.
template <int HP, int NREG, int NW = 4, bool L0H = false>
struct WRegs {
    static constexpr int NTW = HP / (16 * NW);
    static constexpr int S4 = HP / 16;
    static constexpr int NW0 = HP <= 256 ? (L0H ? 2 * NTW : NTW) : 1;   // first-layer weights are register-resident up to HP = 256
    f32x4 w[NREG > 0 ? NREG : 1][NTW][S4];
    f32x4 b[NREG > 0 ? NREG : 1][NTW];
    f32x4 wh[NTW];   // head weights of this wave's K-chunk(s)
    float w0[NW0];   // first layer, inputs 0..3 (one k-step per tile)
    float w0b[NW0];  // inputs 4..7 (only read by kernels that serve networks with more than four inputs: mlp_forward's IN8)
    f32x4 b0[NW0];
};

// nn.LayerNorm (eps 1e-5, affine) over the true units of one trunk layer for the workgroup's 16 trees.  A thread holds 16 units
// of one tree (its wave's tiles, D-register layout); it sums its own values, the 4 lane groups of the wave are added in order,
// then the 4 waves through LDS (s_ln: two 64-float buffers, one per reduction, so one barrier per reduction suffices).
__device__ __forceinline__ float ln_reduce(float s, float* buf, int wave, int lane) {
    const int tree = lane & 15;
    float v0 = __shfl(s, tree), v1 = __shfl(s, 16 + tree), v2 = __shfl(s, 32 + tree), v3 = __shfl(s, 48 + tree);
    float wsum = ((v0 + v1) + v2) + v3;
    if (lane < 16) buf[wave * 16 + tree] = wsum;
    __syncthreads();
    return ((buf[tree] + buf[16 + tree]) + buf[32 + tree]) + buf[48 + tree];
}

template <int HP>
__device__ __forceinline__ void layer_norm_wg(const KParams& P, int layer, f32x4* h, float* s_ln, int wave, int lane) {
    constexpr int NTW = HP / 64;
    const int H = P.Htrue[layer];
    const int g = lane >> 4;
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NTW; ++i) { s = s + h[i].x; s = s + h[i].y; s = s + h[i].z; s = s + h[i].w; }
    const float mean = ln_reduce(s, s_ln, wave, lane) / (float)H;
    float s2 = 0.0f;
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int u0 = 16 * (wave * NTW + i) + 4 * g;
        float d;
        d = u0 + 0 < H ? h[i].x - mean : 0.0f; s2 = s2 + d * d;
        d = u0 + 1 < H ? h[i].y - mean : 0.0f; s2 = s2 + d * d;
        d = u0 + 2 < H ? h[i].z - mean : 0.0f; s2 = s2 + d * d;
        d = u0 + 3 < H ? h[i].w - mean : 0.0f; s2 = s2 + d * d;
    }
    const float var = ln_reduce(s2, s_ln + 64, wave, lane) / (float)H;
    const float inv = 1.0f / __builtin_sqrtf(var + 1e-5f);
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int u0 = 16 * (wave * NTW + i) + 4 * g;
        const f32x4 ga = P.lng[layer][(wave * NTW + i) * 64 + lane], be = P.lnb[layer][(wave * NTW + i) * 64 + lane];
        h[i].x = u0 + 0 < H ? ((h[i].x - mean) * inv) * ga.x + be.x : 0.0f;
        h[i].y = u0 + 1 < H ? ((h[i].y - mean) * inv) * ga.y + be.y : 0.0f;
        h[i].z = u0 + 2 < H ? ((h[i].z - mean) * inv) * ga.z + be.z : 0.0f;
        h[i].w = u0 + 3 < H ? ((h[i].w - mean) * inv) * ga.w + be.w : 0.0f;
    }
}

// The MLP for the workgroup's leaves: NG groups of 16 trees, NW waves.  obsT: [4][16*NG] (input feature k, tree).  Result:
// parts[group][chunk][PSTR] = the partial head sums (head_output() combines them; PSTR = 16 keeps only output rows 0..3,
// enough for value + Normal / 2-action heads).  Activations cross waves through the act buffers ([group][HP/16 tiles][64 lanes]
// float4 = the D registers of each 16x16 output tile as they stand); a layer's output stays in registers until the next layer
// publishes it, and the last layer's output feeds the head MFMAs directly.
// (WR: WRegs<HP, NREG, NW>, or a look-alike that keeps some of the small operands elsewhere -- tools/probes/pair's server)
// NT: trees per group (16, or fewer: the remaining columns of the tile are fed zeros); obsT is then [4][NT*NG].
// IN8: the kernel also serves networks with five to eight inputs (Acrobot: six observations): obsT is [8][NT*NG] and, when P.in8 says
// so, the first layer takes a second k-step over input rows 4..7 (same accumulator: the k-ordered chain simply goes on).
// SPLIT (eight waves, one group: search_kernel.cuh): the first layer of ALL tiles is computed by waves NW/2 .. NW-1 (twice NTW tiles each)
// while waves 0 .. NW/2-1 -- the ones that walk the trees -- finish the bookkeeping of the node they have just created (tree_phase_b2);
// the barrier behind the first layer is replaced by a counter in LDS (l0_flag, monotonic: `step` = number of this network phase, from 1),
// so that the walking waves do not hold the others up: those arrive when their tiles are published and go on as soon as all NW/2 have,
// the walking waves look at the counter when they get there.
template <int HP, int NREG, int NW = 4, int NG = 1, int PSTR = 64, typename WR = WRegs<HP, NREG, NW>, int NT = 16, bool IN8 = false, bool SPLIT = false>
__device__ __forceinline__ void mlp_forward(const KParams& P, const WR& wr, const float* obsT, f32x4* actA, f32x4* actB,
                                            f32x4* parts, float* s_ln, int wave, int lane
#ifdef AZG_STAMPS
                                            , unsigned long long* st_acc
#endif
                                            , int* l0_flag = nullptr, int step = 0) {
    STAMP_M(m0, 4, -1);
    static_assert(!SPLIT || (NW == 8 && NG == 1 && NT == 16 && HP <= 256 && NREG > 0 && !IN8), "split first layer: the eight-wave / 16-tree shape");
    constexpr int NTW = HP / (16 * NW);    // output tiles per wave
    constexpr int S4 = HP / 16;            // groups of 4 MFMA k-steps over a hidden vector
    constexpr int ABUF = HP / 16 * 64;     // float4 entries of one group's activation buffer
    constexpr int NCH = head_chunks<HP>();
    static_assert(NW == 4 || NREG > 0, "the weight-streaming path is written for 4 waves");
    f32x4 h[NG][NTW];                      // this wave's tiles of the latest layer, after the activation
    if constexpr (SPLIT) {
        constexpr int NT0 = 2 * NTW;       // first-layer tiles per computing wave
        if (wave >= NW / 2) {
            const float b = obsT[(lane >> 4) * 16 + (lane & 15)];
            f32x4 a0[NT0];
#pragma unroll
            for (int i = 0; i < NT0; ++i) a0[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr.w0[i], b, wr.b0[i], 0, 0, 0);
            // (one branch around all tiles: their activation chains interleave)
            if (P.act == AZG_ACT_ELU) {
#pragma unroll
                for (int i = 0; i < NT0; ++i) a0[i] = act4<false>(AZG_ACT_ELU, a0[i]);
            } else {
#pragma unroll
                for (int i = 0; i < NT0; ++i) a0[i] = act4<false>(AZG_ACT_RELU, a0[i]);
            }
#pragma unroll
            for (int i = 0; i < NT0; ++i) actA[((wave - NW / 2) * NT0 + i) * 64 + lane] = a0[i];
            // the tiles are in LDS before the arrival is counted (LDS executes a wave's operations in order; release for the compiler)
            if (lane == 0) __hip_atomic_fetch_add(l0_flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        const int want = (NW / 2) * step;
        while (__hip_atomic_load(l0_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < want) __builtin_amdgcn_s_sleep(1);
    } else {
    // layer 0: K = in_dim <= 4 -> one k-step
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        float b;
        if constexpr (NT == 16) b = obsT[(lane >> 4) * (16 * NG) + g * 16 + (lane & 15)];
        else b = (lane & 15) < NT ? obsT[(lane >> 4) * (NT * NG) + g * NT + (lane & 15)] : 0.0f;
        float b2 = 0.0f;   // input rows 4..7
        if constexpr (IN8) {
            if constexpr (NT == 16) b2 = obsT[(4 + (lane >> 4)) * (16 * NG) + g * 16 + (lane & 15)];
            else b2 = (lane & 15) < NT ? obsT[(4 + (lane >> 4)) * (NT * NG) + g * NT + (lane & 15)] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            f32x4 a0;
            if constexpr (HP <= 256) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr.w0[i], b, wr.b0[i], 0, 0, 0);
                if constexpr (IN8) { if (P.in8) a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr.w0b[i], b2, a0, 0, 0, 0); }
            } else {   // wide layers: first-layer weights are not kept in registers
                const int nt = wave * NTW + i;
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(P.W0[nt * 64 + lane], b, P.b0[nt * 64 + lane], 0, 0, 0);
                if constexpr (IN8) { if (P.in8) a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(P.W0b[nt * 64 + lane], b2, a0, 0, 0, 0); }
            }
            if constexpr (NREG == 0) h[g][i] = act4<true>(P.act, a0);   // (weight-streaming kernels: any activation)
            else h[g][i] = a0;
        }
    }
    if constexpr (NREG > 0) {
        // one branch around all of the wave's tiles: their activation chains interleave (register-resident kernels: ELU or ReLU)
        if (P.act == AZG_ACT_ELU) {
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int i = 0; i < NTW; ++i) h[g][i] = act4<false>(AZG_ACT_ELU, h[g][i]);
        } else {
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int i = 0; i < NTW; ++i) h[g][i] = act4<false>(AZG_ACT_RELU, h[g][i]);
        }
    }
    }
    // LayerNorm is compiled into the weight-streaming kernels only (the host selects them when layernorm is on)
    if constexpr (NREG == 0) { if (P.layernorm) layer_norm_wg<HP>(P, 0, h[0], s_ln, wave, lane); }
    f32x4* buf = actA;
    f32x4* other = actB;
    // hidden->hidden layers held in registers
    if (NREG > 0) {
#pragma unroll
        for (int l = 0; l < NREG; ++l) {
            if (!(SPLIT && l == 0)) {   // (SPLIT: the first layer's output is published and waited for above)
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int i = 0; i < NTW; ++i) buf[g * ABUF + (wave * NTW + i) * 64 + lane] = h[g][i];
                __syncthreads();
            }
            STAMP_M(m1, 4, 5);
            f32x4 acc[NG][NTW];
            f32x4 bcur[NG], bnext[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
#pragma unroll
                for (int i = 0; i < NTW; ++i) acc[g][i] = wr.b[l][i];
                bcur[g] = buf[g * ABUF + lane];
            }
#pragma unroll
            for (int s4 = 0; s4 < S4; ++s4) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    bnext[g] = bcur[g];
                    if (s4 + 1 < S4) bnext[g] = buf[g * ABUF + (s4 + 1) * 64 + lane];   // prefetch the next 4 k-steps' B operand
                }
                __builtin_amdgcn_sched_barrier(0);                    // keep the ds_read above this block's MFMAs
                // k-step outer, (tile, group) inner: consecutive MFMAs are independent chains (40-cycle dependent latency)
#pragma unroll
                for (int i = 0; i < NTW; ++i)
#pragma unroll
                    for (int g = 0; g < NG; ++g) acc[g][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr.w[l][i][s4].x, bcur[g].x, acc[g][i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < NTW; ++i)
#pragma unroll
                    for (int g = 0; g < NG; ++g) acc[g][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr.w[l][i][s4].y, bcur[g].y, acc[g][i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < NTW; ++i)
#pragma unroll
                    for (int g = 0; g < NG; ++g) acc[g][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr.w[l][i][s4].z, bcur[g].z, acc[g][i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < NTW; ++i)
#pragma unroll
                    for (int g = 0; g < NG; ++g) acc[g][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr.w[l][i][s4].w, bcur[g].w, acc[g][i], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < NG; ++g) bcur[g] = bnext[g];
            }
            STAMP_M(m2, 5, 6);
            // one branch around all of the wave's tiles: their activation chains interleave (register-resident kernels: ELU or ReLU)
            if (P.act == AZG_ACT_ELU) {
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int i = 0; i < NTW; ++i) h[g][i] = act4<false>(AZG_ACT_ELU, acc[g][i]);
            } else {
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int i = 0; i < NTW; ++i) h[g][i] = act4<false>(AZG_ACT_RELU, acc[g][i]);
            }
            STAMP_M(m2b, 6, -1);
            if (l == 0) { STAMP_M_ADD(4, m0, m1); }
            STAMP_M_ADD(5, m1, m2);
            STAMP_M_ADD(6, m2, m2b);
            f32x4* t = buf; buf = other; other = t;
        }
    } else {
        // weights streamed from global memory (L2-resident), any number of layers
        for (int l = 1; l < P.n_hidden; ++l) {
#pragma unroll
            for (int i = 0; i < NTW; ++i) buf[(wave * NTW + i) * 64 + lane] = h[0][i];
            __syncthreads();
            const f32x4* W = P.Wl[l - 1];
            const f32x4* bb = P.bl[l - 1];
            // tiles in groups of at most 4 (wide layers: 16 tiles per wave would not fit the register file)
            constexpr int TG = NTW < 4 ? NTW : 4;
#pragma unroll
            for (int tg = 0; tg < NTW; tg += TG) {
                f32x4 acc[TG];
#pragma unroll
                for (int i = 0; i < TG; ++i) acc[i] = bb[(wave * NTW + tg + i) * 64 + lane];
#pragma unroll 2
                for (int s4 = 0; s4 < S4; ++s4) {
                    f32x4 b = buf[s4 * 64 + lane];
                    f32x4 a[TG];
#pragma unroll
                    for (int i = 0; i < TG; ++i) a[i] = W[((wave * NTW + tg + i) * S4 + s4) * 64 + lane];
#pragma unroll
                    for (int i = 0; i < TG; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b.x, acc[i], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < TG; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b.y, acc[i], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < TG; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b.z, acc[i], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < TG; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b.w, acc[i], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < TG; ++i) h[0][tg + i] = act4<NREG == 0>(P.act, acc[i]);
            }
            if (P.layernorm) layer_norm_wg<HP>(P, l, h[0], s_ln, wave, lane);
            f32x4* t = buf; buf = other; other = t;
        }
    }
    // heads: every chunk of the wave's hidden units is a chain from 0, straight from registers; the wave's NSUB chunks
    // (x NG groups) are independent chains, issued round-robin
    {
        constexpr int NSUB = HP <= 256 ? 8 / NW : NTW / 4;   // chunks per wave
        constexpr int KS = 4 * NTW / NSUB;                    // MFMA k-steps per chunk
        f32x4 acc[NG][NSUB];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int sc = 0; sc < NSUB; ++sc) acc[g][sc] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < KS; ++j) {
#pragma unroll
            for (int sc = 0; sc < NSUB; ++sc) {
                const int ks = sc * KS + j, ti = ks >> 2, cmp = ks & 3;
                const f32x4 a = (NREG > 0) ? wr.wh[ti] : P.Whead[(wave * NTW + ti) * 64 + lane];
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g][sc] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cmp], h[g][ti][cmp], acc[g][sc], 0, 0, 0);
            }
        }
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int sc = 0; sc < NSUB; ++sc)
                if (PSTR == 64 || lane < 16) parts[(g * NCH + wave * NSUB + sc) * PSTR + lane] = acc[g][sc];
    }
    __syncthreads();
}

// network output o of a tree (column tl of its group of 16): bias + the NCH chunk partials, added in chunk order (the
// oracle's summation order).  PSTR = 16: only outputs 0..3 were kept.
template <int NCH, int PSTR = 64>
__device__ __forceinline__ float head_output(const f32x4* parts, const float* s_bhead, int tl, int o) {
    float total = s_bhead[o];
    const int idx = (o >> 2) * 16 + tl;
#pragma unroll
    for (int w = 0; w < NCH; ++w) {
        f32x4 pv = parts[w * PSTR + idx];
        float p = (o & 3) == 0 ? pv.x : ((o & 3) == 1 ? pv.y : ((o & 3) == 2 ? pv.z : pv.w));
        total = total + p;
    }
    return total;
}

// outputs 0..3 of a tree at once (value + Normal parameters / action logits): one pass over the chunk partials, each component
// summed exactly as head_output sums it (bias first, then the chunks in order)
template <int NCH, int PSTR = 64>
__device__ __forceinline__ f32x4 head_output4(const f32x4* parts, const float* s_bhead, int tl) {
    f32x4 total = {s_bhead[0], s_bhead[1], s_bhead[2], s_bhead[3]};
#pragma unroll
    for (int w = 0; w < NCH; ++w) {
        const f32x4 pv = parts[w * PSTR + tl];
        total.x = total.x + pv.x; total.y = total.y + pv.y; total.z = total.z + pv.z; total.w = total.w + pv.w;
    }
    return total;
}

#define GMM_MAXC 5
// DiagonalGMMPolicy head (policies.py:544-560) of one node from the raw network outputs: mu_c, sigma_c = exp(clamp(log_std_c)),
// cumulative softmax(log_coeff) in component order.  d[15] = mu[5] | sigma[5] | cum[5] (fixed stride so that every index
// below is a compile-time constant and the arrays stay in registers).
template <int NCH, int PSTR = 64>
__device__ __forceinline__ void gmm_params(const f32x4* parts, const float* s_bhead, int tl, int C, float ls_min, float ls_max, float* d) {
    float mx = head_output<NCH, PSTR>(parts, s_bhead, tl, 1 + 2 * C);
#pragma unroll
    for (int c = 1; c < GMM_MAXC; ++c)
        if (c < C) { float v = head_output<NCH, PSTR>(parts, s_bhead, tl, 1 + 2 * C + c); mx = v > mx ? v : mx; }
    float ex[GMM_MAXC], sum = 0.0f, cum = 0.0f;
#pragma unroll
    for (int c = 0; c < GMM_MAXC; ++c) {
        ex[c] = 0.0f;
        if (c < C) { ex[c] = azg_expf(head_output<NCH, PSTR>(parts, s_bhead, tl, 1 + 2 * C + c) - mx); sum = sum + ex[c]; }
    }
#pragma unroll
    for (int c = 0; c < GMM_MAXC; ++c) {
        d[c] = 0.0f; d[GMM_MAXC + c] = 0.0f; d[2 * GMM_MAXC + c] = 2.0f;
        if (c < C) {
            float ls = head_output<NCH, PSTR>(parts, s_bhead, tl, 1 + C + c);
            ls = ls < ls_min ? ls_min : (ls > ls_max ? ls_max : ls);
            d[c] = head_output<NCH, PSTR>(parts, s_bhead, tl, 1 + c);
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
