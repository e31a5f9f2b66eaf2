// tile8.cuh -- PROBE CODE (tools/probes/tile8; not part of the product library): one output tile of a wide hidden->hidden layer by
// an EIGHT-wave workgroup: 64 trees (4 groups of 16) x 64 units (4 unit tiles), the shape a 64-tree team kernel with ONE workgroup
// per CU would use.  Measured (tile8_probe): 0.75 of the fp32 MFMA peak alone on a CU -- what two 4-wave ls_tile workgroups reach
// together (0.73-0.76); a team kernel built on it would trade the partner workgroup that today runs under a workgroup's hand-off
// waits and tree walks for nothing.  Not built; DESIGN.md has the account.
//
// Why another tile routine.  ls_tile (lockstep.cuh: 4 waves, 32 trees x 64 units, two LDS stages, a barrier at the end of every
// chunk) keeps the matrix pipe at about 94 % while TWO workgroups share a CU, one running under the other's barrier and LDS-refill
// bubbles, but at about 68 % when it runs alone (tools/team_profile.py) -- and with two workgroups per CU, each alternating tiles
// with hand-off waits and tree walks, "alone" is the state a CU is in 44 % of the time.  Here one workgroup has to keep the pipe
// busy by itself:
//   * two waves per SIMD (waves w and w + 4), each with two accumulator tiles: one wave's LDS waits, staging stores and barrier
//     arrivals run under the other's MFMAs;
//   * THREE LDS stages, the chunk's barrier in the MIDDLE of its MFMAs instead of at its end: the staging stores of chunk c + 1 are
//     issued in the first half of chunk c, the barrier follows them, and the operands of chunk c + 1's first k-block are read in
//     the shadow of chunk c's last MFMAs -- no drain at a chunk boundary.  (Stage (c + 1) % 3 was last read in chunk c - 2; a
//     wave that writes it in chunk c has passed the barrier inside chunk c - 1, which every wave reaches only after finishing
//     chunk c - 2.)
// Every accumulator still runs over k in ascending order; the head chunk's chain (LAST) passes from the wave that owns unit tiles
// 0-1 of a tree group to the one that owns tiles 2-3 through LDS: the arithmetic of ls_tile, bit for bit.
#pragma once
#include <type_traits>

#include "lockstep.cuh"

#define T8_KC 4                                   // k-blocks (16 k each) per staged chunk
#define T8_STAGES 3
#define T8_STAGE_F4 ((4 + 4) * T8_KC * 64)        // float4 entries of one stage: A [4 unit tiles][KC][64] + B [4 tree groups][KC][64]
#define T8_LDS_F4 (T8_STAGES * T8_STAGE_F4)       // 96 KB

// us: the 64-unit slice; g0: first of the tile's 4 tree groups; s_ab: T8_LDS_F4 float4 of LDS; 512 threads.  The caller puts a
// workgroup barrier between two calls (the next call refills stage 0 while a slow wave may still read the last chunk's stage).
// DBG (tools/probes/tile8 only; 0 in the product): 1 no global loads in the loop, 2 no staging stores / barriers, 4 no LDS operand reads
template <int HP, bool LAST, bool SC1, int DBG = 0>
__device__ __forceinline__ void ls_tile8(const KParams& P, const LockStep& L, int layer, int in_buf, int us, int g0, f32x4* s_ab, bool wt = true) {
    constexpr int KC = T8_KC, S4 = HP / 16, NCHUNK = S4 / KC, NU = HP / 64;
    constexpr int ASZ = 4 * KC * 64, STAGE = T8_STAGE_F4;
    constexpr int NL = 2 * ASZ / 512;                    // staging pieces per thread and chunk: NL / 2 of the weights, NL / 2 of the activations
    static_assert(S4 % KC == 0 && NCHUNK >= 3, "the staging pipeline is written for at least three chunks");
    static_assert(NL == 4, "512 threads, 2 x 1024 float4 per chunk");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wg = wave & 3, wt0 = (wave >> 2) * 2;      // this wave: tree group g0 + wg, unit tiles t0 + wt0, t0 + wt0 + 1
    const int t0 = us * 4;
    const f32x4* W = P.Wl[layer - 1];
    const TileMem<SC1> in(L.act[in_buf]), out(L.act[in_buf ^ 1], wt), parts(L.parts, wt);
    f32x4 rs[NL];
    auto load_one = [&](int c, int j) {
        const int jj = j & 1;
        const int e = jj * 512 + tid, i = e / (KC * 64), r = e % (KC * 64);   // unit tile / tree group, offset inside its chunk
        if (j < 2) rs[j] = W[((size_t)(t0 + i) * S4 + c * KC) * 64 + r];
        else rs[j] = in.load4(((size_t)(g0 + i) * S4 + c * KC) * 64 + r);
    };
    auto store_one = [&](int st, int j) {
        s_ab[st * STAGE + (j < 2 ? 0 : ASZ) + (j & 1) * 512 + tid] = rs[j];
    };
    f32x4 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) acc[i] = P.bl[layer - 1][(t0 + wt0 + i) * 64 + lane];
#pragma unroll
    for (int j = 0; j < NL; ++j) load_one(0, j);
#pragma unroll
    for (int j = 0; j < NL; ++j) store_one(0, j);
#pragma unroll
    for (int j = 0; j < NL; ++j) load_one(1, j);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(acc[i]));   // the bias has arrived before the loop (static waitcnt placement)
    // operands of the first k-block of chunk 0
    f32x4 a[2], b, an[2], bn;
    b = s_ab[ASZ + wg * KC * 64 + lane];
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = s_ab[((wt0 + i) * KC) * 64 + lane];
    // One chunk.  has1 / has2: chunks c + 1 / c + 2 exist.  MFMA groups q = 0 .. 4 KC - 1 (8 MFMAs per k-block, 2 per group);
    // the NL staging pieces go behind groups 1, 3, 5, 7 (stores of chunk c + 1, requests of chunk c + 2), the barrier behind group 9,
    // the first operands of chunk c + 1 are requested behind group 12 (they land under the last 3 groups' MFMAs).
    auto chunk = [&](int c, int st, auto has1_t, auto has2_t) {
        constexpr bool has1 = decltype(has1_t)::value, has2 = decltype(has2_t)::value;
        const int stn = st + 1 == T8_STAGES ? 0 : st + 1;
        const f32x4* sB = s_ab + st * STAGE + ASZ + wg * KC * 64;
        const f32x4* sA = s_ab + st * STAGE + wt0 * KC * 64;
        const f32x4* nB = s_ab + stn * STAGE + ASZ + wg * KC * 64;
        const f32x4* nA = s_ab + stn * STAGE + wt0 * KC * 64;
#pragma unroll
        for (int s = 0; s < KC; ++s) {
            if (s + 1 < KC && !(DBG & 4)) {
                bn = sB[(s + 1) * 64 + lane];
#pragma unroll
                for (int i = 0; i < 2; ++i) an[i] = sA[(i * KC + s + 1) * 64 + lane];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cmp = 0; cmp < 4; ++cmp) {
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][cmp], b[cmp], acc[i], 0, 0, 0);
                const int q = 4 * s + cmp;
                if (q < 8 && (q & 1)) {
                    const int j = q >> 1;
                    __builtin_amdgcn_sched_barrier(0);
                    if (has1 && !(DBG & 2)) store_one(stn, j);
                    if (has2 && !(DBG & 1)) load_one(c + 2, j);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (q == 9) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (has1 && !(DBG & 2)) __syncthreads();           // chunk c + 1 is staged (and every wave is past chunk c - 1)
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (q == 12 && has1 && !(DBG & 4)) {
                    __builtin_amdgcn_sched_barrier(0);
                    bn = nB[lane];
#pragma unroll
                    for (int i = 0; i < 2; ++i) an[i] = nA[(i * KC) * 64 + lane];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if ((s + 1 < KC || has1) && !(DBG & 4)) {
                b = bn;
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = an[i];
            }
        }
    };
    int st = 0;
#pragma unroll 1
    for (int c = 0; c < NCHUNK - 2; ++c) { chunk(c, st, std::true_type{}, std::true_type{}); st = st + 1 == T8_STAGES ? 0 : st + 1; }
    chunk(NCHUNK - 2, (NCHUNK - 2) % T8_STAGES, std::true_type{}, std::false_type{});
    chunk(NCHUNK - 1, (NCHUNK - 1) % T8_STAGES, std::false_type{}, std::false_type{});
    f32x4 h[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) h[i] = act4<true>(P.act, acc[i]);
    const int tg = g0 + wg;
    if constexpr (!LAST) {
#pragma unroll
        for (int i = 0; i < 2; ++i) out.store4(((size_t)tg * S4 + t0 + wt0 + i) * 64 + lane, h[i]);
    } else {
        // the slice's 64 units are one head chunk (chunk index = us): a chain from 0 over its 4 unit tiles in tile order; tiles 0-1
        // live in waves 0..3, tiles 2-3 in waves 4..7: the running sum crosses through LDS, in the stage that neither of the last two
        // chunks reads (a slow wave may still be in the tail of chunk NCHUNK - 2 or in chunk NCHUNK - 1; every wave is past the
        // barrier inside chunk NCHUNK - 2, i.e. done with chunk NCHUNK - 3, the last reader of that stage)
        constexpr int LANDING = (NCHUNK % T8_STAGES) * STAGE;
        f32x4 hs = {0.0f, 0.0f, 0.0f, 0.0f};
        if (wt0 == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) hs = mfma4(P.Whead[(t0 + i) * 64 + lane], h[i], hs);
            s_ab[LANDING + wg * 64 + lane] = hs;
        }
        __syncthreads();
        if (wt0 != 0) {
            hs = s_ab[LANDING + wg * 64 + lane];
#pragma unroll
            for (int i = 0; i < 2; ++i) hs = mfma4(P.Whead[(t0 + wt0 + i) * 64 + lane], h[i], hs);
            parts.store4(((size_t)tg * NU + us) * 64 + lane, hs);
        }
    }
}
