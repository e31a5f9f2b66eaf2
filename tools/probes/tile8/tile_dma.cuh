// tile_dma.cuh -- PROBE CODE (tools/probes/tile8; not part of the product library): ls_tile with LDS-DMA staging.
// Measured on MI355X (tile8_probe, 4x1024 network's hidden tiles, L2-resident operands): 0.62 of the fp32 MFMA peak with one
// workgroup per CU and 0.73 with two -- register staging (ls_tile) reaches 0.66 / 0.73: a tie where it matters, so the product
// keeps the register-staged tile.
#pragma once
#include "lockstep.cuh"

// ---- The same tile with its operands brought into LDS by LDS-DMA (global_load_lds_dwordx4: 1 KB per wave-instruction, no VGPR
// round trip, no ds_write pass): the staging of ls_tile above costs the matrix pipe about a sixth of its rate even with two
// workgroups per CU (tools/probes/tile8: 0.73 of the fp32 MFMA peak; 0.81 without the global loads in the loop, 0.90 without the
// staging stores).  A DMA writes LDS when it lands, so its target stage has to be free when it is ISSUED: the weights (L2 hits,
// short latency) take two stages and are requested one chunk ahead, the activations (sc1 loads across XCDs inside the team kernel:
// long latency) three stages and two chunks ahead; both right behind the barrier that retires the stage they refill.  The chunk's
// barrier is a raw s_barrier behind a COUNTED vmcnt (the pieces of chunk c + 2 stay in flight across it; __syncthreads would drain
// them).  Same accumulation order as ls_tile: identical results.
// LDS (float4 entries): A stages [2][UT KC 64], then B stages [3][TG KC 64].
#define LS_DMA_STAGE_F4(KC) ((2 * 4 + 3 * 2) * (KC) * 64)
// One LDS-DMA piece: every lane's 16 bytes from its own global address to LDS at lds_byte_addr + 16 * lane (lds_byte_addr: wave-uniform,
// in a scalar register).  Written as inline assembly so that the instruction stays out of the compiler's s_waitcnt bookkeeping (with
// the builtin it degrades every lgkmcnt in the loop to lgkmcnt(0)); the waits are counted by hand in ls_tile_dma.  M0 (the DMA's LDS
// base) is compiler-reserved: saved and restored inside the statement.
__device__ __forceinline__ void glds16(const f32x4* src, unsigned lds_byte_addr, bool sc1) {
    unsigned keep;
    if (sc1)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(lds_byte_addr) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(lds_byte_addr) : "memory");
}

template <int HP, bool LAST, bool SC1, int KC = LS_KC>
__device__ __forceinline__ void ls_tile_dma(const KParams& P, const LockStep& L, int layer, int in_buf, int us, int g0, f32x4* s_ab, bool wt = true) {
    constexpr int TG = 2, UT = 4, WT = 2;
    constexpr int S4 = HP / 16, NU = HP / (16 * UT), NCHUNK = S4 / KC;
    constexpr int ASZ = UT * KC * 64, BSZ = TG * KC * 64;
    constexpr int NLA = ASZ / 256, NLB = BSZ / 256;       // 1 KB pieces per wave and chunk
    static_assert(S4 % KC == 0 && NCHUNK >= 3, "the DMA pipeline is written for at least three chunks");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: the DMA's LDS addresses stay in scalar registers
    const int t0 = us * UT;
    const int wg = wave % TG, wt0 = (wave / TG) * WT;
    const f32x4* W = P.Wl[layer - 1];
    const f32x4* Bsrc = L.act[in_buf];
    const TileMem<SC1> out(L.act[in_buf ^ 1], wt), parts(L.parts, wt);
    f32x4* sA = s_ab;
    f32x4* sB = s_ab + 2 * ASZ;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)s_ab;   // LDS byte address of the stages
    auto dma_a = [&](int c, int astage, int j) {
        const int e = j * 256 + tid, i = e / (KC * 64), r = e % (KC * 64);
        glds16(W + ((size_t)(t0 + i) * S4 + c * KC) * 64 + r, lds0 + 16u * (unsigned)(astage * ASZ + j * 256 + wave * 64), false);
    };
    auto dma_b = [&](int c, int bstage, int j) {
        const int e = j * 256 + tid, i = e / (KC * 64), r = e % (KC * 64);
        glds16(Bsrc + ((size_t)(g0 + i) * S4 + c * KC) * 64 + r, lds0 + 16u * (unsigned)(2 * ASZ + bstage * BSZ + j * 256 + wave * 64), SC1);
    };
    f32x4 acc[WT];
#pragma unroll
    for (int i = 0; i < WT; ++i) acc[i] = P.bl[layer - 1][(t0 + wt0 + i) * 64 + lane];
#pragma unroll
    for (int i = 0; i < WT; ++i) asm volatile("" : "+v"(acc[i]));   // the bias is here before any DMA is in flight
#pragma unroll
    for (int j = 0; j < NLA; ++j) dma_a(0, 0, j);
#pragma unroll
    for (int j = 0; j < NLB; ++j) dma_b(0, 0, j);
#pragma unroll
    for (int j = 0; j < NLB; ++j) dma_b(1, 1, j);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLB) : "memory");       // chunk 0 has landed (chunk 1's activations may still fly)
    __builtin_amdgcn_s_barrier();
    f32x4 a[WT], b, an[WT], bn;
    b = sB[wg * KC * 64 + lane];
#pragma unroll
    for (int i = 0; i < WT; ++i) a[i] = sA[((wt0 + i) * KC) * 64 + lane];
    // One chunk: as / bs = the stages chunk c reads.  Pieces are issued one per MFMA group from group 1 on: the weights of chunk
    // c + 1 first, then the activations of chunk c + 2 -- so that at the chunk's end everything but those NLB activation pieces
    // must have landed.
    auto chunk = [&](int c, int as, int bs, auto has1_t, auto has2_t) {
        constexpr bool has1 = decltype(has1_t)::value, has2 = decltype(has2_t)::value;
        const int bs2 = bs == 0 ? 2 : bs - 1;             // (c + 2) % 3
        const f32x4* cB = sB + bs * BSZ + wg * KC * 64;
        const f32x4* cA = sA + as * ASZ + wt0 * KC * 64;
#pragma unroll
        for (int s = 0; s < KC; ++s) {
            if (s + 1 < KC) {
                bn = cB[(s + 1) * 64 + lane];
#pragma unroll
                for (int i = 0; i < WT; ++i) an[i] = cA[(i * KC + s + 1) * 64 + lane];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cmp = 0; cmp < 4; ++cmp) {
#pragma unroll
                for (int i = 0; i < WT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][cmp], b[cmp], acc[i], 0, 0, 0);
                const int q = 4 * s + cmp;
                if (q >= 1 && q <= NLA + NLB) {
                    const int j = q - 1;
                    __builtin_amdgcn_sched_barrier(0);
                    if (j < NLA) { if (has1) dma_a(c + 1, as ^ 1, j); }
                    else if (has2) dma_b(c + 2, bs2, j - NLA);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (s + 1 < KC) {
                b = bn;
#pragma unroll
                for (int i = 0; i < WT; ++i) a[i] = an[i];
            }
        }
        if (has1) {
            if (has2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLB) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int bn_s = bs == 2 ? 0 : bs + 1;        // (c + 1) % 3
            b = sB[bn_s * BSZ + wg * KC * 64 + lane];
#pragma unroll
            for (int i = 0; i < WT; ++i) a[i] = sA[(as ^ 1) * ASZ + ((wt0 + i) * KC) * 64 + lane];
        }
    };
    int as = 0, bs = 0;
#pragma unroll 1
    for (int c = 0; c < NCHUNK - 2; ++c) {
        chunk(c, as, bs, std::true_type{}, std::true_type{});
        as ^= 1;
        bs = bs == 2 ? 0 : bs + 1;
    }
    chunk(NCHUNK - 2, (NCHUNK - 2) & 1, (NCHUNK - 2) % 3, std::true_type{}, std::false_type{});
    chunk(NCHUNK - 1, (NCHUNK - 1) & 1, (NCHUNK - 1) % 3, std::false_type{}, std::false_type{});
    f32x4 h[WT];
#pragma unroll
    for (int i = 0; i < WT; ++i) h[i] = act4<true>(P.act, acc[i]);
    const int tg = g0 + wg;
    if constexpr (!LAST) {
#pragma unroll
        for (int i = 0; i < WT; ++i) out.store4(((size_t)tg * S4 + t0 + wt0 + i) * 64 + lane, h[i]);
    } else {
        // the slice's 64 units are one head chunk (chunk index = us): tiles 0-1 live in waves 0..TG-1, tiles 2-3 in waves TG..3: the
        // chain's running sum crosses through LDS -- through the A stage the last chunk did NOT read (a slow wave may still read the other)
        f32x4* land = sA + (NCHUNK & 1) * ASZ;
        f32x4 hs = {0.0f, 0.0f, 0.0f, 0.0f};
        if (wt0 == 0) {
#pragma unroll
            for (int i = 0; i < WT; ++i) hs = mfma4(P.Whead[(t0 + i) * 64 + lane], h[i], hs);
            land[wg * 64 + lane] = hs;
        }
        __syncthreads();
        if (wt0 != 0) {
            hs = land[wg * 64 + lane];
#pragma unroll
            for (int i = 0; i < WT; ++i) hs = mfma4(P.Whead[(t0 + wt0 + i) * 64 + lane], h[i], hs);
            parts.store4(((size_t)tg * NU + us) * 64 + lane, hs);
        }
    }
}

