// tile8_probe.hip -- how fast does ONE workgroup per CU run the hidden-layer tiles of a 4x1024 network, without hand-offs?
//   build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -I alphazero_gym_amd/csrc -I tools/probes/tile8 -o tools/probes/tile8/tile8_probe tools/probes/tile8/tile8_probe.hip
//   run (GPU box):  tools/probes/tile8/tile8_probe [iterations]
// Variants: ls_tile (4 waves, 32 trees x 64 units, two LDS stages) with one and with two workgroups per CU, ls_tile8 (8 waves, 64 trees
// x 64 units, three stages) with one workgroup per CU.  Every workgroup runs `iters` x 3 layer tiles back to back on L2-resident
// weights and plain activation loads; reported: TFLOP/s and the fraction of the fp32 MFMA peak (157.3 TFLOP/s).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "tile8.cuh"
#include "tile_dma.cuh"

#define CK(x) do { hipError_t rc_ = (x); if (rc_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(rc_)); exit(1); } } while (0)

template <int MINB, int DBG = 0>
__global__ __launch_bounds__(256, MINB) void probe4(KParams P, LockStep L, int iters, int TQ) {
    extern __shared__ f32x4 s_ab[];
    const int NU = 16;
    const int x = blockIdx.x % 8, j = blockIdx.x / 8, sp = NU / 8;
    const int tq = j / sp, us = x * sp + j % sp;
    for (int it = 0; it < iters; ++it)
        for (int l = 1; l <= 3; ++l) {
            ls_tile<1024, false, 2, 4, false, LS_KC, DBG>(P, L, l, (l - 1) & 1, us, 2 * (tq % TQ), s_ab);
            __syncthreads();
        }
}

// the product's tile since round 5: weights straight into the owning wave's registers, only the activations staged (ls_tile_wd)
template <int MINB, int TG, int KC>
__global__ __launch_bounds__(256, MINB) void probe4wd(KParams P, LockStep L, int iters, int TQ) {
    extern __shared__ f32x4 s_ab[];
    const int NU = 16;
    const int x = blockIdx.x % 8, j = blockIdx.x / 8, sp = NU / 8;
    const int tq = j / sp, us = x * sp + j % sp;
    for (int it = 0; it < iters; ++it)
        for (int l = 1; l <= 3; ++l) {
            ls_tile_wd<1024, false, TG, false, KC>(P, L, l, (l - 1) & 1, us, TG * (tq % TQ), s_ab);
            __syncthreads();
        }
}

template <int MINB>
__global__ __launch_bounds__(256, MINB) void probe4dma(KParams P, LockStep L, int iters, int TQ) {
    extern __shared__ f32x4 s_ab[];
    const int NU = 16;
    const int x = blockIdx.x % 8, j = blockIdx.x / 8, sp = NU / 8;
    const int tq = j / sp, us = x * sp + j % sp;
    for (int it = 0; it < iters; ++it)
        for (int l = 1; l <= 3; ++l) {
            ls_tile_dma<1024, false, false>(P, L, l, (l - 1) & 1, us, 2 * (tq % TQ), s_ab);
            __syncthreads();
        }
}

template <int DBG>
__global__ __launch_bounds__(512, 1) void probe8(KParams P, LockStep L, int iters, int TQ, unsigned long long* clk) {
    extern __shared__ f32x4 s_ab[];
    const int NU = 16;
    const int x = blockIdx.x % 8, j = blockIdx.x / 8, sp = NU / 8;
    const int tq = j / sp, us = x * sp + j % sp;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it)
        for (int l = 1; l <= 3; ++l) {
            ls_tile8<1024, false, false, DBG>(P, L, l, (l - 1) & 1, us, 4 * (tq % TQ), s_ab);
            __syncthreads();
        }
    if (threadIdx.x == 0 && blockIdx.x == 37) { clk[0] = __builtin_amdgcn_s_memtime() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    const int tqdiv = argc > 2 ? atoi(argv[2]) : 1;    // > 1: that many times fewer distinct activation blocks (more of them hit in L2)
    const int HP = 1024, S4 = HP / 16, G = 64;   // 1024 trees
    KParams P = {};
    LockStep L = {};
    P.act = AZG_ACT_ELU;
    std::vector<float> w((size_t)HP * HP), bias(HP * 4, 0.01f), act((size_t)G * HP * 16);
    for (size_t i = 0; i < w.size(); ++i) w[i] = 0.03f * ((float)((i * 2654435761u) % 2001) / 1000.0f - 1.0f);
    for (size_t i = 0; i < act.size(); ++i) act[i] = 0.5f * ((float)((i * 40503u) % 2001) / 1000.0f - 1.0f);
    for (int l = 0; l < 3; ++l) {
        float *dw, *db;
        CK(hipMalloc(&dw, w.size() * 4)); CK(hipMalloc(&db, (size_t)S4 * 64 * 16));
        CK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice));
        std::vector<float> bb((size_t)S4 * 64 * 4, 0.01f);
        CK(hipMemcpy(db, bb.data(), bb.size() * 4, hipMemcpyHostToDevice));
        P.Wl[l] = (const f32x4*)dw; P.bl[l] = (const f32x4*)db;
    }
    float *a0, *a1, *parts, *wh;
    CK(hipMalloc(&a0, act.size() * 4)); CK(hipMalloc(&a1, act.size() * 4)); CK(hipMalloc(&parts, (size_t)G * 16 * 64 * 16));
    CK(hipMalloc(&wh, (size_t)S4 * 64 * 16));
    CK(hipMemcpy(a0, act.data(), act.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(a1, act.data(), act.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(wh, 0, (size_t)S4 * 64 * 16));
    L.act[0] = (f32x4*)a0; L.act[1] = (f32x4*)a1; L.parts = (f32x4*)parts; P.Whead = (const f32x4*)wh;
    unsigned long long* dclk;
    CK(hipMalloc(&dclk, 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto report = [&](const char* name, double flop, float ms) {
        printf("%-58s %8.3f ms  %7.1f TFLOP/s  %.3f of the fp32 MFMA peak\n", name, ms, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 157.3);
    };
    const size_t lds4 = (size_t)(2 * (4 + 2) * LS_KC * 64) * 16, lds8 = (size_t)T8_LDS_F4 * 16;
    CK(hipFuncSetAttribute((const void*)probe4<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
    CK(hipFuncSetAttribute((const void*)probe4<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
    CK(hipFuncSetAttribute((const void*)probe8<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8));
    CK(hipFuncSetAttribute((const void*)probe8<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8));
    CK(hipFuncSetAttribute((const void*)probe8<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8));
    CK(hipFuncSetAttribute((const void*)probe8<7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8));
    for (int rep = 0; rep < 2; ++rep) {
        float ms;
        // 4 waves, one workgroup per CU: 256 workgroups = 16 teams of 32 trees (512 trees)
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(probe4<1>, dim3(256), dim3(256), lds4, 0, P, L, iters, 16 / tqdiv > 0 ? 16 / tqdiv : 1); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) report("ls_tile  4 waves, 32x64, 1 workgroup per CU (512 trees)", 256.0 * iters * 3 * 2.0 * 32 * 64 * 1024, ms);
        // 4 waves, two per CU: 512 workgroups (1024 trees)
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(probe4<2>, dim3(512), dim3(256), lds4, 0, P, L, iters, 32 / tqdiv > 0 ? 32 / tqdiv : 1); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) report("ls_tile  4 waves, 32x64, 2 workgroups per CU (1024 trees)", 512.0 * iters * 3 * 2.0 * 32 * 64 * 1024, ms);
        auto run4 = [&](auto kern, const char* name) {
            CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(512), dim3(256), lds4, 0, P, L, iters, 32 / tqdiv > 0 ? 32 / tqdiv : 1); CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) report(name, 512.0 * iters * 3 * 2.0 * 32 * 64 * 1024, ms);
        };
        run4(probe4<2, 8>, "   the same, loads issued but never waited for");
        run4(probe4<2, 1>, "   the same without global loads in the loop");
        run4(probe4<2, 3>, "   ... and without staging stores and barriers");
        run4(probe4<2, 7>, "   ... and without LDS operand reads (MFMAs only)");
        auto runwd = [&](auto kern, int tg, int kc, int per_cu, const char* name) {
            const size_t ldsw = (size_t)(2 * tg * kc * 64) * 16 < 16384 ? 16384 : (size_t)(2 * tg * kc * 64) * 16;
            const int teams = 256 * per_cu / 16, tqn = teams / tqdiv > 0 ? teams / tqdiv : 1;
            CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw));
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(256 * per_cu), dim3(256), ldsw, 0, P, L, iters, tqn > G / tg ? G / tg : tqn); CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) report(name, 256.0 * per_cu * iters * 3 * 2.0 * 16 * tg * 64 * 1024, ms);
        };
        runwd(probe4wd<2, 2, 2>, 2, 2, 1, "ls_tile_wd 32x64, chunks of 2, 1 workgroup per CU");
        runwd(probe4wd<2, 2, 2>, 2, 2, 2, "ls_tile_wd 32x64, chunks of 2, 2 workgroups per CU");
        runwd(probe4wd<2, 2, 4>, 2, 4, 2, "ls_tile_wd 32x64, chunks of 4, 2 workgroups per CU");
        runwd(probe4wd<3, 2, 2>, 2, 2, 3, "ls_tile_wd 32x64, chunks of 2, 3 workgroups per CU");
        runwd(probe4wd<2, 4, 2>, 4, 2, 1, "ls_tile_wd 64x64, chunks of 2, 1 workgroup per CU");
        runwd(probe4wd<2, 4, 2>, 4, 2, 2, "ls_tile_wd 64x64, chunks of 2, 2 workgroups per CU");
        runwd(probe4wd<3, 4, 2>, 4, 2, 3, "ls_tile_wd 64x64, chunks of 2, 3 workgroups per CU");
        // 8 waves, 64 x 64, one per CU (1024 trees)
        {
            const size_t ldsd = (size_t)LS_DMA_STAGE_F4(LS_KC) * 16;
            CK(hipFuncSetAttribute((const void*)probe4dma<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsd));
            CK(hipFuncSetAttribute((const void*)probe4dma<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsd));
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(probe4dma<1>, dim3(256), dim3(256), ldsd, 0, P, L, iters, 16 / tqdiv > 0 ? 16 / tqdiv : 1); CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) report("ls_tile_dma 4 waves, 32x64, LDS-DMA, 1 workgroup per CU", 256.0 * iters * 3 * 2.0 * 32 * 64 * 1024, ms);
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(probe4dma<2>, dim3(512), dim3(256), ldsd, 0, P, L, iters, 32 / tqdiv > 0 ? 32 / tqdiv : 1); CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) report("ls_tile_dma 4 waves, 32x64, LDS-DMA, 2 workgroups per CU", 512.0 * iters * 3 * 2.0 * 32 * 64 * 1024, ms);
        }
        auto run8 = [&](auto kern, const char* name) {
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds8, 0, P, L, iters, 16 / tqdiv > 0 ? 16 / tqdiv : 1, dclk); CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) {
                report(name, 256.0 * iters * 3 * 2.0 * 64 * 64 * 1024, ms);
                unsigned long long hc[2];
                CK(hipMemcpy(hc, dclk, 16, hipMemcpyDeviceToHost));
                printf("   one workgroup: %.0f shader cycles per tile (MFMA floor 32768), shader clock %.0f MHz\n", (double)hc[0] / (iters * 3.0), (double)hc[0] / ((double)hc[1] / 100.0));
            }
        };
        run8(probe8<0>, "ls_tile8 8 waves, 64x64, 3 stages, 1 workgroup per CU (1024 trees)");
        run8(probe8<1>, "   the same without global loads in the loop");
        run8(probe8<3>, "   ... and without staging stores and barriers");
        run8(probe8<7>, "   ... and without LDS operand reads (MFMAs only)");
    }
    CK(hipDeviceSynchronize());
    return 0;
}
