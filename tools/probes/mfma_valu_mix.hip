// Probe: what K vector instructions cost when they sit between the v_mfma_f32_16x16x4_f32 of the SAME wave ("MFMA
// time and vector-ALU time of a SIMD add up").  One wave per SIMD, then two: cycles per MFMA for K = 0 .. 8 independent v_fma_f32 /
// v_pk_fma_f32 / v_fma_f64 placed behind every MFMA (sched_barrier-pinned, two accumulator chains so that the MFMAs themselves issue
// back to back at K = 0), and what a wave that issues ONLY those vector instructions needs for them.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_valu_mix tools/probes/mfma_valu_mix.hip && tools/probes/mfma_valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// KIND 0: v_fma_f32, 1: v_pk_fma_f32, 2: v_fma_f64.  K instructions, each on its own chain (8 chains).
template <int KIND, int K>
__device__ __forceinline__ void valu(float (&f)[8], f32x2 (&p)[8], double (&d)[8]) {
    const float c1 = 1.0000001f, c2 = 1e-9f;
    const f32x2 p1 = {1.0000001f, 1.0000001f}, p2 = {1e-9f, 1e-9f};
    const double d1 = 1.0000001, d2 = 1e-9;
#pragma unroll
    for (int i = 0; i < K; ++i) {   // (inline asm: exactly these instructions, in this order)
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(c1), "v"(c2));
        if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(p1), "v"(p2));
        if (KIND == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(d1), "v"(d2));
    }
}

template <int KIND, int K, bool WITH_MFMA>
__global__ __launch_bounds__(1024) void mix_kernel(double* out, int iters) {
    const int wave = threadIdx.x >> 6;
    float f[8]; f32x2 p[8]; double d[8];
    for (int i = 0; i < 8; ++i) { f[i] = 1.0f + 1e-3f * (threadIdx.x + i); p[i] = f32x2{f[i], f[i] + 1.0f}; d[i] = 1.0 + 1e-3 * (threadIdx.x + i); }
    f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (WITH_MFMA) acc[j & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, 1.0f, acc[j & 1], 0, 0, 0);
            valu<KIND, K>(f, p, d);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = acc[0].x + acc[1].x;
    for (int i = 0; i < 8; ++i) s += f[i] + p[i].x + p[i].y + d[i];
    if ((threadIdx.x & 63) == 0) out[wave] = (double)(t1 - t0) / (iters * 8.0);   // per MFMA slot (8 per iteration)
    if (s == 123.456) out[0] = s;
}

template <int KIND, int K>
static void one(double* dout) {
    double r[3], r2[2] = {0, 0};
    for (int c = 0; c < 3; ++c) {
        hipMemset(dout, 0, 64);
        if (c == 0) mix_kernel<KIND, K, true><<<1, 256>>>(dout, 4000);        // one wave per SIMD
        else if (c == 1) mix_kernel<KIND, K, true><<<1, 512>>>(dout, 4000);   // two waves per SIMD
        else mix_kernel<KIND, K, false><<<1, 256>>>(dout, 4000);              // the vector instructions alone
        hipDeviceSynchronize();
        std::vector<double> h(8);
        hipMemcpy(h.data(), dout, 64, hipMemcpyDeviceToHost);
        const int nw = c == 1 ? 8 : 4;
        double m = 0; for (int w = 0; w < nw; ++w) m += h[w] / nw;
        r[c] = m;
        if (c == 1) { r2[0] = (h[0] + h[1] + h[2] + h[3]) / 4; r2[1] = (h[4] + h[5] + h[6] + h[7]) / 4; }
    }
    // (two waves per SIMD: the first and the second wave's cycles per OWN MFMA slot; the second one's / 2 = per slot of the SIMD)
    printf("  K = %d: %6.1f per MFMA, one wave per SIMD | two waves per SIMD: %6.1f and %6.1f per own MFMA (%5.1f per MFMA of the SIMD) | the K instructions alone %5.1f\n",
           K, r[0], r2[0], r2[1], r2[1] / 2, r[2]);
}

template <int KIND>
static void run(const char* name, double* dout) {
    printf("%s behind every v_mfma_f32_16x16x4_f32 of the same wave [shader cycles]\n", name);
    one<KIND, 0>(dout); one<KIND, 1>(dout); one<KIND, 2>(dout); one<KIND, 3>(dout); one<KIND, 4>(dout); one<KIND, 6>(dout); one<KIND, 8>(dout);
}

// back-to-back MFMAs only, 1 .. 4 waves per SIMD (ACC accumulator chains per wave), on one CU and on every CU at once
template <int ACC>
__global__ __launch_bounds__(1024) void mfma_only(double* out, int iters) {
    const int wave = threadIdx.x >> 6;
    f32x4 acc[ACC];
    for (int i = 0; i < ACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j % ACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, 1.0f, acc[j % ACC], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < ACC; ++i) s += acc[i].x;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[wave] = (double)(t1 - t0) / (iters * 8.0);
    if (s == 123.456) out[0] = s;
}

template <int ACC>
static void mfma_rate(double* dout) {
    for (int grid : {1, 256}) {
        printf("  %d accumulator chain(s) per wave, %3d workgroup(s):", ACC, grid);
        for (int wps = 1; wps <= 4; ++wps) {
            hipMemset(dout, 0, 128);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            const int iters = 20000;
            hipEventRecord(e0);
            mfma_only<ACC><<<grid, 256 * wps>>>(dout, iters);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            std::vector<double> h(16);
            hipMemcpy(h.data(), dout, 128, hipMemcpyDeviceToHost);
            // the waves of SIMD 0 in launch order (waves 0, 4, 8, 12 of the workgroup): cycles per OWN MFMA.  The last one's figure / wps is
            // what the SIMD needs per MFMA (all waves issue the same number)
            const double tf = (double)grid * 4 * wps * iters * 8.0 * 2048.0 / (ms * 1e-3) / 1e12;
            printf("  %d wave(s)/SIMD:", wps);
            for (int k = 0; k < wps; ++k) printf(" %5.1f", h[4 * k]);
            printf(" -> %4.1f per MFMA of the SIMD", h[4 * (wps - 1)] / wps);
            if (grid > 1) printf(" (%5.1f TFLOP/s by the event clock)", tf);
            printf(";");
        }
        printf("\n");
    }
}

int main() {
    double* dout; hipMalloc(&dout, 128);
    printf("v_mfma_f32_16x16x4_f32 back to back [shader cycles by s_memtime]\n");
    mfma_rate<2>(dout); mfma_rate<4>(dout);
    run<0>("K x v_fma_f32", dout);
    run<1>("K x v_pk_fma_f32", dout);
    run<2>("K x v_fma_f64", dout);
    hipFree(dout);
    return 0;
}
