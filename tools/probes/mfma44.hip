// Probe of v_mfma_f32_4x4x1_16B_f32 on gfx950: operand / result lane layout, fma semantics, issue rate and dependent latency.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma44 tools/probes/mfma44.hip && tools/probes/mfma44
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const float* a, const float* b, float* d) {
    int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}

// chain of K dependent mfma on one accumulator: compare with a float fma chain
__global__ void chain_kernel(const float* a, const float* b, float c, int K, float* d) {
    int l = threadIdx.x;
    f32x4 acc = {c, c, c, c};
    for (int k = 0; k < K; ++k) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[k * 64 + l], b[k * 64 + l], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}

template <int NACC>
__global__ __launch_bounds__(512) void rate_kernel(double* out, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = 1.0f + 1e-7f * threadIdx.x, b = 1.0f - 1e-7f * threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16 / NACC; ++j)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i].x;
    if (threadIdx.x % 64 == 0) out[blockIdx.x * 8 + threadIdx.x / 64] = (double)(t1 - t0) / (iters * 16.0);
    if (s == 123.456f) out[0] = s;
}

__global__ void simd_kernel(unsigned* out) {
    unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
    if (threadIdx.x % 64 == 0) out[blockIdx.x * 16 + threadIdx.x / 64] = hw;
}

int main() {
    float *a, *b, *d;
    hipMalloc(&a, 64 * 64 * 4); hipMalloc(&b, 64 * 64 * 4); hipMalloc(&d, 256 * 4);
    std::vector<float> ha(64 * 64), hb(64 * 64), hd(256);
    // layout: a[l] = 100 + l, b[l] = 1000 + l  ->  d[l][r] = a[?] * b[?]
    for (int l = 0; l < 64; ++l) { ha[l] = 2.0f + l; hb[l] = 1000.0f + 3 * l; }
    hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
    layout_kernel<<<1, 64>>>(a, b, d);
    hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost);
    printf("layout: for lane l, register r: which (a-lane, b-lane) product\n");
    for (int l = 0; l < 64; l += 1) {
        if (l >= 8 && l < 56) continue;
        printf(" lane %2d:", l);
        for (int r = 0; r < 4; ++r) {
            int fa = -1, fb = -1;
            for (int x = 0; x < 64 && fa < 0; ++x) for (int y = 0; y < 64; ++y) if (ha[x] * hb[y] == hd[l * 4 + r]) { fa = x; fb = y; break; }
            printf("  r%d = a[%2d]*b[%2d]", r, fa, fb);
        }
        printf("\n");
    }
    // chain semantics
    const int K = 48;
    for (int i = 0; i < K * 64; ++i) { ha[i] = (float)sin(i * 0.37) * (1 + (i % 7)); hb[i] = (float)cos(i * 0.11) * powf(10.f, (i % 5) - 2); }
    hipMemcpy(a, ha.data(), K * 256, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), K * 256, hipMemcpyHostToDevice);
    chain_kernel<<<1, 64>>>(a, b, 0.3f, K, d);
    hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        // assume D[lane l][r] = sum_k a[k][4*(l/4) + r?]...: determine from the layout probe: try both conventions
        float c1 = 0.3f, c2 = 0.3f;
        for (int k = 0; k < K; ++k) {
            c1 = fmaf(ha[k * 64 + (l / 4) * 4 + r], hb[k * 64 + l], c1);       // row from register, col from lane
            c2 = fmaf(ha[k * 64 + l], hb[k * 64 + (l / 4) * 4 + r], c2);       // row from lane, col from register
        }
        if (hd[l * 4 + r] != c1 && hd[l * 4 + r] != c2) ++bad;
        if (l == 5) printf("chain lane 5 r%d: dev %.9g  conv1 %.9g conv2 %.9g\n", r, hd[l * 4 + r], c1, c2);
    }
    printf("chain mismatches vs float fma chain (either convention): %d of 256\n", bad);
    // rates
    double* out; hipMalloc(&out, 4096 * 8);
    std::vector<double> ho(4096);
    rate_kernel<4><<<256, 256>>>(out, 2000); hipDeviceSynchronize();
    rate_kernel<4><<<256, 256>>>(out, 2000); hipMemcpy(ho.data(), out, 64, hipMemcpyDeviceToHost);
    printf("4 accumulators, 1 wave/SIMD: %.2f cycles per mfma\n", ho[0]);
    rate_kernel<1><<<256, 256>>>(out, 2000); hipMemcpy(ho.data(), out, 64, hipMemcpyDeviceToHost);
    printf("1 accumulator (dependent), 1 wave/SIMD: %.2f cycles per mfma\n", ho[0]);
    rate_kernel<2><<<256, 256>>>(out, 2000); hipMemcpy(ho.data(), out, 64, hipMemcpyDeviceToHost);
    printf("2 accumulators, 1 wave/SIMD: %.2f cycles per mfma\n", ho[0]);
    rate_kernel<4><<<256, 512>>>(out, 2000); hipMemcpy(ho.data(), out, 64, hipMemcpyDeviceToHost);
    printf("4 accumulators, 2 waves/SIMD: %.2f cycles per mfma per wave (wave 0), %.2f (wave 4)\n", ho[0], ho[4]);
    rate_kernel<1><<<256, 512>>>(out, 2000); hipMemcpy(ho.data(), out, 64, hipMemcpyDeviceToHost);
    printf("1 accumulator, 2 waves/SIMD: %.2f cycles per mfma per wave\n", ho[0]);
    unsigned* uo; hipMalloc(&uo, 1024);
    simd_kernel<<<1, 512>>>(uo);
    unsigned hu[16]; hipMemcpy(hu, uo, 64, hipMemcpyDeviceToHost);
    printf("wave -> simd id of a 512-thread workgroup:");
    for (int w = 0; w < 8; ++w) printf(" w%d:%u", w, (hu[w] >> 4) & 3);
    printf("\n");
    return 0;
}
