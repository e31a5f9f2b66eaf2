// Probe: issue rate (cycles per instruction, one wave) of the vector instructions the tree phases are made of -- fp64 / fp32 fma,
// fp64 reciprocal, 64-bit integer compare, DPP moves, i32->f64 conversion -- alone on a SIMD, with a second wave of the same
// kind on the SIMD, and next to a wave that issues v_mfma_f32_16x16x4_f32 back to back.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/valu_rate tools/probes/valu_rate.hip && tools/probes/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

// KIND: 0 fma f64, 1 fma f32, 2 rcp f64, 3 cmp i64 (v_cmp_gt_i64 + cndmask), 4 dpp mov (row_ror), 5 cvt i32->f64, 6 mul f64, 7 add f64,
//       8 ds_read_b64 (LDS round trips, independent), 9 v_pk_fma_f32
template <int KIND>
__device__ __forceinline__ void body(double (&d)[8], float (&f)[8], long long (&q)[8], int (&n)[8], const double* lds) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (KIND == 0) d[i] = __builtin_fma(d[i], 1.0000001, 1e-9);
        if (KIND == 1) f[i] = __builtin_fmaf(f[i], 1.0000001f, 1e-9f);
        if (KIND == 2) d[i] = __builtin_amdgcn_rcp(d[i]);
        if (KIND == 3) q[i] = q[i] > q[(i + 1) & 7] ? q[i] + 1 : q[(i + 1) & 7];
        if (KIND == 4) n[i] = __builtin_amdgcn_update_dpp(n[i], n[i], 0x121, 0xf, 0xf, false) + 1;   // row_ror:1
        if (KIND == 5) d[i] = (double)n[i] + d[i];
        if (KIND == 6) d[i] = d[i] * 1.0000001;
        if (KIND == 7) d[i] = d[i] + 1e-9;
        if (KIND == 8) d[i] += lds[(threadIdx.x * 8 + i * 37 + (int)d[i]) & 1023];
        if (KIND == 9) { f[i] = __builtin_fmaf(f[i], 1.0000001f, 1e-9f); }
    }
}

// role 0: the probed instruction stream; role 1: MFMA stream (waves with wave id >= n_probe)
template <int KIND>
__global__ __launch_bounds__(512) void rate_kernel(double* out, int iters, int n_probe, int prio_probe, int prio_mfma) {
    __shared__ double lds[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = 1e-12 * i;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    double d[8]; float f[8]; long long q[8]; int n[8];
    for (int i = 0; i < 8; ++i) { d[i] = 1.0 + 1e-3 * (threadIdx.x + i); f[i] = 1.0f + 1e-3f * i; q[i] = threadIdx.x * 8 + i; n[i] = threadIdx.x + i; }
    f32x4 acc[4] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    // s_setprio takes an immediate: 0 (default) or 3 here
    if (wave < n_probe) { if (prio_probe) __builtin_amdgcn_s_setprio(3); } else { if (prio_mfma) __builtin_amdgcn_s_setprio(3); }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < n_probe) {
#pragma unroll 1
        for (int it = 0; it < iters; ++it) { body<KIND>(d, f, q, n, lds); body<KIND>(d, f, q, n, lds); body<KIND>(d, f, q, n, lds); body<KIND>(d, f, q, n, lds); }
    } else {
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, 1.0f, acc[j & 3], 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += d[i] + f[i] + (double)q[i] + n[i];
    s += acc[0].x + acc[1].x + acc[2].x + acc[3].x;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = (double)(t1 - t0) / (iters * (wave < n_probe ? 32.0 : 8.0));
    if (s == 123.456) out[0] = s;
}

template <int KIND>
static void run(const char* name, double* dout) {
    // 1 wave per SIMD (4 waves), 2 probe waves per SIMD (8 waves), 1 probe + 1 MFMA wave per SIMD (8 waves, 4 probing)
    const int cfgs[5][4] = {{256, 4, 0, 0}, {512, 8, 0, 0}, {512, 4, 0, 0}, {512, 4, 1, 0}, {512, 4, 0, 1}};
    const char* what[5] = {"alone", "two such waves per SIMD", "next to an MFMA wave", "same, this wave at s_setprio 3", "same, the MFMA wave at s_setprio 3"};
    printf("%-28s", name);
    for (int c = 0; c < 5; ++c) {
        hipMemset(dout, 0, 64 * 8 * 8);
        rate_kernel<KIND><<<1, cfgs[c][0]>>>(dout, 2000, cfgs[c][1], cfgs[c][2], cfgs[c][3]);
        hipDeviceSynchronize();
        std::vector<double> h(8);
        hipMemcpy(h.data(), dout, 64, hipMemcpyDeviceToHost);
        double probe = 0; for (int w = 0; w < 4; ++w) probe += h[w] / 4;
        if (c >= 2) { double m = 0; for (int w = 4; w < 8; ++w) m += h[w] / 4; printf("  %s: %.1f (MFMA wave: %.1f per MFMA)", what[c], probe, m); }
        else printf("  %s: %.1f", what[c], probe);
        if (c >= 1) printf("\n%-28s", "");
    }
    printf("   [memtime ticks per instruction]\n");
}

int main() {
    double* dout; hipMalloc(&dout, 64 * 8 * 8);
    // scale of the clock: a wave of dependent-free MFMAs alone
    run<0>("v_fma_f64", dout);
    run<1>("v_fma_f32", dout);
    run<6>("v_mul_f64", dout);
    run<7>("v_add_f64", dout);
    run<2>("v_rcp_f64", dout);
    run<3>("cmp_i64 + select (3 instr)", dout);
    run<4>("dpp mov + add (2 instr)", dout);
    run<5>("cvt_f64_i32 + add (2 instr)", dout);
    run<8>("ds_read_b64 + addr + add", dout);
    return 0;
}
