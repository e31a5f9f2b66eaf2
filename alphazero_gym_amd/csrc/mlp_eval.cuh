// mlp_eval.cuh -- batched policy/value inference for arbitrary observations (azg_mlp_eval): the network phase of the search
// kernel on its own.  One workgroup evaluates 16 observations with the weight-streaming form of mlp_forward (any width, depth,
// activation, LayerNorm, head), i.e. the same MFMA chains and chunked head sums as inside a search, so its outputs are the
// numbers a search would cache for a node with that observation.
#pragma once
#include "records.h"
#include "mlp.cuh"

template <int HP>
__global__ __launch_bounds__(256, 1) void mlp_eval_kernel(KParams P, const float* obs, int n, int S_obs, int nd, float* value, float* dist,
                                                        float* raw) {
    constexpr int NCH = head_chunks<HP>();
    __shared__ f32x4 s_parts[NCH * 64];
    __shared__ float s_obsT[128];   // [8 input rows][16 observations]
    __shared__ float s_bhead[16];
    __shared__ float s_ln[2 * 64];
    extern __shared__ f32x4 s_act[];   // two activation buffers of HP/16 tiles x 64 lanes
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid < 128) {
        const int k = tid >> 4, row = blockIdx.x * 16 + (tid & 15);
        s_obsT[tid] = (row < n && k < S_obs) ? obs[(size_t)row * S_obs + k] : 0.0f;
    }
    if (tid < 16) s_bhead[tid] = P.bhead[tid];
    WRegs<HP, 0, 4> wr;
    constexpr int NTW = HP / 64;
    if constexpr (HP <= 256) {
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            wr.w0[i] = P.W0[(wave * NTW + i) * 64 + lane];
            if (P.in8) wr.w0b[i] = P.W0b[(wave * NTW + i) * 64 + lane];
            wr.b0[i] = P.b0[(wave * NTW + i) * 64 + lane];
        }
    }
    __syncthreads();
#ifdef AZG_STAMPS
    unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    mlp_forward<HP, 0, 4, 1, 64, WRegs<HP, 0, 4>, 16, true>(P, wr, s_obsT, s_act, s_act + HP / 16 * 64, s_parts, s_ln, wave, lane, st_acc);
#else
    mlp_forward<HP, 0, 4, 1, 64, WRegs<HP, 0, 4>, 16, true>(P, wr, s_obsT, s_act, s_act + HP / 16 * 64, s_parts, s_ln, wave, lane);
#endif
    const int row = blockIdx.x * 16 + tid;
    if (tid >= 16 || row >= n) return;
    const int tl = tid;
    const float V = head_output<NCH, 64>(s_parts, s_bhead, tl, 0);
    if (value) value[row] = V;
    if (raw)
        for (int o = 0; o <= nd; ++o) raw[(size_t)row * (nd + 1) + o] = head_output<NCH, 64>(s_parts, s_bhead, tl, o);
    if (!dist) return;
    float* d = dist + (size_t)row * nd;
    if (P.mode == AZG_MODE_CONTINUOUS && P.ncomp >= 2) {
        // DiagonalGMMPolicy.forward (policies.py:544-560): mu_c, sigma_c, cumulative mixture probabilities
        float gd[15];
        gmm_params<NCH, 64>(s_parts, s_bhead, tl, P.ncomp, P.ls_min, P.ls_max, gd);
        for (int part = 0; part < 3; ++part)
            for (int c = 0; c < P.ncomp; ++c) d[part * P.ncomp + c] = gd[part * GMM_MAXC + c];
    } else if (P.mode == AZG_MODE_CONTINUOUS) {
        // DiagonalNormalPolicy.forward (policies.py:436-464)
        float ls = head_output<NCH, 64>(s_parts, s_bhead, tl, 2);
        ls = ls < P.ls_min ? P.ls_min : (ls > P.ls_max ? P.ls_max : ls);
        d[0] = head_output<NCH, 64>(s_parts, s_bhead, tl, 1);
        d[1] = azg_expf(ls);
    } else {
        // DiscretePolicy.predict_pi (policies.py:340-352): softmax of the logits
        float mx = head_output<NCH, 64>(s_parts, s_bhead, tl, 1);
        for (int a = 1; a < nd; ++a) { float v = head_output<NCH, 64>(s_parts, s_bhead, tl, 1 + a); mx = v > mx ? v : mx; }
        float sum = 0.0f;
        for (int a = 0; a < nd; ++a) sum = sum + azg_expf(head_output<NCH, 64>(s_parts, s_bhead, tl, 1 + a) - mx);
        for (int a = 0; a < nd; ++a) d[a] = azg_expf(head_output<NCH, 64>(s_parts, s_bhead, tl, 1 + a) - mx) / sum;
    }
}

template <int HP>
static hipError_t mlp_eval_launch(azg_engine* e, const float* obs, int n, float* value, float* dist, float* raw) {
    auto kern = mlp_eval_kernel<HP>;
    const size_t lds = (size_t)2 * HP * 64;
    if (lds > 48 * 1024) {
        hipError_t rc = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (rc != hipSuccess) return rc;
    }
    hipLaunchKernelGGL(kern, dim3((n + 15) / 16), dim3(256), lds, e->stream, e->P, obs, n, e->S_obs, e->nd, value, dist, raw);
    return hipGetLastError();
}
