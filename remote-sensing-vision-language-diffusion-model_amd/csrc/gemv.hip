// gemv.hip -- y[n] = sum_k W[n][k] x[k] (+ bias[n]) for ONE activation row: the weight-streaming products of the caption pass's
// token loop (reference: models/util.py:17-66 -> llava/model/language_model/llava_llama.py:118-137 -> Llama decode; 7 linear
// layers per decoder layer and the 128 256-row lm_head with M = 1).  HBM-bound: every weight byte is read once and used once,
// so there is no LDS tile and no MFMA -- weights go straight to registers in 16-byte pieces (cdna_hip_programming.md section 5,
// "GEMV / M <= 16 decode weights": load straight to VGPRs, deep unroll, late wait), x sits in LDS, accumulation is fp32.
// A wave owns RPW output rows at a time and keeps RPW x 2 sixteen-byte weight loads in flight per lane; the 64 partial sums of a row
// meet in a wave reduction.  Algorithmic bytes = N * K * 2 (+ K * 2 per workgroup for x, from L2).
#include "rsvld_common.h"

namespace {

constexpr int GV_WAVES = 4;    // waves per workgroup
constexpr int GV_RPW = 4;      // rows a wave accumulates at a time

template <typename T> __device__ __forceinline__ float dot8(const u32x4& w, const u32x4& x, float acc) {
    float wf[8], xf[8];
    unpack8<T>(w, wf);
    unpack8<T>(x, xf);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc = __builtin_fmaf(wf[e], xf[e], acc);
    return acc;
}

template <typename T>
__global__ __launch_bounds__(64 * GV_WAVES) void gemv_kernel(const T* __restrict__ W, const T* __restrict__ x, const T* __restrict__ bias,
                                                              T* __restrict__ y, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // x: K elements
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid * 8; i < K; i += 64 * GV_WAVES * 8) *(u32x4*)(smem + i * 2) = *(const u32x4*)(x + i);
    __syncthreads();
    const int row0 = (blockIdx.x * GV_WAVES + w) * GV_RPW;
    if (row0 >= N) return;
    const T* wr[GV_RPW];
#pragma unroll
    for (int r = 0; r < GV_RPW; ++r) wr[r] = W + (int64_t)min(row0 + r, N - 1) * K;   // rows past N re-read the last row, never stored
    float acc[GV_RPW];
#pragma unroll
    for (int r = 0; r < GV_RPW; ++r) acc[r] = 0.f;
    const int steps = K >> 9;                                   // whole 512-element steps (64 lanes x 8 elements)
    int kk = lane * 8;
    for (int s = 0; s + 1 < steps; s += 2, kk += 1024) {        // two steps per iteration: 2 x RPW loads in flight per lane
        u32x4 wv[2][GV_RPW];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < GV_RPW; ++r) wv[u][r] = __builtin_nontemporal_load((const u32x4*)(wr[r] + kk + u * 512));
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const u32x4 xv = *(const u32x4*)(smem + (kk + u * 512) * 2);
#pragma unroll
            for (int r = 0; r < GV_RPW; ++r) acc[r] = dot8<T>(wv[u][r], xv, acc[r]);
        }
    }
    for (; kk < K; kk += 512) {                                 // odd step and / or the ragged tail (K % 8 == 0)
        const u32x4 xv = *(const u32x4*)(smem + kk * 2);
#pragma unroll
        for (int r = 0; r < GV_RPW; ++r) acc[r] = dot8<T>(__builtin_nontemporal_load((const u32x4*)(wr[r] + kk)), xv, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < GV_RPW; ++r) {
        const float v = wave_sum(acc[r]);
        if (lane == 0 && row0 + r < N) y[row0 + r] = (T)(v + (bias != nullptr ? (float)bias[row0 + r] : 0.f));
    }
}

}  // namespace

extern "C" int rsvld_gemv(const void* w, const void* x, const void* bias, void* y, int N, int K, int dtype, void* stream) {
    if (!w || !x || !y || N <= 0 || K <= 0) return RSVLD_EINVAL;
    if (dtype != RSVLD_F16 && dtype != RSVLD_BF16) return RSVLD_EINVAL;
    if (K % 8 != 0 || K > 32768) return RSVLD_EUNSUPPORTED;      // 16-byte pieces; x (<= 64 KiB) in LDS
    if (((uintptr_t)w | (uintptr_t)x) & 15) return RSVLD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((N + GV_WAVES * GV_RPW - 1) / (GV_WAVES * GV_RPW)));
    const size_t smem = (size_t)K * 2;
    if (dtype == RSVLD_F16)
        hipLaunchKernelGGL(gemv_kernel<f16>, grid, dim3(64 * GV_WAVES), smem, s, (const f16*)w, (const f16*)x, (const f16*)bias, (f16*)y, N, K);
    else
        hipLaunchKernelGGL(gemv_kernel<bf16>, grid, dim3(64 * GV_WAVES), smem, s, (const bf16*)w, (const bf16*)x, (const bf16*)bias, (bf16*)y, N, K);
    return rsvld_check_launch();
}
