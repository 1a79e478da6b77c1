// gemv.hip -- y[n] = sum_k W[n][k] x[k] (+ bias[n]) for ONE activation row: the weight-streaming products of the caption pass's
// token loop (reference: models/util.py:17-66 -> llava/model/language_model/llava_llama.py:118-137 -> Llama decode; 7 linear
// layers per decoder layer and the 128 256-row lm_head with M = 1).  HBM-bound: every weight byte is read once and used once,
// so there is no LDS tile and no MFMA -- weights go straight to registers in 16-byte pieces (cdna_hip_programming.md section 5,
// "GEMV / M <= 16 decode weights": load straight to VGPRs, deep unroll, late wait), x sits in LDS, accumulation is fp32.
// A wave owns RPW output rows at a time and keeps RPW x 2 sixteen-byte weight loads in flight per lane; the 64 partial sums of a row
// meet in a wave reduction.  Algorithmic bytes = N * K * 2 (+ K * 2 per workgroup for x, from L2).
// Round 5, KS = 4: layers with few output rows (o_proj / down_proj / q|k|v of the 8 B decoder: 4 096 .. 6 144 rows = 256 .. 384 workgroups
// of 16 rows, ONE per CU with 32 KiB of weight loads in flight where ~64 KiB per CU are needed to keep HBM busy) run with the four waves
// of a workgroup SPLITTING K for the same GV_RPW rows: four times the workgroups, partial sums through LDS.
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

template <typename T, int KS>
__global__ __launch_bounds__(64 * GV_WAVES) void gemv_kernel(const T* __restrict__ W, const T* __restrict__ x, const T* __restrict__ bias,
                                                              T* __restrict__ y, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // x: K elements (+ KS = 4: 4 x GV_RPW partial sums)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid * 8; i < K; i += 64 * GV_WAVES * 8) *(u32x4*)(smem + i * 2) = *(const u32x4*)(x + i);
    __syncthreads();
    if constexpr (KS == 4) {
        // wave w owns K range [w K/4, (w + 1) K/4) (K % 2048 == 0: whole 512-element steps per wave) of the workgroup's GV_RPW rows
        const int row0 = blockIdx.x * GV_RPW;
        const T* wr[GV_RPW];
#pragma unroll
        for (int r = 0; r < GV_RPW; ++r) wr[r] = W + (int64_t)min(row0 + r, N - 1) * K;
        float acc[GV_RPW];
#pragma unroll
        for (int r = 0; r < GV_RPW; ++r) acc[r] = 0.f;
        const int kq = K >> 2, kend = (w + 1) * kq;
        int kk = w * kq + lane * 8;
        for (; kk + 512 < kend; kk += 1024) {
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
        for (; kk < kend; kk += 512) {
            const u32x4 xv = *(const u32x4*)(smem + kk * 2);
#pragma unroll
            for (int r = 0; r < GV_RPW; ++r) acc[r] = dot8<T>(__builtin_nontemporal_load((const u32x4*)(wr[r] + kk)), xv, acc[r]);
        }
        float* part = (float*)(smem + (size_t)K * 2);
#pragma unroll
        for (int r = 0; r < GV_RPW; ++r) {
            const float v = wave_sum(acc[r]);
            if (lane == 0) part[w * GV_RPW + r] = v;
        }
        __syncthreads();
        if (tid < GV_RPW && row0 + tid < N) {   // fixed order: wave 0 .. 3
            const float v = ((part[tid] + part[GV_RPW + tid]) + part[2 * GV_RPW + tid]) + part[3 * GV_RPW + tid];
            y[row0 + tid] = (T)(v + (bias != nullptr ? (float)bias[row0 + tid] : 0.f));
        }
        return;
    }
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
    // few rows (fewer than ~4 row-split workgroups per CU of a 256-CU chip): the four waves of a workgroup split K instead
    const bool ksplit = K % 2048 == 0 && (N + GV_WAVES * GV_RPW - 1) / (GV_WAVES * GV_RPW) < 1024;
    const dim3 grid(ksplit ? (unsigned)((N + GV_RPW - 1) / GV_RPW) : (unsigned)((N + GV_WAVES * GV_RPW - 1) / (GV_WAVES * GV_RPW)));
    const size_t smem = (size_t)K * 2 + (ksplit ? 4 * GV_RPW * sizeof(float) : 0);
    const f16* wh = (const f16*)w; const f16* xh = (const f16*)x; const f16* bh = (const f16*)bias;
    const bf16* wb = (const bf16*)w; const bf16* xb = (const bf16*)x; const bf16* bb = (const bf16*)bias;
    if (dtype == RSVLD_F16) {
        if (ksplit) hipLaunchKernelGGL((gemv_kernel<f16, 4>), grid, dim3(64 * GV_WAVES), smem, s, wh, xh, bh, (f16*)y, N, K);
        else hipLaunchKernelGGL((gemv_kernel<f16, 1>), grid, dim3(64 * GV_WAVES), smem, s, wh, xh, bh, (f16*)y, N, K);
    } else {
        if (ksplit) hipLaunchKernelGGL((gemv_kernel<bf16, 4>), grid, dim3(64 * GV_WAVES), smem, s, wb, xb, bb, (bf16*)y, N, K);
        else hipLaunchKernelGGL((gemv_kernel<bf16, 1>), grid, dim3(64 * GV_WAVES), smem, s, wb, xb, bb, (bf16*)y, N, K);
    }
    return rsvld_check_launch();
}
