// rsvld_common.h — shared device helpers for the gfx950 (CDNA4, wave64) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rsvld_hip.h"

typedef _Float16 f16;
typedef __bf16 bf16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define RSVLD_WAVE 64

// Per-dtype traits: 8-wide operand vector and the 32x32x16 MFMA.
// Operand maps (cdna_hip_programming.md §3): lane l holds A[row l&31][k = 8*(l>>5)+j] and
// B[k = 8*(l>>5)+j][col l&31]; D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
template <typename T> struct Mfma;
#ifndef RSVLD_MFMA16_TIMING
#define RSVLD_MFMA16_TIMING 0   // diagnostic builds only (results WRONG, timing only): the same FLOPs issued as two 16x16x32 MFMAs
#endif
template <> struct Mfma<f16> {
    typedef f16x8 v8;
    typedef f16x4 v4;
    static __device__ __forceinline__ f32x16 mma(v8 a, v8 b, f32x16 c) {
#if RSVLD_MFMA16_TIMING
        f32x4 c0 = {c[0], c[1], c[2], c[3]}, c2 = {c[8], c[9], c[10], c[11]};
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c2, 0, 0, 0);
        c[0] = c0[0]; c[1] = c0[1]; c[2] = c0[2]; c[3] = c0[3]; c[8] = c2[0]; c[9] = c2[1]; c[10] = c2[2]; c[11] = c2[3];
        return c;
#else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
    }
};
template <> struct Mfma<bf16> {
    typedef bf16x8 v8;
    typedef bf16x4 v4;
    static __device__ __forceinline__ f32x16 mma(v8 a, v8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};

template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }

// unpack / pack 8 x 16-bit <-> 8 x fp32 through a 16-byte register quad
template <typename T> __device__ __forceinline__ void unpack8(const u32x4& r, float (&f)[8]) {
    typename Mfma<T>::v8 v = __builtin_bit_cast(typename Mfma<T>::v8, r);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
}
template <typename T> __device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
    typename Mfma<T>::v8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (T)f[e];
    return __builtin_bit_cast(u32x4, v);
}

// x * sigmoid(x) with v_exp_f32 + v_rcp_f32 (1 ulp): an IEEE division here made the GroupNorm apply pass VALU-bound
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// exact-erf GELU (sgm/modules/attention.py:84-96 uses F.gelu's default).  erf by Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7, far below 16-bit output resolution): one v_rcp, one v_exp and a degree-5 Horner chain instead of
// libm erff's ~100 instructions -- the GEGLU epilogue of the K = 640 feed-forward GEMMs cost as much as their K loop.
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    float poly = 1.061405429f;
    poly = poly * t - 1.453152027f;
    poly = poly * t + 1.421413741f;
    poly = poly * t - 0.284496736f;
    poly = poly * t + 0.254829592f;
    const float e = 1.0f - poly * t * __expf(-z * z);   // erf(|x| / sqrt 2)
    return 0.5f * x * (1.0f + copysignf(e, x));
}

// The same GELU (same erf approximation, same coefficients) on a PAIR of values, written so that hipcc emits packed fp32 arithmetic
// (6 v_pk_fma_f32 + 4 v_pk_mul_f32 per pair beside the 2 v_rcp + 2 v_exp): gelu(x) = max(x, 0) - |x| * (0.5 P(t) t) exp(-x^2 / 2),
// t = 1 / (1 + 0.3275911 |x| / sqrt 2).  The GEGLU epilogue of the persistent gemm256 is pure vector work (64 GELUs per lane and
// tile, two waves per SIMD: 8-10 us beside a 20-40 us K loop); this form needs 10 packed + 6 plain + 4 transcendental instructions
// per pair where the scalar form needs 2 x (14 + 2).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu_erf2_f(f32x2_t x) {
    const f32x2_t ax = {__builtin_fabsf(x[0]), __builtin_fabsf(x[1])};
    const f32x2_t u = ax * 0.2316418882f + 1.0f;                  // 0.3275911 / sqrt 2
    const f32x2_t t = {__builtin_amdgcn_rcpf(u[0]), __builtin_amdgcn_rcpf(u[1])};
    const f32x2_t w = (x * -0.7213475204f) * x;                   // -x^2 / 2 * log2 e
    const f32x2_t e = {__builtin_amdgcn_exp2f(w[0]), __builtin_amdgcn_exp2f(w[1])};
    f32x2_t poly = t * 0.5307027145f - 0.7265760135f;             // the Abramowitz-Stegun 7.1.26 coefficients, halved
    poly = poly * t + 0.7107068705f;
    poly = poly * t - 0.142248368f;
    poly = poly * t + 0.127414796f;
    const f32x2_t q = (poly * t) * e;
    const f32x2_t r = {__builtin_fmaxf(x[0], 0.f), __builtin_fmaxf(x[1], 0.f)};
    return r - ax * q;
}

// ---- RSVLD_F16Q8 rows (round 6): C values as  [ h16: C x fp16 | C / 32 blocks of 64 B ]  = 4 C bytes.  A block holds two e4m3 parts P0, P1 of
// its 32 channels in 16-byte pieces  { P0[0:16] | P1[0:16] | P0[16:32] | P1[16:32] }: the halo kernel's lane half h reads 16-byte slots h and
// 2 + h of a 64-byte row (the fragment addresses of its fp16 k-steps), i.e. all of P_h -- its 32 bytes of v_mfma_scale_f32_32x32x64_f8f6f4.
// Activations: h16 = fp16(x), P0 = e4m3((x - h16) 2^SX_LO), P1 = e4m3(x 2^SX_HI); weights: h16 = fp16(w), P0 = e4m3(w 2^SW_HI),
// P1 = e4m3((w - h16) 2^SW_LO): lane half 0 contracts x_lo w_hi, lane half 1 x_hi w_lo -- the two cross terms of the split product.
// OCP e4m3 through v_cvt_pk_fp8_f32 (round to nearest even), saturating at +-448 by a clamp in fp32.
__device__ __forceinline__ uint32_t cvt4_e4m3(float a, float b, float c, float d) {
    int r = 0;
    r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, r, false);
    r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
    return (uint32_t)r;
}
__device__ __forceinline__ float sat_e4m3(float v) { return fminf(fmaxf(v, -448.f), 448.f); }
__device__ __forceinline__ float sat_f16_nan(float s) { return fabsf(s) > 65504.f ? copysignf(65504.f, s) : s; }   // (NaN stays NaN)
// 8 consecutive channels c0 .. c0 + 7 (c0 % 8 == 0) of one row; FIRST_LO: the lo part is P0 (activations), else P1 (weights)
template <bool FIRST_LO, int S_LO, int S_HI>
__device__ __forceinline__ void st_hq8(f16* row, int C, int c0, const float (&f)[8]) {
    f16x8 h;
    float lo[8], hs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        h[e] = (f16)sat_f16_nan(f[e]);
        lo[e] = sat_e4m3((f[e] - (float)h[e]) * (float)(1 << S_LO));
        hs[e] = sat_e4m3(f[e] * (float)(1 << S_HI));
    }
    *(u32x4*)(row + c0) = __builtin_bit_cast(u32x4, h);
    char* q = (char*)(row + C) + (c0 >> 5) * 64 + ((c0 & 16) << 1) + (c0 & 15);
    const u32x2 vlo = {cvt4_e4m3(lo[0], lo[1], lo[2], lo[3]), cvt4_e4m3(lo[4], lo[5], lo[6], lo[7])};
    const u32x2 vhi = {cvt4_e4m3(hs[0], hs[1], hs[2], hs[3]), cvt4_e4m3(hs[4], hs[5], hs[6], hs[7])};
    *(u32x2*)(q + (FIRST_LO ? 0 : 16)) = vlo;
    *(u32x2*)(q + (FIRST_LO ? 16 : 0)) = vhi;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

static inline int rsvld_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? RSVLD_OK : RSVLD_ELAUNCH;
}
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
