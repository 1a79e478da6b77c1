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
