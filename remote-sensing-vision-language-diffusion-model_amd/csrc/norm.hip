// norm.hip — GroupNorm (+SiLU, +ZeroSFT modulation) and LayerNorm over NHWC / token-major
// 16-bit tensors.  HBM-bound: every pass moves 16 B per lane, statistics in fp32 with a
// deterministic two-level reduction (per-block partials -> fp64 finalize), no atomics.
#include "rsvld_common.h"

namespace {

// ---------------------------------------------------------------------------------------
// pass 1: per (image, row-chunk) partial sums  part[b][chunk][g] = (sum, sumsq)
// thread -> fixed 8-channel chunk, strided over rows; per-channel sums go through LDS so
// that any group size (2 .. C/groups, not necessarily a multiple of 8) is handled.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* __restrict__ x1, const T* __restrict__ x2,
                                                         float* __restrict__ part, int HW, int C1, int C2,
                                                         int groups, int rows_per_chunk, int nchunks) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sm = (float*)smem_raw;  // [rif][C][2]
    const int C = C1 + C2, C8 = C >> 3, C1_8 = C1 >> 3;
    const int TPR = C8 < 256 ? C8 : 256;  // threads per row
    const int rif = 256 / TPR;            // rows in flight
    const int tid = threadIdx.x;
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int row_lo = chunk * rows_per_chunk;
    const int row_hi = min(HW, row_lo + rows_per_chunk);
    const int tc = tid % TPR, rsub = tid / TPR;
    if (rsub < rif) {
        for (int cc = tc; cc < C8; cc += TPR) {
            float s[8], ss[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { s[e] = 0.f; ss[e] = 0.f; }
            const T* src;
            int64_t cstride;
            int coff;
            if (cc < C1_8) { src = x1 + (int64_t)b * HW * C1; cstride = C1; coff = cc * 8; }
            else { src = x2 + (int64_t)b * HW * C2; cstride = C2; coff = (cc - C1_8) * 8; }
            int r = row_lo + rsub;
            for (; r + 3 * rif < row_hi; r += 4 * rif) {  // 4 independent 16-B loads in flight
                u32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *(const u32x4*)(src + (int64_t)(r + u * rif) * cstride + coff);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float f[8];
                    unpack8<T>(v[u], f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { s[e] += f[e]; ss[e] += f[e] * f[e]; }
                }
            }
            for (; r < row_hi; r += rif) {
                const u32x4 v = *(const u32x4*)(src + (int64_t)r * cstride + coff);
                float f[8];
                unpack8<T>(v, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) { s[e] += f[e]; ss[e] += f[e] * f[e]; }
            }
            float* dst = sm + ((int64_t)rsub * C + cc * 8) * 2;
#pragma unroll
            for (int e = 0; e < 8; ++e) { dst[2 * e] = s[e]; dst[2 * e + 1] = ss[e]; }
        }
    }
    __syncthreads();
    const int gs = C / groups;
    for (int g = tid; g < groups; g += 256) {
        float s = 0.f, ss = 0.f;
        for (int r = 0; r < rif; ++r) {
            const float* src = sm + ((int64_t)r * C + g * gs) * 2;
            for (int e = 0; e < gs; ++e) { s += src[2 * e]; ss += src[2 * e + 1]; }
        }
        float* o = part + (((int64_t)b * nchunks + chunk) * groups + g) * 2;
        o[0] = s;
        o[1] = ss;
    }
}

// pass 2: stats[b][g] = (mean, biased var); one wave per (image, group), lanes stride over the
// chunk partials, fp64 merge in a fixed order (deterministic)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ part, float* __restrict__ stats,
                                                          int groups, int nchunks, double inv_count, int total) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= total) return;
    const int b = i / groups, g = i - b * groups;
    double s = 0.0, ss = 0.0;
    for (int c = lane; c < nchunks; c += 64) {
        const float* p = part + (((int64_t)b * nchunks + c) * groups + g) * 2;
        s += (double)p[0];
        ss += (double)p[1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        ss += __shfl_xor(ss, o);
    }
    if (lane == 0) {
        const double mean = s * inv_count;
        double var = ss * inv_count - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[2 * i] = (float)mean;
        stats[2 * i + 1] = (float)var;
    }
}

// One block per (image, group): merge partial sums in fp64 (fixed order -> deterministic) and emit the
// per-channel affine ab[b][c] = (gamma*rstd, beta - mean*gamma*rstd) of that group's channels.
//   PER_CHANNEL = false: partials [b][chunk][group][2] from gn_partial_kernel (a statistics pass over x)
//   PER_CHANNEL = true : partials [b][tile][Cset][2] written by a conv epilogue (conv_halo.hip), one or two
//                        producers (the skip concat [x | x2] is normalised jointly)
template <bool PER_CHANNEL>
__global__ __launch_bounds__(256) void gn_ab_kernel(const float* __restrict__ part1, int n1, int C1,
                                                    const float* __restrict__ part2, int n2, int C2,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    float* __restrict__ ab, float* __restrict__ stats_out, int groups,
                                                    float eps, double inv_count) {
    __shared__ double red[2][4];
    __shared__ float mr[2];
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int C = C1 + C2, gs = C / groups;
    double s = 0.0, ss = 0.0;
    if (PER_CHANNEL) {
        for (int c = g * gs; c < (g + 1) * gs; ++c) {
            const float* base;
            int n, Cs, cc;
            if (c < C1) { base = part1 + (int64_t)b * n1 * C1 * 2; n = n1; Cs = C1; cc = c; }
            else { base = part2 + (int64_t)b * n2 * C2 * 2; n = n2; Cs = C2; cc = c - C1; }
            for (int t = tid; t < n; t += 256) {
                s += (double)base[((int64_t)t * Cs + cc) * 2];
                ss += (double)base[((int64_t)t * Cs + cc) * 2 + 1];
            }
        }
    } else {
        for (int t = tid; t < n1; t += 256) {
            const float* q = part1 + (((int64_t)b * n1 + t) * groups + g) * 2;
            s += (double)q[0];
            ss += (double)q[1];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = s; red[1][tid >> 6] = ss; }
    __syncthreads();
    if (tid == 0) {
        s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        ss = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        const double mean = s * inv_count;
        double var = ss * inv_count - mean * mean;
        if (var < 0.0) var = 0.0;
        mr[0] = (float)mean;
        mr[1] = (float)var;
        if (stats_out != nullptr) {
            stats_out[((int64_t)b * groups + g) * 2] = (float)mean;
            stats_out[((int64_t)b * groups + g) * 2 + 1] = (float)var;
        }
    }
    __syncthreads();
    if (ab != nullptr) {
        const float mean = mr[0], rstd = 1.0f / sqrtf(mr[1] + eps);
        for (int c = g * gs + tid; c < (g + 1) * gs; c += 256) {
            const float a = (gamma ? gamma[c] : 1.f) * rstd;
            ab[((int64_t)b * C + c) * 2] = a;
            ab[((int64_t)b * C + c) * 2 + 1] = (beta ? beta[c] : 0.f) - mean * a;
        }
    }
}

// pass 3: y = act((x-mean)*rstd*gamma+beta) [*(1+scale1p)+shift]
// grid (row-chunks, B).  Same thread <-> channel-chunk mapping as pass 1: a thread keeps ONE
// 8-channel chunk, so its 8 (scale, shift) pairs live in registers and the row loop is pure
// 16-byte streaming with 4 loads in flight.
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x1, const T* __restrict__ x2,
                                                       T* __restrict__ y, const float* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const T* __restrict__ mod_scale, const T* __restrict__ mod_shift,
                                                       int HW, int C1, int C2, int groups, float eps, int silu,
                                                       int rows_per_block, int mod_stride) {
    const int C = C1 + C2, C8 = C >> 3, C1_8 = C1 >> 3;
    const int TPR = C8 < 256 ? C8 : 256;
    const int rif = 256 / TPR;
    const int tid = threadIdx.x;
    const int tc = tid % TPR, rsub = tid / TPR;
    if (rsub >= rif) return;
    const int b = blockIdx.y;
    const int gs = C / groups;
    const int row_lo = blockIdx.x * rows_per_block;
    const int row_hi = min(HW, row_lo + rows_per_block);
    for (int cc = tc; cc < C8; cc += TPR) {
        float sa[8], sb[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = cc * 8 + e;
            const int g = ch / gs;
            const float mean = stats[((int64_t)b * groups + g) * 2];
            const float var = stats[((int64_t)b * groups + g) * 2 + 1];
            const float a = (gamma ? gamma[ch] : 1.f) / sqrtf(var + eps);
            sa[e] = a;
            sb[e] = (beta ? beta[ch] : 0.f) - mean * a;
        }
        const T* src;
        int64_t cstride;
        int coff;
        if (cc < C1_8) { src = x1 + (int64_t)b * HW * C1; cstride = C1; coff = cc * 8; }
        else { src = x2 + (int64_t)b * HW * C2; cstride = C2; coff = (cc - C1_8) * 8; }
        T* dst = y + (int64_t)b * HW * C + cc * 8;
        auto emit = [&](int r, const u32x4& v) {
            float f[8];
            unpack8<T>(v, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t = f[e] * sa[e] + sb[e];
                if (silu) t = silu_f(t);
                f[e] = t;
            }
            const int64_t o = (int64_t)r * C;
            if (mod_scale != nullptr) {
                const int64_t mo = ((int64_t)b * HW + r) * mod_stride + cc * 8;
                float ms[8], mh[8];
                unpack8<T>(*(const u32x4*)(mod_scale + mo), ms);
                unpack8<T>(*(const u32x4*)(mod_shift + mo), mh);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = f[e] * (1.f + ms[e]) + mh[e];
            }
            *(u32x4*)(dst + o) = pack8<T>(f);
        };
        int r = row_lo + rsub;
        for (; r + 3 * rif < row_hi; r += 4 * rif) {
            u32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *(const u32x4*)(src + (int64_t)(r + u * rif) * cstride + coff);
#pragma unroll
            for (int u = 0; u < 4; ++u) emit(r + u * rif, v[u]);
        }
        for (; r < row_hi; r += rif) emit(r, *(const u32x4*)(src + (int64_t)r * cstride + coff));
    }
}

// LayerNorm: one wave per row, ROWS rows per wave with all their loads issued up front (a single row per wave
// is one dependent load -> reduce -> store chain: measured 15 us for 10 MB); rows kept in registers (C <= 4096),
// exact two-pass variance.
template <typename T, int MAXC, int ROWS>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int64_t rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int C8 = C >> 3;
    // gamma / beta of this lane's chunks stay in registers for every row the wave processes (read per element inside
    // the row loop they doubled the L1 traffic of the kernel: 2.2 TB/s on the 168 MB token tensors of Stage 2)
    float ga[MAXC][8], be[MAXC][8];
#pragma unroll
    for (int j = 0; j < MAXC; ++j) {
        const int cc = lane + 64 * j;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ga[j][e] = (gamma != nullptr && cc < C8) ? gamma[cc * 8 + e] : 1.f;
            be[j][e] = (beta != nullptr && cc < C8) ? beta[cc * 8 + e] : 0.f;
        }
    }
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t row0 = wave_id * ROWS; row0 < rows; row0 += nwaves * ROWS) {
        u32x4 raw[ROWS][MAXC];
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
            for (int j = 0; j < MAXC; ++j) {
                const int cc = lane + 64 * j;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (cc < C8 && row0 + r < rows) v = *(const u32x4*)(x + (row0 + r) * C + cc * 8);
                raw[r][j] = v;
            }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            if (row0 + r >= rows) break;
            float f[MAXC][8];
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < MAXC; ++j) {
                unpack8<T>(raw[r][j], f[j]);
                if (lane + 64 * j < C8) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) s += f[j][e];
                }
            }
            const float mean = wave_sum(s) / (float)C;
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < MAXC; ++j) {
                if (lane + 64 * j < C8) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float d = f[j][e] - mean; ss += d * d; }
                }
            }
            const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)C + eps);
#pragma unroll
            for (int j = 0; j < MAXC; ++j) {
                const int cc = lane + 64 * j;
                if (cc < C8) {
                    float o[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (f[j][e] - mean) * rstd * ga[j][e] + be[j][e];
                    *(u32x4*)(y + (row0 + r) * C + cc * 8) = pack8<T>(o);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Small tensors (deep UNet levels: 32x32 / 64x64 maps): the three launches above are latency-bound (3 x ~7 us for
// 4 MB).  One workgroup per (image, group) instead: the group's HW x (C/groups) slab is read ONCE into registers
// (<= 32 sixteen-byte vectors per thread), reduced in fp64 through LDS in a fixed order, and either normalised and
// written (APPLY) or turned into the per-channel (scale, shift) rows of the fused conv prologue (!APPLY).
// Needs 8 | C/groups (a vector never straddles groups) and, for [x | x2], groups that do not straddle the sources.
// ---------------------------------------------------------------------------------------
constexpr int GN_SMALL_MAXV = 32;

template <typename T, bool APPLY>
__global__ __launch_bounds__(256) void gn_small_kernel(const T* __restrict__ x1, const T* __restrict__ x2, T* __restrict__ y,
                                                       float* __restrict__ ab, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int HW, int C1, int C2, int groups,
                                                       float eps, int silu) {
    __shared__ double red[2][4];
    __shared__ float mr[2];
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int C = C1 + C2, gs = C / groups, gs8 = gs >> 3;   // gs8 is a power of two <= 256
    const int ch0 = g * gs;
    const T* src;
    int cs, coff;
    if (ch0 < C1) { src = x1 + (int64_t)b * HW * C1; cs = C1; coff = ch0; }
    else { src = x2 + (int64_t)b * HW * C2; cs = C2; coff = ch0 - C1; }
    const int chunk = tid & (gs8 - 1);           // fixed per thread: 256 is a multiple of gs8
    const int row0 = tid / gs8, rstep = 256 / gs8;
    const int nv = (HW - row0 + rstep - 1) / rstep;   // vectors of this thread (<= GN_SMALL_MAXV, may be <= 0)
    u32x4 v[GN_SMALL_MAXV];
#pragma unroll
    for (int i = 0; i < GN_SMALL_MAXV; ++i)
        if (i < nv) v[i] = *(const u32x4*)(src + (int64_t)(row0 + i * rstep) * cs + coff + chunk * 8);
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int i = 0; i < GN_SMALL_MAXV; ++i)
        if (i < nv) {
            float f[8];
            unpack8<T>(v[i], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) { s += f[e]; ss += f[e] * f[e]; }
        }
    double ds = (double)s, dss = (double)ss;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ds += __shfl_xor(ds, o); dss += __shfl_xor(dss, o); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = ds; red[1][tid >> 6] = dss; }
    __syncthreads();
    if (tid == 0) {
        ds = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        dss = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        const double inv_count = 1.0 / ((double)HW * (double)gs);
        const double mean = ds * inv_count;
        double var = dss * inv_count - mean * mean;
        if (var < 0.0) var = 0.0;
        mr[0] = (float)mean;
        mr[1] = (float)var;
    }
    __syncthreads();
    const float mean = mr[0], rstd = 1.0f / sqrtf(mr[1] + eps);
    if (!APPLY) {
        if (tid < gs) {
            const int c = ch0 + tid;
            const float a = (gamma ? gamma[c] : 1.f) * rstd;
            ab[((int64_t)b * C + c) * 2] = a;
            ab[((int64_t)b * C + c) * 2 + 1] = (beta ? beta[c] : 0.f) - mean * a;
        }
        return;
    }
    float sa[8], sb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = ch0 + chunk * 8 + e;
        sa[e] = (gamma ? gamma[c] : 1.f) * rstd;
        sb[e] = (beta ? beta[c] : 0.f) - mean * sa[e];
    }
    T* dst = y + (int64_t)b * HW * C + ch0 + chunk * 8;
#pragma unroll
    for (int i = 0; i < GN_SMALL_MAXV; ++i)
        if (i < nv) {
            float f[8];
            unpack8<T>(v[i], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = f[e] * sa[e] + sb[e];
                f[e] = silu ? silu_f(t) : t;
            }
            *(u32x4*)(dst + (int64_t)(row0 + i * rstep) * C) = pack8<T>(f);
        }
}


// ---------------------------------------------------------------------------------------
// fp32-input forms for the split-operand product path (round 4; RSVLD_SPLIT): the residual stream of a network in the "split"
// precision is fp32 NHWC, and every tensor that only feeds a matrix product leaves its producer as two bf16 planes per row,
// [lo(C) | hi(C)] with hi = bf16(v), lo = bf16(v - hi).  Same thread <-> 8-channel-chunk mapping and the same two-level
// deterministic reduction as the 16-bit kernels above; 32 bytes per lane and row.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void ld8f(const float* p, float (&f)[8]) {
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
    f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
}
// planes row: lo at [c], hi at [C + c]
__device__ __forceinline__ void st_planes8(bf16* row, int C, int c, const float (&f)[8]) {
    bf16x8 hv;
    float lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { hv[e] = (bf16)f[e]; lo[e] = f[e] - (float)hv[e]; }
    *(u32x4*)(row + c) = pack8<bf16>(lo);
    *(u32x4*)(row + C + c) = __builtin_bit_cast(u32x4, hv);
}

__global__ __launch_bounds__(256) void gn_partial_f32_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                             float* __restrict__ part, int HW, int C1, int C2, int groups,
                                                             int rows_per_chunk, int nchunks) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sm = (float*)smem_raw;  // [rif][C][2]
    const int C = C1 + C2, C8 = C >> 3, C1_8 = C1 >> 3;
    const int TPR = C8 < 256 ? C8 : 256, rif = 256 / TPR;
    const int tid = threadIdx.x, chunk = blockIdx.x, b = blockIdx.y;
    const int row_lo = chunk * rows_per_chunk, row_hi = min(HW, row_lo + rows_per_chunk);
    const int tc = tid % TPR, rsub = tid / TPR;
    if (rsub < rif) {
        for (int cc = tc; cc < C8; cc += TPR) {
            float s[8], ss[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { s[e] = 0.f; ss[e] = 0.f; }
            const float* src;
            int64_t cstride;
            int coff;
            if (cc < C1_8) { src = x1 + (int64_t)b * HW * C1; cstride = C1; coff = cc * 8; }
            else { src = x2 + (int64_t)b * HW * C2; cstride = C2; coff = (cc - C1_8) * 8; }
            int r = row_lo + rsub;
            for (; r + rif < row_hi; r += 2 * rif) {   // two rows (4 x 16 B) in flight
                float f0[8], f1[8];
                ld8f(src + (int64_t)r * cstride + coff, f0);
                ld8f(src + (int64_t)(r + rif) * cstride + coff, f1);
#pragma unroll
                for (int e = 0; e < 8; ++e) { s[e] += f0[e]; ss[e] += f0[e] * f0[e]; }
#pragma unroll
                for (int e = 0; e < 8; ++e) { s[e] += f1[e]; ss[e] += f1[e] * f1[e]; }
            }
            for (; r < row_hi; r += rif) {
                float f[8];
                ld8f(src + (int64_t)r * cstride + coff, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) { s[e] += f[e]; ss[e] += f[e] * f[e]; }
            }
            float* dst = sm + ((int64_t)rsub * C + cc * 8) * 2;
#pragma unroll
            for (int e = 0; e < 8; ++e) { dst[2 * e] = s[e]; dst[2 * e + 1] = ss[e]; }
        }
    }
    __syncthreads();
    const int gs = C / groups;
    for (int g = tid; g < groups; g += 256) {
        float s = 0.f, ss = 0.f;
        for (int r = 0; r < rif; ++r) {
            const float* src = sm + ((int64_t)r * C + g * gs) * 2;
            for (int e = 0; e < gs; ++e) { s += src[2 * e]; ss += src[2 * e + 1]; }
        }
        float* o = part + (((int64_t)b * nchunks + chunk) * groups + g) * 2;
        o[0] = s;
        o[1] = ss;
    }
}

// y = act(a[b,c] * v + s[b,c]) [* (1 + mod_scale) + mod_shift]; fp32 in (one or two sources); OUT: 0 planes, 1 fp32, 2 fp16, 3 RSVLD_F16Q8 rows
template <int OUT>
__global__ __launch_bounds__(256) void gn_apply_split_kernel(const float* __restrict__ x1, const float* __restrict__ x2, void* __restrict__ y,
                                                             const float* __restrict__ ab, const float* __restrict__ mod_scale,
                                                             const float* __restrict__ mod_shift, int HW, int C1, int C2, int silu,
                                                             int rows_per_block, int mod_stride) {
    const int C = C1 + C2, C8 = C >> 3, C1_8 = C1 >> 3;
    const int TPR = C8 < 256 ? C8 : 256, rif = 256 / TPR;
    const int tid = threadIdx.x, tc = tid % TPR, rsub = tid / TPR;
    if (rsub >= rif) return;
    const int b = blockIdx.y;
    const int row_lo = blockIdx.x * rows_per_block, row_hi = min(HW, row_lo + rows_per_block);
    for (int cc = tc; cc < C8; cc += TPR) {
        float sa[8], sb[8];
        {
            const float* a = ab + ((int64_t)b * C + cc * 8) * 2;
#pragma unroll
            for (int e = 0; e < 8; ++e) { sa[e] = a[2 * e]; sb[e] = a[2 * e + 1]; }
        }
        const float* src;
        int64_t cstride;
        int coff;
        if (cc < C1_8) { src = x1 + (int64_t)b * HW * C1; cstride = C1; coff = cc * 8; }
        else { src = x2 + (int64_t)b * HW * C2; cstride = C2; coff = (cc - C1_8) * 8; }
        auto emit = [&](int r, float (&f)[8]) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = f[e] * sa[e] + sb[e];
                f[e] = silu ? silu_f(t) : t;
            }
            if (mod_scale != nullptr) {
                const int64_t mo = ((int64_t)b * HW + r) * mod_stride + cc * 8;
                float ms[8], mh[8];
                ld8f(mod_scale + mo, ms);
                ld8f(mod_shift + mo, mh);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = f[e] * (1.f + ms[e]) + mh[e];
            }
            const int64_t pix = (int64_t)b * HW + r;
            if (OUT == 1) {
                float* o = (float*)y + pix * C + cc * 8;
                *(f32x4*)o = (f32x4){f[0], f[1], f[2], f[3]};
                *(f32x4*)(o + 4) = (f32x4){f[4], f[5], f[6], f[7]};
            } else if (OUT == 2) {
                *(u32x4*)((f16*)y + pix * C + cc * 8) = pack8<f16>(f);
            } else if (OUT == 3) {
                st_hq8<true, RSVLD_HQ8_SX_LO, RSVLD_HQ8_SX_HI>((f16*)y + pix * (2 * (int64_t)C), C, cc * 8, f);
            } else {
                st_planes8((bf16*)y + pix * (2 * C), C, cc * 8, f);
            }
        };
        int r = row_lo + rsub;
        for (; r + rif < row_hi; r += 2 * rif) {
            float f0[8], f1[8];
            ld8f(src + (int64_t)r * cstride + coff, f0);
            ld8f(src + (int64_t)(r + rif) * cstride + coff, f1);
            emit(r, f0);
            emit(r + rif, f1);
        }
        for (; r < row_hi; r += rif) {
            float f[8];
            ld8f(src + (int64_t)r * cstride + coff, f);
            emit(r, f);
        }
    }
}

// LayerNorm of fp32 rows -> planes (or fp32): the 16-bit kernel's structure (one wave per row group, rows and the lane's
// gamma / beta in registers, exact two-pass variance) with 32-byte pieces
template <int MAXC, int ROWS, int OUT>
__global__ __launch_bounds__(256) void layernorm_split_kernel(const float* __restrict__ x, void* __restrict__ y,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              int64_t rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int C8 = C >> 3;
    float ga[MAXC][8], be[MAXC][8];
#pragma unroll
    for (int j = 0; j < MAXC; ++j) {
        const int cc = lane + 64 * j;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ga[j][e] = (gamma != nullptr && cc < C8) ? gamma[cc * 8 + e] : 1.f;
            be[j][e] = (beta != nullptr && cc < C8) ? beta[cc * 8 + e] : 0.f;
        }
    }
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t row0 = wave_id * ROWS; row0 < rows; row0 += nwaves * ROWS) {
        float f[ROWS][MAXC][8];
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
            for (int j = 0; j < MAXC; ++j) {
                const int cc = lane + 64 * j;
                if (cc < C8 && row0 + r < rows) ld8f(x + (row0 + r) * C + cc * 8, f[r][j]);
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[r][j][e] = 0.f;
                }
            }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            if (row0 + r >= rows) break;
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < MAXC; ++j)
                if (lane + 64 * j < C8) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) s += f[r][j][e];
                }
            const float mean = wave_sum(s) / (float)C;
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < MAXC; ++j)
                if (lane + 64 * j < C8) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float d = f[r][j][e] - mean; ss += d * d; }
                }
            const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)C + eps);
#pragma unroll
            for (int j = 0; j < MAXC; ++j) {
                const int cc = lane + 64 * j;
                if (cc < C8) {
                    float o[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (f[r][j][e] - mean) * rstd * ga[j][e] + be[j][e];
                    if (OUT == 1) {
                        float* d = (float*)y + (row0 + r) * C + cc * 8;
                        *(f32x4*)d = (f32x4){o[0], o[1], o[2], o[3]};
                        *(f32x4*)(d + 4) = (f32x4){o[4], o[5], o[6], o[7]};
                    } else if (OUT == 2) {
                        *(u32x4*)((f16*)y + (row0 + r) * C + cc * 8) = pack8<f16>(o);
                    } else {
                        st_planes8((bf16*)y + (row0 + r) * (2 * (int64_t)C), C, cc * 8, o);
                    }
                }
            }
        }
    }
}

// (mean, biased variance) per (image, group) -> the per-channel affine (gamma rstd, beta - mean gamma rstd)
__global__ void gn_ab_from_stats_kernel(const float* __restrict__ mean_var, const float* __restrict__ gamma, const float* __restrict__ beta,
                                        float* __restrict__ ab, int C, int groups, float eps, int total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // (b, c)
    if (i >= total) return;
    const int b = i / C, c = i - b * C, g = c / (C / groups);
    const float mean = mean_var[((int64_t)b * groups + g) * 2], var = mean_var[((int64_t)b * groups + g) * 2 + 1];
    const float a = (gamma ? gamma[c] : 1.f) * (1.0f / sqrtf(var + eps));
    ab[2 * (int64_t)i] = a;
    ab[2 * (int64_t)i + 1] = (beta ? beta[c] : 0.f) - mean * a;
}

bool gn_small_ok(int B, int HW, int C1, int C2, int groups) {
    const int C = C1 + C2, gs = C / groups;
    if (gs % 8) return false;
    const int gs8 = gs / 8;
    if (gs8 > 256 || (gs8 & (gs8 - 1))) return false;
    if (C2 > 0 && C1 % gs) return false;
    if ((int64_t)HW * gs8 > (int64_t)256 * GN_SMALL_MAXV) return false;
    (void)B;                   // (the choice must not depend on the batch: batch-invariant results)
    return groups >= 32;       // enough workgroups per image to be worth one launch
}

struct GnPlan {
    int nchunks, rows_per_chunk;
};
GnPlan gn_plan(int B, int HW) {
    (void)B;   // the chunking fixes the order of the fp32 partial sums: it depends on the image size only, so an image's
               // statistics are bit-identical whatever the batch around it
    const int max_chunks = 512;
    int rpc = (HW + max_chunks - 1) / max_chunks;
    if (rpc < 64) rpc = 64;
    GnPlan p;
    p.rows_per_chunk = rpc;
    p.nchunks = (HW + rpc - 1) / rpc;
    return p;
}

bool gn_shape_ok(int B, int HW, int C1, int C2, int groups) {
    if (B <= 0 || HW <= 0 || C1 <= 0 || C1 % 8 || C2 < 0 || C2 % 8 || groups <= 0) return false;
    const int C = C1 + C2;
    if (C % groups) return false;
    if (C > 8192 || groups > 256) return false;
    return true;
}

template <typename T>
int gn_stats_impl(const void* x, const void* x2, float* stats, int B, int HW, int C1, int C2, int groups, void* ws,
                  hipStream_t s) {
    const GnPlan pl = gn_plan(B, HW);
    const int C = C1 + C2, C8 = C / 8;
    const int TPR = C8 < 256 ? C8 : 256, rif = 256 / TPR;
    const size_t smem = (size_t)rif * C * 2 * sizeof(float);
    float* part = (float*)ws;
    hipLaunchKernelGGL(gn_partial_kernel<T>, dim3(pl.nchunks, B), dim3(256), smem, s, (const T*)x, (const T*)x2, part,
                       HW, C1, C2, groups, pl.rows_per_chunk, pl.nchunks);
    const int total = B * groups;
    const double inv_count = 1.0 / ((double)HW * (double)(C / groups));
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((total + 3) / 4), dim3(256), 0, s, part, stats, groups, pl.nchunks,
                       inv_count, total);
    return rsvld_check_launch();
}

template <typename T>
int gn_apply_impl(const void* x, const void* x2, void* y, const float* stats, const float* gamma, const float* beta,
                  const void* mscale, const void* mshift, int mod_stride, int B, int HW, int C1, int C2, int groups,
                  float eps, int silu, hipStream_t s) {
    const int C = C1 + C2, C8 = C / 8;
    const int TPR = C8 < 256 ? C8 : 256, rif = 256 / TPR;
    // ~2048 blocks over the chip, at least 4*rif rows per block so the unrolled loop is used
    int max_blocks = 2048 / (B > 0 ? B : 1);
    if (max_blocks < 1) max_blocks = 1;
    int rpb = (HW + max_blocks - 1) / max_blocks;
    if (rpb < 4 * rif) rpb = 4 * rif;
    const int nblk = (HW + rpb - 1) / rpb;
    hipLaunchKernelGGL(gn_apply_kernel<T>, dim3((unsigned)nblk, B), dim3(256), 0, s, (const T*)x, (const T*)x2,
                       (T*)y, stats, gamma, beta, (const T*)mscale, (const T*)mshift, HW, C1, C2, groups, eps, silu,
                       rpb, mod_stride > 0 ? mod_stride : C);
    return rsvld_check_launch();
}

}  // namespace

extern "C" int64_t rsvld_groupnorm_ws_bytes(int B, int HW, int C, int groups) {
    (void)C;
    if (B <= 0 || HW <= 0 || groups <= 0) return 0;
    const GnPlan pl = gn_plan(B, HW);
    // partials + final stats
    return ((int64_t)B * pl.nchunks * groups * 2 + (int64_t)B * groups * 2) * (int64_t)sizeof(float);
}

extern "C" int rsvld_groupnorm_stats(const void* x, const void* x2, float* mean_var, int B, int HW, int C1, int C2,
                                     int groups, int dtype, void* ws, void* stream) {
    if (!x || !mean_var || !ws || !gn_shape_ok(B, HW, C1, C2, groups) || ((C2 > 0) != (x2 != nullptr))) return RSVLD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == RSVLD_F16) return gn_stats_impl<f16>(x, x2, mean_var, B, HW, C1, C2, groups, ws, s);
    if (dtype == RSVLD_BF16) return gn_stats_impl<bf16>(x, x2, mean_var, B, HW, C1, C2, groups, ws, s);
    return RSVLD_EINVAL;
}

extern "C" int rsvld_groupnorm_apply(const void* x, const void* x2, void* y, const float* mean_var,
                                     const float* gamma, const float* beta, const void* mod_scale1p,
                                     const void* mod_shift, int mod_stride, int B, int HW, int C1, int C2, int groups,
                                     float eps, int silu, int dtype, void* stream) {
    if (!x || !y || !mean_var || !gn_shape_ok(B, HW, C1, C2, groups) || ((C2 > 0) != (x2 != nullptr))) return RSVLD_EINVAL;
    if ((mod_scale1p != nullptr) != (mod_shift != nullptr) || mod_stride < 0 || (mod_stride & 7)) return RSVLD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == RSVLD_F16)
        return gn_apply_impl<f16>(x, x2, y, mean_var, gamma, beta, mod_scale1p, mod_shift, mod_stride, B, HW, C1, C2, groups, eps, silu, s);
    if (dtype == RSVLD_BF16)
        return gn_apply_impl<bf16>(x, x2, y, mean_var, gamma, beta, mod_scale1p, mod_shift, mod_stride, B, HW, C1, C2, groups, eps, silu, s);
    return RSVLD_EINVAL;
}

extern "C" int rsvld_groupnorm_nhwc(const void* x, const void* x2, void* y, const float* gamma, const float* beta,
                                    const void* mod_scale1p, const void* mod_shift, int mod_stride, int B, int HW, int C1,
                                    int C2, int groups, float eps, int silu, int dtype, void* ws, void* stream) {
    if (!ws) return RSVLD_EINVAL;
    if (!gn_shape_ok(B, HW, C1, C2, groups)) return RSVLD_EINVAL;
    if (mod_scale1p == nullptr && mod_shift == nullptr && x && y && ((C2 > 0) == (x2 != nullptr)) && B <= 65535 &&
        gn_small_ok(B, HW, C1, C2, groups) && (dtype == RSVLD_F16 || dtype == RSVLD_BF16)) {
        hipStream_t s = (hipStream_t)stream;
        if (dtype == RSVLD_F16)
            hipLaunchKernelGGL((gn_small_kernel<f16, true>), dim3(groups, B), dim3(256), 0, s, (const f16*)x, (const f16*)x2,
                               (f16*)y, nullptr, gamma, beta, HW, C1, C2, groups, eps, silu);
        else
            hipLaunchKernelGGL((gn_small_kernel<bf16, true>), dim3(groups, B), dim3(256), 0, s, (const bf16*)x, (const bf16*)x2,
                               (bf16*)y, nullptr, gamma, beta, HW, C1, C2, groups, eps, silu);
        return rsvld_check_launch();
    }
    const GnPlan pl = gn_plan(B, HW);
    float* stats = (float*)ws + (int64_t)B * pl.nchunks * groups * 2;
    int rc = rsvld_groupnorm_stats(x, x2, stats, B, HW, C1, C2, groups, dtype, ws, stream);
    if (rc != RSVLD_OK) return rc;
    return rsvld_groupnorm_apply(x, x2, y, stats, gamma, beta, mod_scale1p, mod_shift, mod_stride, B, HW, C1, C2, groups,
                                 eps, silu, dtype, stream);
}

extern "C" int rsvld_groupnorm_scale_shift(const void* x, const void* x2, const float* gamma, const float* beta,
                                           float* scale_shift, int B, int HW, int C1, int C2, int groups, float eps,
                                           int dtype, void* ws, void* stream) {
    if (!x || !ws || !scale_shift || !gn_shape_ok(B, HW, C1, C2, groups) || ((C2 > 0) != (x2 != nullptr))) return RSVLD_EINVAL;
    if (dtype != RSVLD_F16 && dtype != RSVLD_BF16) return RSVLD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (B <= 65535 && gn_small_ok(B, HW, C1, C2, groups)) {
        if (dtype == RSVLD_F16)
            hipLaunchKernelGGL((gn_small_kernel<f16, false>), dim3(groups, B), dim3(256), 0, s, (const f16*)x, (const f16*)x2,
                               (f16*)nullptr, scale_shift, gamma, beta, HW, C1, C2, groups, eps, 0);
        else
            hipLaunchKernelGGL((gn_small_kernel<bf16, false>), dim3(groups, B), dim3(256), 0, s, (const bf16*)x, (const bf16*)x2,
                               (bf16*)nullptr, scale_shift, gamma, beta, HW, C1, C2, groups, eps, 0);
        return rsvld_check_launch();
    }
    const GnPlan pl = gn_plan(B, HW);
    const int C = C1 + C2, C8 = C / 8;
    const int TPR = C8 < 256 ? C8 : 256, rif = 256 / TPR;
    const size_t smem = (size_t)rif * C * 2 * sizeof(float);
    float* part = (float*)ws;
    if (dtype == RSVLD_F16)
        hipLaunchKernelGGL(gn_partial_kernel<f16>, dim3(pl.nchunks, B), dim3(256), smem, s, (const f16*)x, (const f16*)x2, part, HW, C1, C2, groups, pl.rows_per_chunk, pl.nchunks);
    else
        hipLaunchKernelGGL(gn_partial_kernel<bf16>, dim3(pl.nchunks, B), dim3(256), smem, s, (const bf16*)x, (const bf16*)x2, part, HW, C1, C2, groups, pl.rows_per_chunk, pl.nchunks);
    const double inv_count = 1.0 / ((double)HW * (double)(C / groups));
    hipLaunchKernelGGL(gn_ab_kernel<false>, dim3(groups, B), dim3(256), 0, s, part, pl.nchunks, C, nullptr, 0, 0, gamma, beta,
                       scale_shift, nullptr, groups, eps, inv_count);
    return rsvld_check_launch();
}

extern "C" int rsvld_groupnorm_scale_shift_from_partials(const float* part1, int ntiles1, int C1, const float* part2,
                                                         int ntiles2, int C2, const float* gamma, const float* beta,
                                                         float* scale_shift, int B, int HW, int groups, float eps,
                                                         void* stream) {
    if (!part1 || !scale_shift || B <= 0 || HW <= 0 || C1 <= 0 || C2 < 0 || ntiles1 <= 0 || groups <= 0) return RSVLD_EINVAL;
    if ((C2 > 0) != (part2 != nullptr) || (C2 > 0 && ntiles2 <= 0) || (C1 + C2) % groups != 0 || B > 65535) return RSVLD_EINVAL;
    const double inv_count = 1.0 / ((double)HW * (double)((C1 + C2) / groups));
    hipLaunchKernelGGL(gn_ab_kernel<true>, dim3(groups, B), dim3(256), 0, (hipStream_t)stream, part1, ntiles1, C1, part2,
                       ntiles2, C2, gamma, beta, scale_shift, nullptr, groups, eps, inv_count);
    return rsvld_check_launch();
}

template <typename T>
static void launch_layernorm(const void* x, void* y, const float* gamma, const float* beta, int64_t rows, int C, float eps,
                             hipStream_t s) {
    const int chunks_per_lane = (C / 8 + 63) / 64;
    auto blocks = [&](int rows_per_block) {   // at most 4096 blocks: waves keep gamma / beta in registers over their rows
        const int64_t n = cdiv64(rows, rows_per_block);
        return (unsigned)(n < 4096 ? n : 4096);
    };
    if (chunks_per_lane <= 2) {        // C <= 1024: 4 rows per wave in flight
        hipLaunchKernelGGL((layernorm_kernel<T, 2, 4>), dim3(blocks(16)), dim3(256), 0, s, (const T*)x, (T*)y, gamma, beta, rows, C, eps);
    } else if (chunks_per_lane == 3) { // C <= 1536 (the 1280-channel transformer blocks): 3 rows per wave in flight
        hipLaunchKernelGGL((layernorm_kernel<T, 3, 3>), dim3(blocks(12)), dim3(256), 0, s, (const T*)x, (T*)y, gamma, beta, rows, C, eps);
    } else if (chunks_per_lane <= 4) { // C <= 2048
        hipLaunchKernelGGL((layernorm_kernel<T, 4, 2>), dim3(blocks(8)), dim3(256), 0, s, (const T*)x, (T*)y, gamma, beta, rows, C, eps);
    } else {
        hipLaunchKernelGGL((layernorm_kernel<T, 8, 1>), dim3(blocks(4)), dim3(256), 0, s, (const T*)x, (T*)y, gamma, beta, rows, C, eps);
    }
}

extern "C" int rsvld_layernorm(const void* x, void* y, const float* gamma, const float* beta, int64_t rows, int C,
                               float eps, int dtype, void* stream) {
    if (!x || !y || rows <= 0 || C <= 0 || C % 8 || C > 4096) return RSVLD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == RSVLD_F16) launch_layernorm<f16>(x, y, gamma, beta, rows, C, eps, s);
    else if (dtype == RSVLD_BF16) launch_layernorm<bf16>(x, y, gamma, beta, rows, C, eps, s);
    else return RSVLD_EINVAL;
    return rsvld_check_launch();
}

// ---- split-operand product path (fp32 NHWC in; see the kernels above)
extern "C" int rsvld_groupnorm_scale_shift_f32(const float* x, const float* x2, const float* gamma, const float* beta,
                                               float* scale_shift, int B, int HW, int C1, int C2, int groups, float eps, void* ws,
                                               void* stream) {
    if (!x || !ws || !scale_shift || !gn_shape_ok(B, HW, C1, C2, groups) || ((C2 > 0) != (x2 != nullptr)) || B > 65535) return RSVLD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const GnPlan pl = gn_plan(B, HW);
    const int C = C1 + C2, C8 = C / 8;
    const int TPR = C8 < 256 ? C8 : 256, rif = 256 / TPR;
    const size_t smem = (size_t)rif * C * 2 * sizeof(float);
    float* part = (float*)ws;
    hipLaunchKernelGGL(gn_partial_f32_kernel, dim3(pl.nchunks, B), dim3(256), smem, s, x, x2, part, HW, C1, C2, groups,
                       pl.rows_per_chunk, pl.nchunks);
    const double inv_count = 1.0 / ((double)HW * (double)(C / groups));
    hipLaunchKernelGGL(gn_ab_kernel<false>, dim3(groups, B), dim3(256), 0, s, part, pl.nchunks, C, nullptr, 0, 0, gamma, beta,
                       scale_shift, nullptr, groups, eps, inv_count);
    return rsvld_check_launch();
}

extern "C" int rsvld_groupnorm_apply_split(const float* x, const float* x2, void* out, const float* scale_shift,
                                           const float* mod_scale1p, const float* mod_shift, int mod_stride, int B, int HW, int C1,
                                           int C2, int silu, int out_f32, void* stream) {
    if (!x || !out || !scale_shift || B <= 0 || B > 65535 || HW <= 0 || C1 <= 0 || C1 % 8 || C2 < 0 || C2 % 8 || ((C2 > 0) != (x2 != nullptr)))
        return RSVLD_EINVAL;
    if ((mod_scale1p != nullptr) != (mod_shift != nullptr) || mod_stride < 0 || (mod_stride & 3)) return RSVLD_EINVAL;
    const int C = C1 + C2, C8 = C / 8;
    const int TPR = C8 < 256 ? C8 : 256, rif = 256 / TPR;
    int max_blocks = 2048 / B;
    if (max_blocks < 1) max_blocks = 1;
    int rpb = (HW + max_blocks - 1) / max_blocks;
    if (rpb < 2 * rif) rpb = 2 * rif;
    const int nblk = (HW + rpb - 1) / rpb;
    hipStream_t s = (hipStream_t)stream;
    if (out_f32 < 0 || out_f32 > 3 || (out_f32 == 3 && (C1 + C2) % 32)) return RSVLD_EINVAL;
    if (out_f32 == 3)
        hipLaunchKernelGGL(gn_apply_split_kernel<3>, dim3((unsigned)nblk, B), dim3(256), 0, s, x, x2, out, scale_shift, mod_scale1p,
                           mod_shift, HW, C1, C2, silu, rpb, mod_stride > 0 ? mod_stride : C);
    else if (out_f32 == 1)
        hipLaunchKernelGGL(gn_apply_split_kernel<1>, dim3((unsigned)nblk, B), dim3(256), 0, s, x, x2, out, scale_shift, mod_scale1p,
                           mod_shift, HW, C1, C2, silu, rpb, mod_stride > 0 ? mod_stride : C);
    else if (out_f32 == 2)
        hipLaunchKernelGGL(gn_apply_split_kernel<2>, dim3((unsigned)nblk, B), dim3(256), 0, s, x, x2, out, scale_shift, mod_scale1p,
                           mod_shift, HW, C1, C2, silu, rpb, mod_stride > 0 ? mod_stride : C);
    else
        hipLaunchKernelGGL(gn_apply_split_kernel<0>, dim3((unsigned)nblk, B), dim3(256), 0, s, x, x2, out, scale_shift, mod_scale1p,
                           mod_shift, HW, C1, C2, silu, rpb, mod_stride > 0 ? mod_stride : C);
    return rsvld_check_launch();
}

template <int OUT_F32>
static void launch_layernorm_split(const float* x, void* y, const float* gamma, const float* beta, int64_t rows, int C, float eps,
                                   hipStream_t s) {
    const int chunks_per_lane = (C / 8 + 63) / 64;
    // at most ~3 resident workgroups per CU of a 256-CU chip: a wave then walks several row groups and its gamma / beta rows (as many
    // bytes as two rows of x at C = 1 280) are loaded once per wave instead of once per two rows (round 5: 4 096 workgroups of one row
    // group each ran the 32 768 x 1 280 LayerNorms of Stage 2 at 2.5 TB/s)
    auto blocks = [&](int rows_per_block) {
        const int64_t n = cdiv64(rows, rows_per_block);
        return (unsigned)(n < 768 ? n : 768);
    };
    if (chunks_per_lane <= 2) hipLaunchKernelGGL((layernorm_split_kernel<2, 2, OUT_F32>), dim3(blocks(8)), dim3(256), 0, s, x, y, gamma, beta, rows, C, eps);
    else if (chunks_per_lane == 3) hipLaunchKernelGGL((layernorm_split_kernel<3, 2, OUT_F32>), dim3(blocks(8)), dim3(256), 0, s, x, y, gamma, beta, rows, C, eps);
    else if (chunks_per_lane <= 4) hipLaunchKernelGGL((layernorm_split_kernel<4, 1, OUT_F32>), dim3(blocks(4)), dim3(256), 0, s, x, y, gamma, beta, rows, C, eps);
    else hipLaunchKernelGGL((layernorm_split_kernel<8, 1, OUT_F32>), dim3(blocks(4)), dim3(256), 0, s, x, y, gamma, beta, rows, C, eps);
}

extern "C" int rsvld_layernorm_split(const float* x, void* out, const float* gamma, const float* beta, int64_t rows, int C, float eps,
                                     int out_f32, void* stream) {
    if (!x || !out || rows <= 0 || C <= 0 || C % 8 || C > 4096) return RSVLD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (out_f32 < 0 || out_f32 > 2) return RSVLD_EINVAL;
    if (out_f32 == 1) launch_layernorm_split<1>(x, out, gamma, beta, rows, C, eps, s);
    else if (out_f32 == 2) launch_layernorm_split<2>(x, out, gamma, beta, rows, C, eps, s);
    else launch_layernorm_split<0>(x, out, gamma, beta, rows, C, eps, s);
    return rsvld_check_launch();
}

extern "C" int rsvld_groupnorm_stats_f32_fast(const float* x, const float* x2, float* mean_var, int B, int HW, int C1, int C2, int groups,
                                              void* ws, void* stream) {
    if (!x || !mean_var || !ws || !gn_shape_ok(B, HW, C1, C2, groups) || ((C2 > 0) != (x2 != nullptr)) || B > 65535) return RSVLD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const GnPlan pl = gn_plan(B, HW);
    const int C = C1 + C2, C8 = C / 8;
    const int TPR = C8 < 256 ? C8 : 256, rif = 256 / TPR;
    const size_t smem = (size_t)rif * C * 2 * sizeof(float);
    float* part = (float*)ws;
    hipLaunchKernelGGL(gn_partial_f32_kernel, dim3(pl.nchunks, B), dim3(256), smem, s, x, x2, part, HW, C1, C2, groups,
                       pl.rows_per_chunk, pl.nchunks);
    const int total = B * groups;
    const double inv_count = 1.0 / ((double)HW * (double)(C / groups));
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((total + 3) / 4), dim3(256), 0, s, part, mean_var, groups, pl.nchunks, inv_count, total);
    return rsvld_check_launch();
}

extern "C" int rsvld_groupnorm_scale_shift_from_stats(const float* mean_var, const float* gamma, const float* beta, float* scale_shift,
                                                      int B, int C, int groups, float eps, void* stream) {
    if (!mean_var || !scale_shift || B <= 0 || C <= 0 || groups <= 0 || C % groups) return RSVLD_EINVAL;
    const int total = B * C;
    hipLaunchKernelGGL(gn_ab_from_stats_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, mean_var, gamma, beta,
                       scale_shift, C, groups, eps, total);
    return rsvld_check_launch();
}
