// split.hip — producers / consumers of bf16 PLANES around the RSVLD_SPLIT matrix kernels (round 4), and the split-operand
// flash attention for d = 64.
//
// The "split" precision (models/SR_model.py:28-33 run without autocast is what it has to match: the reference's CPU path is fp32):
// an fp32 value v is carried as hi = bf16(v), lo = bf16(v - hi) (16 mantissa bits, fp32's exponent range) and a product of two such
// numbers as three bf16 MFMAs, a_lo b_hi + a_hi b_lo + a_hi b_hi, into an fp32 accumulator.  Round 3 did the split on the fly inside
// the fp32 family's simple kernels; here the planes are DATA: a tensor that only feeds matrix products leaves its producer as
// [rows][lo(C) | hi(C)], weights are packed once as [W_hi | W_lo | W_hi], and the tuned 16-bit kernels (gemm.hip, conv_halo.hip,
// conv_igemm.hip) run the three terms as one contraction over 3 K.  This file holds what surrounds them:
//   split / merge / weight-triple packing, plane re-packs for attention run as GEMMs (single-head d = 512), the row softmax
//   that turns fp32 scores into P planes, and attn_split_d64_kernel -- flash attention on planes for the SDXL blocks.
#include "rsvld_common.h"
#include <type_traits>

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void ld8f(const float* p, float (&f)[8]) {
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
    f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
}
__device__ __forceinline__ void split8v(const float (&f)[8], u32x4& lo, u32x4& hi) {
    bf16x8 hv;
    float l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { hv[e] = (bf16)f[e]; l[e] = f[e] - (float)hv[e]; }
    lo = pack8<bf16>(l);
    hi = __builtin_bit_cast(u32x4, hv);
}

__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, bf16* __restrict__ y, int64_t items, int C8) {
    const int C = C8 * 8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / C8;
        const int c = (int)(i - row * C8) * 8;
        float f[8];
        ld8f(x + row * C + c, f);
        u32x4 lo, hi;
        split8v(f, lo, hi);
        *(u32x4*)(y + row * (2 * (int64_t)C) + c) = lo;
        *(u32x4*)(y + row * (2 * (int64_t)C) + C + c) = hi;
    }
}

__global__ __launch_bounds__(256) void merge_planes_kernel(const bf16* __restrict__ y, float* __restrict__ x, int64_t items, int C8) {
    const int C = C8 * 8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / C8;
        const int c = (int)(i - row * C8) * 8;
        float l[8], h[8];
        unpack8<bf16>(*(const u32x4*)(y + row * (2 * (int64_t)C) + c), l);
        unpack8<bf16>(*(const u32x4*)(y + row * (2 * (int64_t)C) + C + c), h);
        float* o = x + row * C + c;
        *(f32x4*)o = (f32x4){h[0] + l[0], h[1] + l[1], h[2] + l[2], h[3] + l[3]};
        *(f32x4*)(o + 4) = (f32x4){h[4] + l[4], h[5] + l[5], h[6] + l[6], h[7] + l[7]};
    }
}

// clamp to the finite fp16 range WITHOUT swallowing NaN: fminf / fmaxf lower to minnum / maxnum, which return the non-NaN operand
// (fmaxf(NaN, -65504) = -65504), so the clamp is a compare + select on |s| and NaN (all compares false) falls through unchanged
__device__ __forceinline__ float sat_f16_keep_nan(float s) {
    return fabsf(s) > 65504.f ? copysignf(65504.f, s) : s;
}

// planes (rows of lo | hi, the two planes `ps` elements apart, rows `ld` elements apart: a channel slice of a wider planes tensor
// is fine) -> fp16 [rows][C] (row stride old): the operands of the 16-bit attention kernels in the "split, attention in fp16" mode
__global__ __launch_bounds__(256) void planes_to_f16_kernel(const bf16* __restrict__ y, f16* __restrict__ o, int64_t items, int C8, int64_t ld,
                                                            int64_t ps, int64_t old) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / C8;
        const int c = (int)(i - row * C8) * 8;
        float l[8], h[8];
        unpack8<bf16>(*(const u32x4*)(y + row * ld + c), l);
        unpack8<bf16>(*(const u32x4*)(y + row * ld + ps + c), h);
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = sat_f16_keep_nan(h[e] + l[e]);   // saturate: a finite fp32 value stays finite, NaN stays NaN
        *(u32x4*)(o + row * old + c) = pack8<f16>(h);
    }
}

// fp16 [rows][C] (row stride ld) -> planes [rows][lo(C) | hi(C)]: exact (11 significant bits fit hi + lo)
__global__ __launch_bounds__(256) void f16_to_planes_kernel(const f16* __restrict__ x, bf16* __restrict__ y, int64_t items, int C8, int64_t ld) {
    const int C = C8 * 8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / C8;
        const int c = (int)(i - row * C8) * 8;
        float f[8];
        unpack8<f16>(*(const u32x4*)(x + row * ld + c), f);
        u32x4 lo, hi;
        split8v(f, lo, hi);
        *(u32x4*)(y + row * (2 * (int64_t)C) + c) = lo;
        *(u32x4*)(y + row * (2 * (int64_t)C) + C + c) = hi;
    }
}

// w [R][Ctot] fp32 (R = Cout * taps) -> [R][hi | lo | hi]
__global__ __launch_bounds__(256) void split_pack_weights_kernel(const float* __restrict__ w, bf16* __restrict__ o, int64_t items, int C8) {
    const int C = C8 * 8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / C8;
        const int c = (int)(i - row * C8) * 8;
        float f[8];
        ld8f(w + row * C + c, f);
        u32x4 lo, hi;
        split8v(f, lo, hi);
        bf16* d = o + row * (3 * (int64_t)C) + c;
        *(u32x4*)d = hi;
        *(u32x4*)(d + C) = lo;
        *(u32x4*)(d + 2 * C) = hi;
    }
}

// planes [rows][ld] (lo at c, hi at C + c) -> [rows_p][hi | lo | hi]; rows past `rows` are zero
__global__ __launch_bounds__(256) void planes_to_triple_kernel(const bf16* __restrict__ y, bf16* __restrict__ o, int64_t rows, int64_t items,
                                                               int C8, int64_t ld) {
    const int C = C8 * 8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / C8;
        const int c = (int)(i - row * C8) * 8;
        u32x4 lo = {0u, 0u, 0u, 0u}, hi = {0u, 0u, 0u, 0u};
        if (row < rows) {
            lo = *(const u32x4*)(y + row * ld + c);
            hi = *(const u32x4*)(y + row * ld + C + c);
        }
        bf16* d = o + row * (3 * (int64_t)C) + c;
        *(u32x4*)d = hi;
        *(u32x4*)(d + C) = lo;
        *(u32x4*)(d + 2 * C) = hi;
    }
}

// planes [rows][ld] -> transposed triple [C][hi^T(rows_p) | lo^T(rows_p) | hi^T(rows_p)]: 64 rows x 64 channels per workgroup
// through LDS (rows of 66 halfwords: the column walk of the store side hits distinct banks)
__global__ __launch_bounds__(256) void planes_transpose_triple_kernel(const uint16_t* __restrict__ y, uint16_t* __restrict__ o, int64_t rows,
                                                                      int64_t rows_p, int C, int64_t ld) {
    __shared__ uint16_t tl[2][64][66];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (tid >> 3) + 32 * i, c = (tid & 7) * 8;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (r0 + r < rows && c0 + c < C) v = *(const u32x4*)(y + (r0 + r) * ld + pl * C + c0 + c);
            uint16_t h[8];
            __builtin_memcpy(h, &v, 16);
#pragma unroll
            for (int e = 0; e < 8; ++e) tl[pl][r][c + e] = h[e];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = (tid >> 3) + 32 * i, r = (tid & 7) * 8;
        if (c0 + c >= C || r0 + r >= rows_p) continue;   // rows_p % 8 == 0: an 8-row piece is inside or outside as a whole
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            uint16_t h[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = tl[pl][r + e][c];
            u32x4 v;
            __builtin_memcpy(&v, h, 16);
            uint16_t* d = o + (int64_t)(c0 + c) * (3 * rows_p) + r0 + r;
            if (pl == 0) {
                *(u32x4*)(d + rows_p) = v;             // lo -> segment 1
            } else {
                *(u32x4*)d = v;                        // hi -> segments 0 and 2
                *(u32x4*)(d + 2 * rows_p) = v;
            }
        }
    }
}

// row softmax of fp32 scores -> P planes.  One workgroup of 256 threads per row; pass 1: per-thread online (max, sum) over 8-wide
// pieces, merged through LDS; pass 2: p = exp2((s - m) c) / l written as lo | hi.  Pad columns [cols, cols_p) are written as zeros.
__global__ __launch_bounds__(256) void softmax_rows_split_kernel(const float* __restrict__ s, bf16* __restrict__ pp, int cols, int cols_p,
                                                                 int64_t ld, float scale_log2e) {
    __shared__ float red[2][4];
    const int64_t row = blockIdx.x;
    const float* sr = s + row * ld;
    const int tid = threadIdx.x;
    float m = -INFINITY, l = 0.f;
    for (int c = tid * 8; c < cols; c += 2048) {
        float f[8];
        ld8f(sr + c, f);   // ld >= cols_p >= c + 8: inside the row
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            f[e] = (c + e < cols) ? f[e] * scale_log2e : -INFINITY;
            mx = fmaxf(mx, f[e]);
        }
        const float mn = fmaxf(m, mx);
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) a += __builtin_amdgcn_exp2f(f[e] - mn);
        l = l * __builtin_amdgcn_exp2f(m - mn) + a;
        m = mn;
    }
    // wave merge, then the four waves through LDS (fixed order)
    const float wm = wave_max(m);
    l = wave_sum(m == -INFINITY ? 0.f : l * __builtin_amdgcn_exp2f(m - wm));
    if ((tid & 63) == 0) { red[0][tid >> 6] = wm; red[1][tid >> 6] = l; }
    __syncthreads();
    const float M = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    float L = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) L += red[0][w] == -INFINITY ? 0.f : red[1][w] * __builtin_amdgcn_exp2f(red[0][w] - M);
    const float inv = 1.0f / L;
    bf16* pr = pp + row * (2 * (int64_t)cols_p);
    for (int c = tid * 8; c < cols_p; c += 2048) {
        float f[8];
        ld8f(sr + c, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (c + e < cols) ? __builtin_amdgcn_exp2f(f[e] * scale_log2e - M) * inv : 0.f;
        u32x4 lo, hi;
        split8v(f, lo, hi);
        *(u32x4*)(pr + c) = lo;
        *(u32x4*)(pr + cols_p + c) = hi;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Flash attention on planes, d = 64 (the Stage-2 transformer blocks under diffusion_dtype "split").
//
// The structure of attn_d64b (attention.hip): 4 waves x 32 query rows per workgroup, S^T = K Q^T computed swapped so that the
// softmax is register-local and P is consumed as the next MFMA's B operand straight from the score registers, K / V tiles of 64
// keys by LDS-DMA into a double buffer, V read through the hardware transpose.  What changes: every tile has FOUR planes
// (K_lo, K_hi, V_lo, V_hi: 32 KiB per stage, two workgroups per CU), Q is held as hi + lo fragments (pre-multiplied by
// scale log2 e in fp32, then re-split), P is split in registers after the fp32 softmax, and both contractions issue three MFMAs per
// fragment pair, small terms first: 48 MFMAs per wave and tile against ~230 vector instructions -- the matrix pipe is the long side
// here (in the 16-bit kernel the vector port is), so the softmax is the plain exact online form and the MFMAs are compiler-scheduled
// builtins; two waves per SIMD give each other cover.
// ---------------------------------------------------------------------------------------------------------------------------
struct AttnSplitArgs {
    const bf16* q; const bf16* k; const bf16* v; void* out;
    int Nq, Nk;
    int64_t q_bs, q_ts, q_pl, k_bs, k_ts, k_pl, v_bs, v_ts, v_pl, o_bs, o_ts, o_pl;
    float scale_log2e;
    int out_f32;
};
constexpr int AS_TILE = 64 * 128;           // 64 keys x 64 d, bf16
constexpr int AS_SMEM = 8 * AS_TILE;        // (K_lo, K_hi, V_lo, V_hi) x 2 buffers

__device__ __forceinline__ f32x16 mma_bf16(const bf16x8& a, const bf16x8& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__global__ __launch_bounds__(256, 2) void attn_split_d64_kernel(AttnSplitArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int q0 = (blockIdx.x * 4 + w) * 32, h = blockIdx.y, b = blockIdx.z;
    const bf16* Qb = p.q + (int64_t)b * p.q_bs + (int64_t)h * 64;
    const bf16* Kb = p.k + (int64_t)b * p.k_bs + (int64_t)h * 64;
    const bf16* Vb = p.v + (int64_t)b * p.v_bs + (int64_t)h * 64;
    const int nt = (p.Nk + 63) >> 6;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;

    // ---- tile DMA (inline asm: a DMA the compiler knows of makes it drain vmcnt in front of the transposed V reads of the SAME
    // iteration; M0 saved / restored inside the statement).  Wave w moves key rows 16 w .. 16 w + 15 of each of the four planes, two
    // pieces of 8 rows x 128 B each = 8 pieces per tile.  Lane (row = lane>>3, pos = lane&7) fills LDS chunk `pos` of its row with
    // source chunk pos ^ ((r16>>1)&7) for K and pos ^ (((r16>>1)&1)<<2) for V (r16 = tile row & 15): the images the fragment
    // reads below expect (attention.hip).  Full tiles: one scalar base per tensor + a 32-bit lane offset.
    auto dma_one = [&](const char* base, uint32_t voff, uint32_t dst) {
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(dst), "s"(base) : "memory");
    };
    const int64_t k_rowb = p.k_ts * (int64_t)sizeof(bf16), v_rowb = p.v_ts * (int64_t)sizeof(bf16);
    const int64_t k_plb = p.k_pl * (int64_t)sizeof(bf16), v_plb = p.v_pl * (int64_t)sizeof(bf16);
    uint32_t kvo[2], vvo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rr = 8 * i + (lane >> 3), r16 = (16 * w + rr) & 15;
        kvo[i] = (uint32_t)(rr * k_rowb) + (uint32_t)(((lane & 7) ^ ((r16 >> 1) & 7)) << 4);
        vvo[i] = (uint32_t)(rr * v_rowb) + (uint32_t)(((lane & 7) ^ (((r16 >> 1) & 1) << 2)) << 4);
    }
    const char* k_tile0 = (const char*)(Kb + (int64_t)(16 * w) * p.k_ts);
    const char* v_tile0 = (const char*)(Vb + (int64_t)(16 * w) * p.v_ts);
    auto dma_piece = [&](int t, int j) {   // j = 0..7: tensor j >> 2 (K, V), plane (j >> 1) & 1 (lo, hi), row group j & 1
        const int tens = j >> 2, pl = (j >> 1) & 1, i = j & 1;
        const uint32_t dst = lds0 + (4 * tens + 2 * pl + (t & 1)) * AS_TILE + (16 * w + 8 * i) * 128;
        if (t * 64 + 64 <= p.Nk) {
            if (tens == 0) dma_one(k_tile0 + (int64_t)t * 64 * k_rowb + pl * k_plb, kvo[i], dst);
            else dma_one(v_tile0 + (int64_t)t * 64 * v_rowb + pl * v_plb, vvo[i], dst);
        } else {   // ragged last tile: rows past Nk re-read the last key (their scores are masked); per-lane 64-bit addresses
            const int rr = 8 * i + (lane >> 3), r16 = (16 * w + rr) & 15;
            const int key = min(t * 64 + 16 * w + rr, p.Nk - 1);
            const char* src = tens == 0 ? (const char*)(Kb + (int64_t)key * p.k_ts) + pl * k_plb + (((lane & 7) ^ ((r16 >> 1) & 7)) << 4)
                                        : (const char*)(Vb + (int64_t)key * p.v_ts) + pl * v_plb + (((lane & 7) ^ (((r16 >> 1) & 1) << 2)) << 4);
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        }
    };
#pragma unroll
    for (int j = 0; j < 8; ++j) dma_piece(0, j);

    // ---- Q fragments (B operand: column = the lane's query row, k = d), scaled in fp32 and re-split
    const int qrow = q0 + l31;
    bf16x8 qh[4], ql[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = 0.f;
        if (qrow < p.Nq) {
            const bf16* qp = Qb + (int64_t)qrow * p.q_ts + ks * 16 + lh * 8;
            float l[8], hh[8];
            unpack8<bf16>(*(const u32x4*)qp, l);
            unpack8<bf16>(*(const u32x4*)(qp + p.q_pl), hh);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (hh[e] + l[e]) * p.scale_log2e;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            qh[ks][e] = (bf16)f[e];
            ql[ks][e] = (bf16)(f[e] - (float)qh[ks][e]);
        }
    }
    f32x16 oacc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = l31 * 128 + (((2 * ks + lh) ^ ((l31 >> 1) & 7)) << 4);
    // V (transposed read): lane 16g + 4q + p supplies row 16 s4 + 8 hf + 4 lh + q, d = 32 dt + 16 (g&1) + 4p .. +3
    int voff[2];
    {
        const int qq = (lane >> 2) & 3, pp = lane & 3, g1 = (lane >> 4) & 1;
        const int row = 4 * lh + qq;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
            voff[dt] = row * 128 + (((4 * dt + 2 * g1 + (pp >> 1)) ^ (((row >> 1) & 1) << 2)) << 4) + ((pp & 1) << 3);
    }
    auto read_vt = [&](const char* base, int off) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off + 1024));
        typedef short s16x8 __attribute__((__vector_size__(8 * sizeof(short))));
        return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // tile 0 landed
    __builtin_amdgcn_sched_barrier(0);

    // The tile loop exists twice: a steady-state copy for tiles whose successor exists and is FULL (no test around the 8 LDS-DMA
    // pieces, no ragged mask, the next tile's row bases one scalar add away) and the general copy for the last tiles.
    const char* k_nxt = k_tile0;
    const char* v_nxt = v_tile0;
    auto body = [&](int t, auto steady_c) __attribute__((always_inline)) {
        constexpr bool ST = decltype(steady_c)::value;
        k_nxt += 64 * k_rowb;   // row 16 w of tile t + 1
        v_nxt += 64 * v_rowb;
        const int buf = t & 1;
        const bool more = ST ? true : t + 1 < nt;
        const char* Kl = smem + (0 + buf) * AS_TILE;
        const char* Kh = smem + (2 + buf) * AS_TILE;
        const char* Vl = smem + (4 + buf) * AS_TILE;
        const char* Vh = smem + (6 + buf) * AS_TILE;

        // ---- S^T[key][q] = K Q^T: two 32-key halves x 4 k-steps x 3 terms.  The fragments of step n + 1 are requested before the
        // MFMAs of step n, and one of the 8 LDS-DMA pieces of tile t + 1 is issued behind every step (the other buffer was last
        // read in iteration t - 1, behind its closing barrier).
        f32x16 sacc[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kt][r] = 0.f;
        bf16x8 kl[2], kh[2];
        kl[0] = *(const bf16x8*)(Kl + koff[0]);
        kh[0] = *(const bf16x8*)(Kh + koff[0]);
#pragma unroll
        for (int n = 0; n < 8; ++n) {   // n = 4 kt + ks
            if (n + 1 < 8) {
                const int o = ((n + 1) >> 2) * 4096 + koff[(n + 1) & 3];
                kl[(n + 1) & 1] = *(const bf16x8*)(Kl + o);
                kh[(n + 1) & 1] = *(const bf16x8*)(Kh + o);
            }
            sacc[n >> 2] = mma_bf16(kl[n & 1], qh[n & 3], sacc[n >> 2]);
            sacc[n >> 2] = mma_bf16(kh[n & 1], ql[n & 3], sacc[n >> 2]);
            sacc[n >> 2] = mma_bf16(kh[n & 1], qh[n & 3], sacc[n >> 2]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (ST) {   // piece n: tensor n >> 2, plane (n >> 1) & 1, row group n & 1
                const uint32_t dst = lds0 + (4 * (n >> 2) + 2 * ((n >> 1) & 1) + ((t + 1) & 1)) * AS_TILE + (16 * w + 8 * (n & 1)) * 128;
                if ((n >> 2) == 0) dma_one(k_nxt + ((n >> 1) & 1) * k_plb, kvo[n & 1], dst);
                else dma_one(v_nxt + ((n >> 1) & 1) * v_plb, vvo[n & 1], dst);
            } else if (more) dma_piece(t + 1, n);
            __builtin_amdgcn_sched_barrier(0);
        }
        // first V fragments of PV(t): in flight behind the softmax
        bf16x8 vl[2], vh[2];
        vl[0] = read_vt(Vl, voff[0]);
        vh[0] = read_vt(Vh, voff[0]);
        if (!ST && (t + 1) * 64 > p.Nk) {   // ragged last tile only (uniform branch)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kv = t * 64 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (kv >= p.Nk) sacc[kt][r] = -INFINITY;
                }
        }

        // ---- exact online softmax in fp32 (this lane: 32 of its query's 64 scores, lane ^ 32 the rest), deferred maximum
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[kt][r]);
        {
            const uint32_t mb = __builtin_bit_cast(uint32_t, mx);
            const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
            mx = fmaxf(__builtin_bit_cast(float, (uint32_t)sw[0]), __builtin_bit_cast(float, (uint32_t)sw[1]));
        }
        float alpha = 1.0f;
        const bool need = mx > m_run + 8.0f;   // true on the first tile (m_run = -inf); P <= 2^8 otherwise: as good a bf16 pair as P <= 1
        if (need) {
            alpha = __builtin_amdgcn_exp2f(m_run - mx);
            m_run = mx;
        }
        float r4[4] = {0.f, 0.f, 0.f, 0.f};
        bf16x8 ph[4], pl[4];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(sacc[kt][r] - m_run);
                r4[r >> 2] += pv;
                const bf16 hv = (bf16)pv;
                ph[2 * kt + (r >> 3)][r & 7] = hv;
                pl[2 * kt + (r >> 3)][r & 7] = (bf16)(pv - (float)hv);
            }
        l_run = l_run * alpha + ((r4[0] + r4[1]) + (r4[2] + r4[3]));
        if (__builtin_expect(__any(need), 0)) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
        }

        // ---- O^T[d][q] += V^T P^T: 4 key steps x 2 d-halves x 3 terms, fragments one pair ahead.  A operand (row = d, k = key in
        // P's register order): elements 0..3 = keys 16 s4 + 4 lh + 0..3, elements 4..7 = keys 16 s4 + 8 + 4 lh + 0..3
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < 8; ++n) {   // n = 2 s4 + dt
            if (n + 1 < 8) {
                const int o = voff[(n + 1) & 1] + ((n + 1) >> 1) * 2048;
                vl[(n + 1) & 1] = read_vt(Vl, o);
                vh[(n + 1) & 1] = read_vt(Vh, o);
            }
            oacc[n & 1] = mma_bf16(vh[n & 1], pl[n >> 1], oacc[n & 1]);
            oacc[n & 1] = mma_bf16(vl[n & 1], ph[n >> 1], oacc[n & 1]);
            oacc[n & 1] = mma_bf16(vh[n & 1], ph[n >> 1], oacc[n & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // tile t + 1 landed; everyone done with this buffer
        __builtin_amdgcn_sched_barrier(0);
    };
    int t = 0;
    for (; (t + 2) * 64 <= p.Nk; ++t) body(t, std::true_type{});
    for (; t < nt; ++t) body(t, std::false_type{});

    if (qrow >= p.Nq) return;
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    if (p.out_f32) {
        float* Ob = (float*)p.out + (int64_t)b * p.o_bs + (int64_t)h * 64 + (int64_t)qrow * p.o_ts;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(f32x4*)(Ob + dt * 32 + 8 * g + 4 * lh) = (f32x4){oacc[dt][4 * g] * inv, oacc[dt][4 * g + 1] * inv, oacc[dt][4 * g + 2] * inv,
                                                                  oacc[dt][4 * g + 3] * inv};
    } else {
        bf16* Ob = (bf16*)p.out + (int64_t)b * p.o_bs + (int64_t)h * 64 + (int64_t)qrow * p.o_ts;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 hv, lv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float o = oacc[dt][4 * g + e] * inv;
                    hv[e] = (bf16)o;
                    lv[e] = (bf16)(o - (float)hv[e]);
                }
                *(bf16x4*)(Ob + dt * 32 + 8 * g + 4 * lh) = lv;
                *(bf16x4*)(Ob + p.o_pl + dt * 32 + 8 * g + 4 * lh) = hv;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The ping-pong form of the split d = 64 attention -- an EXPERIMENT kept behind a developer override (rsvld_attention_split_d64, out_f32
// bit 2), bit-identical to the 4-wave kernel and 10-13 % slower on MI355X; see the measurements at the end of this comment.
//
// In attn_split_d64_kernel a wave alternates  S chain (24 MFMAs) | softmax + hi|lo split of P (~230 vector instructions) | PV chain
// (24 MFMAs), and the two waves of a SIMD belong to two unsynchronised workgroups: whenever both are in their vector phase the matrix
// pipe idles (PMC: 0.55 busy).  Here ONE workgroup of 8 waves (two per SIMD: wave w and w + 4) runs the two groups of four in
// ANTI-PHASE by construction (the scheme of attn_d64c): per 64-key tile a wave has a vector segment V(t) = softmax(t) and a matrix
// segment M(t) = PV(t) + S(t+1) (48 MFMAs); ONE barrier per tile, between two barriers the early group runs M(k) then V(k+1), the late
// group V(k) then M(k) -- on every SIMD one wave feeds the matrix pipe while the other computes exponentials, in both halves of the
// period.  M(k) reads the V planes of tile k and the K planes of tile k + 1, so two tiles are live and the ring has three stages
// (3 x 32 KiB): tile k + 2 is requested at the start of period k and waited for at its end.  256 query rows per K / V tile: half the
// LDS-DMA traffic per row.  (First version: two barriers per tile, one segment per slot: 2 530 cycles per 48-MFMA segment -- barrier
// release + the ramp of the fragment ring in every slot -- and 20 % SLOWER than the 4-wave kernel.)
// Measured on 2 x 10 heads x 65 536 tokens (effective TFLOP/s; the 4-wave kernel: 391): two barriers per tile 319; + fragments two steps
// ahead 341; + branch-free M copies 354; + fragments three steps ahead 344; one barrier per tile 349; accumulators interleaved pairwise
// 342; the three possible pairings of waves into groups (w >> 2, w & 1, (w >> 1) & 1): 62.9 / 63.2 / 63.6 ms -- no difference.  Diagnostic
// builds: matrix segments alone 2 530 cycles per 48 MFMAs (1 536 nominal), vector segments alone 1 130, both 3 170 per slot: the two
// waves of a SIMD do not overlap their segments here whatever the grouping, and the unsynchronised 4-wave form does better by chance.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int ASP_SMEM = 12 * AS_TILE;      // 3 stages x (K_lo, K_hi, V_lo, V_hi)

__global__ __launch_bounds__(512) void attn_split_d64pp_kernel(AttnSplitArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifndef ASP_GRP
#define ASP_GRP 0   // which waves form the two anti-phase groups: 0: w >> 2, 1: w & 1, 2: (w >> 1) & 1 (the two waves of a SIMD must differ)
#endif
    const int grp = ASP_GRP == 0 ? (w >> 2) : ASP_GRP == 1 ? (w & 1) : ((w >> 1) & 1);
    const int l31 = lane & 31, lh = lane >> 5;
    const int q0 = (blockIdx.x * 8 + w) * 32, h = blockIdx.y, b = blockIdx.z;
    const bf16* Qb = p.q + (int64_t)b * p.q_bs + (int64_t)h * 64;
    const bf16* Kb = p.k + (int64_t)b * p.k_bs + (int64_t)h * 64;
    const bf16* Vb = p.v + (int64_t)b * p.v_bs + (int64_t)h * 64;
    const int nt = (p.Nk + 63) >> 6;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;

    // ---- tile DMA: wave w moves key rows 8 w .. 8 w + 7 of each of the four planes (one 1-KiB piece per plane)
    auto dma_one = [&](const char* base, uint32_t voff, uint32_t dst) {
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(dst), "s"(base) : "memory");
    };
    const int64_t k_rowb = p.k_ts * (int64_t)sizeof(bf16), v_rowb = p.v_ts * (int64_t)sizeof(bf16);
    const int64_t k_plb = p.k_pl * (int64_t)sizeof(bf16), v_plb = p.v_pl * (int64_t)sizeof(bf16);
    const int drow = lane >> 3, r16 = (8 * w + drow) & 15;
    const uint32_t kswz = (uint32_t)(((lane & 7) ^ ((r16 >> 1) & 7)) << 4), vswz = (uint32_t)(((lane & 7) ^ (((r16 >> 1) & 1) << 2)) << 4);
    const uint32_t kvo = (uint32_t)(drow * k_rowb) + kswz, vvo = (uint32_t)(drow * v_rowb) + vswz;
    const char* k_tile0 = (const char*)(Kb + (int64_t)(8 * w) * p.k_ts);
    const char* v_tile0 = (const char*)(Vb + (int64_t)(8 * w) * p.v_ts);
    auto dma_tile = [&](int t, int stage) {
        const uint32_t dst = lds0 + (uint32_t)(stage * 4 * AS_TILE + 8 * w * 128);
        if (t * 64 + 64 <= p.Nk) {
            const char* kb = k_tile0 + (int64_t)t * 64 * k_rowb;
            const char* vb = v_tile0 + (int64_t)t * 64 * v_rowb;
            dma_one(kb, kvo, dst);
            dma_one(kb + k_plb, kvo, dst + AS_TILE);
            dma_one(vb, vvo, dst + 2 * AS_TILE);
            dma_one(vb + v_plb, vvo, dst + 3 * AS_TILE);
        } else {   // ragged last tile: rows past Nk re-read the last key (their scores are masked); per-lane 64-bit addresses
            const int key = min(t * 64 + 8 * w + drow, p.Nk - 1);
            const char* ks = (const char*)(Kb + (int64_t)key * p.k_ts) + kswz;
            const char* vs = (const char*)(Vb + (int64_t)key * p.v_ts) + vswz;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const char* src = j == 0 ? ks : j == 1 ? ks + k_plb : j == 2 ? vs : vs + v_plb;
                uint32_t keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(dst + j * AS_TILE) : "memory");
            }
        }
    };
    dma_tile(0, 0);
    if (nt > 1) dma_tile(1, 1);
    if (nt > 2) dma_tile(2, 2);

    // ---- Q fragments (B operand), scaled in fp32 and re-split
    const int qrow = q0 + l31;
    bf16x8 qh[4], ql[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = 0.f;
        if (qrow < p.Nq) {
            const bf16* qp = Qb + (int64_t)qrow * p.q_ts + ks * 16 + lh * 8;
            float l[8], hh[8];
            unpack8<bf16>(*(const u32x4*)qp, l);
            unpack8<bf16>(*(const u32x4*)(qp + p.q_pl), hh);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (hh[e] + l[e]) * p.scale_log2e;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            qh[ks][e] = (bf16)f[e];
            ql[ks][e] = (bf16)(f[e] - (float)qh[ks][e]);
        }
    }
    f32x16 oacc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = l31 * 128 + (((2 * ks + lh) ^ ((l31 >> 1) & 7)) << 4);
    int voff[2];
    {
        const int qq = (lane >> 2) & 3, pp = lane & 3, g1 = (lane >> 4) & 1;
        const int row = 4 * lh + qq;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
            voff[dt] = row * 128 + (((4 * dt + 2 * g1 + (pp >> 1)) ^ (((row >> 1) & 1) << 2)) << 4) + ((pp & 1) << 3);
    }
    auto read_vt = [&](const char* base, int off) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off + 1024));
        typedef short s16x8 __attribute__((__vector_size__(8 * sizeof(short))));
        return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    f32x16 sacc[2];
    bf16x8 ph[4], pl[4];
    // S^T of tile 0 (prologue): two 32-key halves x 4 k-steps x 3 terms
    auto s_chain0 = [&]() __attribute__((always_inline)) {
        const char* Kl = smem;
        const char* Kh = Kl + AS_TILE;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kt][r] = 0.f;
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const int o = (n >> 2) * 4096 + koff[n & 3];
            const bf16x8 kl = *(const bf16x8*)(Kl + o), kh = *(const bf16x8*)(Kh + o);
            sacc[n >> 2] = mma_bf16(kl, qh[n & 3], sacc[n >> 2]);
            sacc[n >> 2] = mma_bf16(kh, ql[n & 3], sacc[n >> 2]);
            sacc[n >> 2] = mma_bf16(kh, qh[n & 3], sacc[n >> 2]);
        }
    };
    // ---- V(t): exact online softmax in fp32 with a deferred maximum, P split into hi | lo
    auto v_seg = [&](int t) __attribute__((always_inline)) {
        if ((t + 1) * 64 > p.Nk) {   // ragged last tile only (uniform branch)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kv = t * 64 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (kv >= p.Nk) sacc[kt][r] = -INFINITY;
                }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[kt][r]);
        {
            const uint32_t mb = __builtin_bit_cast(uint32_t, mx);
            const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
            mx = fmaxf(__builtin_bit_cast(float, (uint32_t)sw[0]), __builtin_bit_cast(float, (uint32_t)sw[1]));
        }
        float alpha = 1.0f;
        const bool need = mx > m_run + 8.0f;   // true on the first tile (m_run = -inf)
        if (need) {
            alpha = __builtin_amdgcn_exp2f(m_run - mx);
            m_run = mx;
        }
        float r4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(sacc[kt][r] - m_run);
                r4[r >> 2] += pv;
                const bf16 hv = (bf16)pv;
                ph[2 * kt + (r >> 3)][r & 7] = hv;
                pl[2 * kt + (r >> 3)][r & 7] = (bf16)(pv - (float)hv);
            }
        l_run = l_run * alpha + ((r4[0] + r4[1]) + (r4[2] + r4[3]));
        if (__builtin_expect(__any(need), 0)) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
        }
    };
    // ---- M(t): O^T += V^T P^T from the V planes of tile t (steps 0..7: n = 2 s4 + dt), then S^T of tile t + 1 from its K planes (steps
    // 8..15: 8 + 4 kt + ks) as ONE sequence of 16 steps x 3 MFMAs, the fragment pair of step n + 3 requested before the MFMAs of step n --
    // across the PV / S border too (fa = lo plane, fb = hi plane of the step's operand).  Two branch-free copies (MORE: tile t + 1 exists):
    // a wave in its matrix segment has the pipe to feed, and every uniform branch is an instruction-fetch bubble of ~40 cycles.
    // ``dma_u`` >= 0: this wave's four LDS-DMA pieces of tile dma_u are issued behind the first MFMAs.
    auto m_seg = [&](int st_t, int dma_u, auto more_c) __attribute__((always_inline)) {
        constexpr bool MORE = decltype(more_c)::value;
        constexpr int NSTEP = MORE ? 16 : 8;
        // Steps are processed in PAIRS whose accumulators differ (PV: d-halves dt = 0, 1 of one key step; S: key halves kt = 0, 1 of one
        // k-step) with their three terms interleaved, so that no MFMA reads the accumulator the previous one writes: a dependent MFMA
        // waits at the head of the SIMD's issue port and blocks the vector wave behind it (what attn_d64c's step-major chains avoid).
        constexpr int NPAIR = NSTEP / 2;
        const char* Vl = smem + (st_t * 4 + 2) * AS_TILE;
        const char* Vh = Vl + AS_TILE;
        const char* Kl = smem + (st_t == 2 ? 0 : st_t + 1) * 4 * AS_TILE;
        const char* Kh = Kl + AS_TILE;
        bf16x8 fa[2][2], fb[2][2];     // [ring slot][member of the pair]
        auto req = [&](int pr, int slot) __attribute__((always_inline)) {
            if (pr < 4) {              // PV pair pr: key step s4 = pr, dt = 0, 1
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    fa[slot][j] = read_vt(Vl, voff[j] + pr * 2048);
                    fb[slot][j] = read_vt(Vh, voff[j] + pr * 2048);
                }
            } else {                   // S pair: k-step ks = pr - 4, key halves kt = 0, 1
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    fa[slot][j] = *(const bf16x8*)(Kl + j * 4096 + koff[pr - 4]);
                    fb[slot][j] = *(const bf16x8*)(Kh + j * 4096 + koff[pr - 4]);
                }
            }
        };
        req(0, 0);
        if (MORE) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[kt][r] = 0.f;
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int pr = 0; pr < NPAIR; ++pr) {
            if (pr + 1 < NPAIR) req(pr + 1, (pr + 1) & 1);
            const int sl = pr & 1;
            if (pr < 4) {
                oacc[0] = mma_bf16(fb[sl][0], pl[pr], oacc[0]);
                oacc[1] = mma_bf16(fb[sl][1], pl[pr], oacc[1]);
                oacc[0] = mma_bf16(fa[sl][0], ph[pr], oacc[0]);
                oacc[1] = mma_bf16(fa[sl][1], ph[pr], oacc[1]);
                oacc[0] = mma_bf16(fb[sl][0], ph[pr], oacc[0]);
                oacc[1] = mma_bf16(fb[sl][1], ph[pr], oacc[1]);
            } else {
                const int ks = pr - 4;
                sacc[0] = mma_bf16(fa[sl][0], qh[ks], sacc[0]);
                sacc[1] = mma_bf16(fa[sl][1], qh[ks], sacc[1]);
                sacc[0] = mma_bf16(fb[sl][0], ql[ks], sacc[0]);
                sacc[1] = mma_bf16(fb[sl][1], ql[ks], sacc[1]);
                sacc[0] = mma_bf16(fb[sl][0], qh[ks], sacc[0]);
                sacc[1] = mma_bf16(fb[sl][1], qh[ks], sacc[1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (pr == 0) {
                if (dma_u >= 0) dma_tile(dma_u, dma_u % 3);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // tiles 0 .. 2 landed
    __builtin_amdgcn_sched_barrier(0);
    s_chain0();

    // Time is cut into slots: group g runs V(t) in slot 2t + g and M(t) in slot 2t + g + 1, so in every slot one wave of a SIMD is in
    // a matrix segment and the other in a vector segment.  ONE barrier per tile, at the end of the EVEN slots: between two barriers
    // (slots 2k + 1, 2k + 2) the early group runs M(k) then V(k+1), the late group V(k) then M(k); nothing inside such a period depends
    // on the other group.  Tile k + 2 is requested in slot 2k + 1 (k >= 1; its stage held tile k - 1, last read in slot 2k, before the
    // barrier) and waited for at the period's end.  (One segment type per loop trip keeps ONE copy of V and two of M in the code:
    // an if / else with both orders written out needed 256 registers + 112 spilled dwords.)
    const int nslots = 2 * nt + 1;
    int st_t = 0;                    // stage of this group's current tile (t % 3), advanced behind its M segment
    for (int s = 0; s < nslots; ++s) {
        const int u_dma = (s + 3) >> 1;
        const bool do_dma = (s & 1) && s >= 3 && u_dma < nt;
        bool dma_done = false;
        const int rel = s - grp;
        if (rel >= 0 && rel < 2 * nt) {
            const int t = rel >> 1;
            if ((rel & 1) == 0) {
                if (do_dma) { dma_tile(u_dma, u_dma % 3); dma_done = true; }   // (the vector wave: at once)
                v_seg(t);
            } else {
                if (t + 1 < nt) m_seg(st_t, do_dma ? u_dma : -1, std::true_type{});
                else m_seg(st_t, do_dma ? u_dma : -1, std::false_type{});
                dma_done = true;
                st_t = st_t == 2 ? 0 : st_t + 1;
            }
        }
        if (do_dma && !dma_done) dma_tile(u_dma, u_dma % 3);   // (a group idle in this slot still moves its rows)
        if (!(s & 1)) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    if (qrow >= p.Nq) return;
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    if (p.out_f32) {
        float* Ob = (float*)p.out + (int64_t)b * p.o_bs + (int64_t)h * 64 + (int64_t)qrow * p.o_ts;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(f32x4*)(Ob + dt * 32 + 8 * g + 4 * lh) = (f32x4){oacc[dt][4 * g] * inv, oacc[dt][4 * g + 1] * inv, oacc[dt][4 * g + 2] * inv,
                                                                  oacc[dt][4 * g + 3] * inv};
    } else {
        bf16* Ob = (bf16*)p.out + (int64_t)b * p.o_bs + (int64_t)h * 64 + (int64_t)qrow * p.o_ts;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 hv, lv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float o = oacc[dt][4 * g + e] * inv;
                    hv[e] = (bf16)o;
                    lv[e] = (bf16)(o - (float)hv[e]);
                }
                *(bf16x4*)(Ob + dt * 32 + 8 * g + 4 * lh) = lv;
                *(bf16x4*)(Ob + p.o_pl + dt * 32 + 8 * g + 4 * lh) = hv;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Flash attention on planes, single head, d = 512, keys and values THE SAME tensor X (SR3's SelfAttention after the pack-time
// re-association of sr3_modules/unet.py:114-143: keys and values are the normalised input itself, csrc/attention.hip "SH").
//
// The 16-bit kernel gives one wave 32 query rows over the whole head dimension: Q in 128 registers, O^T in 256.  With split
// operands Q is two planes (256 registers) and the register file cannot hold that next to O^T.  Here TWO waves share 32 query rows,
// each owning HALF of the head dimension -- its 256 channels of Q_hi | Q_lo (128 registers) and of O^T (128 registers):
//   S^T  : a wave contracts its own 256 channels (16 k-steps x 3 terms = 48 MFMAs) -> a PARTIAL score tile; the two partials
//          are exchanged through LDS (4 KiB per wave) and added -- both waves then hold the same scores bit for bit (a + b = b + a)
//          and take identical softmax decisions;
//   PV   : P (hi | lo split in registers after the fp32 softmax) times the wave's own 256 channels of X: 8 x 2 x 3 = 48 MFMAs.
// No MFMA is issued twice; the softmax (16 scores per lane) is -- 96 MFMAs per wave and 32-key tile against ~150 vector instructions.
// X tiles (32 keys x 512, planes lo | hi = 64 KiB) are double-buffered by LDS-DMA into the dual-use image of the 16-bit kernel
// (chunk c of row r at c ^ f(r): conflict-free for the row-wise K reads AND the transposed V reads), one workgroup (4 waves, 64
// query rows) per CU.  Two barriers per tile (score exchange; ring), the first one only behind the 48 MFMAs of the S chain.
// ---------------------------------------------------------------------------------------------------------------------------
struct AttnSplit512Args {
    const bf16* q; const bf16* x; void* out;
    int Nq, Nk;
    int64_t q_bs, q_ts, q_pl, x_bs, x_ts, x_pl, o_bs, o_ts, o_pl;
    float scale_log2e;
    int out_f32;
};
constexpr int A5S_IMG = 32 * 1024;            // one plane of a 32-key tile
constexpr int A5S_STAGE = 2 * A5S_IMG;        // lo | hi
constexpr int A5S_XCH = 2 * A5S_STAGE;        // score exchange: 4 waves x 4 KiB
constexpr int A5S_SMEM = A5S_XCH + 4 * 4096;
__host__ __device__ constexpr int a5s_f(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }

__global__ __launch_bounds__(256) void attn_split_d512_kernel(AttnSplit512Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = w >> 1, half = w & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const int q0 = blockIdx.x * 64 + pair * 32, b = blockIdx.y;
    const bf16* Qb = p.q + (int64_t)b * p.q_bs;
    const bf16* Xb = p.x + (int64_t)b * p.x_bs;
    const int nt = (p.Nk + 31) >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;

    // ---- tile DMA (inline asm: the compiler must not know of it, or it drains vmcnt in front of every LDS read that follows; M0 is
    // saved and restored inside the statement).  Wave w moves key rows 8w .. 8w+7 of both planes, one 1-KiB row per piece; lane ->
    // LDS chunk position `lane`, source chunk lane ^ f(row & 15).
    auto dma_one = [&](const char* base, uint32_t voff, uint32_t dst) {
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(dst), "s"(base) : "memory");
    };
    const int64_t rowb = p.x_ts * (int64_t)sizeof(bf16), plb = p.x_pl * (int64_t)sizeof(bf16);
    uint32_t xvo[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) xvo[i] = (uint32_t)(i * rowb) + (uint32_t)((lane ^ a5s_f((w * 8 + i) & 15)) << 4);
    const char* x_tile0 = (const char*)(Xb + (int64_t)(w * 8) * p.x_ts);
    auto dma_piece = [&](int t, int j) {   // j = 0..15: row 8w + (j >> 1) of plane j & 1 of tile t -> stage t & 1
        const int i = j >> 1, pl = j & 1, r = w * 8 + i;
        const uint32_t dst = lds0 + (t & 1) * A5S_STAGE + pl * A5S_IMG + r * 1024;
        if (t * 32 + 32 <= p.Nk) {
            dma_one(x_tile0 + (int64_t)t * 32 * rowb + pl * plb, xvo[i], dst);
        } else {   // ragged last tile: rows past Nk re-read the last key; their scores are masked
            const int key = min(t * 32 + r, p.Nk - 1);
            dma_one((const char*)(Xb + (int64_t)key * p.x_ts) + pl * plb, (uint32_t)((lane ^ a5s_f(r & 15)) << 4), dst);
        }
    };
#pragma unroll
    for (int j = 0; j < 16; ++j) dma_piece(0, j);

    // ---- Q fragments of this wave's 256 channels (B operand: column = the lane's query row), scaled in fp32 and re-split
    const int qrow = q0 + l31;
    bf16x8 qh[16], ql[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = 0.f;
        if (qrow < p.Nq) {
            const bf16* qp = Qb + (int64_t)qrow * p.q_ts + (16 * half + ks) * 16 + lh * 8;
            float l[8], hh[8];
            unpack8<bf16>(*(const u32x4*)qp, l);
            unpack8<bf16>(*(const u32x4*)(qp + p.q_pl), hh);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (hh[e] + l[e]) * p.scale_log2e;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            qh[ks][e] = (bf16)f[e];
            ql[ks][e] = (bf16)(f[e] - (float)qh[ks][e]);
        }
    }
    f32x16 oacc[8];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // per-lane read offsets.  K: row l31, chunk 2 ksg + lh (ksg = 16 half + ks) = 16 a + (2 c + lh): off = kbase[c] + 256 a
    int kbase[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) kbase[c] = l31 * 1024 + (((2 * c + lh) ^ a5s_f(l31 & 15)) << 4);
    // V (transposed read): lane 16g + 4q + p supplies row (4 lh + q) + {16 s + 8 hf}, d = 32 dtg + 16 (g&1) + 4p .. +3 (dtg = 8 half + dt);
    // with dtg = 4 e + f: off = vbase[f] + 256 e + 1024 (16 s), the hf = 1 rows at (off ^ 32) + 8 * 1024
    int vbase[4];
    {
        const int qq = (lane >> 2) & 3, pp = lane & 3, g1 = (lane >> 4) & 1;
#pragma unroll
        for (int f = 0; f < 4; ++f)
            vbase[f] = (4 * lh + qq) * 1024 + ((f ^ qq) << 6) + (((2 * g1 + (pp >> 1)) ^ lh) << 4) + ((pp & 1) << 3);
    }
    auto read_vt = [&](const char* img, int off) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + off));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + (off ^ 32) + 8 * 1024));
        typedef short s16x8 __attribute__((__vector_size__(8 * sizeof(short))));
        return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    char* xch_own = smem + A5S_XCH + w * 4096 + lane * 16;
    const char* xch_peer = smem + A5S_XCH + (w ^ 1) * 4096 + lane * 16;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // tile 0 landed
    __builtin_amdgcn_sched_barrier(0);

    // The tile loop exists twice: a steady-state copy for tiles whose successor exists and is FULL (no test around the 16 LDS-DMA
    // pieces, no ragged mask, the next tile's row base one scalar add away) and the general copy for the last tiles -- one wave per
    // SIMD: nobody covers the instruction-fetch bubble of a branch ("Branch-free steady-state loops").
    const char* x_nxt = x_tile0;
    auto body = [&](int t, auto steady_c) __attribute__((always_inline)) {
        constexpr bool ST = decltype(steady_c)::value;
        x_nxt += 32 * rowb;     // row 8w of tile t + 1
        const char* Il = smem + (t & 1) * A5S_STAGE;
        const char* Ih = Il + A5S_IMG;
        const bool more = ST ? true : t + 1 < nt;

        // ---- partial S^T over this wave's 256 channels; the 16 LDS-DMA pieces of tile t + 1 are issued between the k-steps (the
        // other stage was last read in iteration t - 1, behind its closing barrier).  Fragments of k-step ks + 1 are requested
        // before the MFMAs of k-step ks.
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
        // (fragments TWO k-steps ahead: three MFMAs = 96 cycles do not cover an LDS round trip on one wave per SIMD)
        bf16x8 kl[3], kh[3];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int o = kbase[ks & 7] + 256 * (2 * half + (ks >> 3));
            kl[ks] = *(const bf16x8*)(Il + o);
            kh[ks] = *(const bf16x8*)(Ih + o);
        }
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            if (ks + 2 < 16) {
                const int o = kbase[(ks + 2) & 7] + 256 * (2 * half + ((ks + 2) >> 3));
                kl[(ks + 2) % 3] = *(const bf16x8*)(Il + o);
                kh[(ks + 2) % 3] = *(const bf16x8*)(Ih + o);
            }
            sacc = mma_bf16(kl[ks % 3], qh[ks], sacc);
            sacc = mma_bf16(kh[ks % 3], ql[ks], sacc);
            sacc = mma_bf16(kh[ks % 3], qh[ks], sacc);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (ST) dma_one(x_nxt + (ks & 1) * plb, xvo[ks >> 1], lds0 + ((t + 1) & 1) * A5S_STAGE + (ks & 1) * A5S_IMG + (w * 8 + (ks >> 1)) * 1024);
            else if (more) dma_piece(t + 1, ks);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- exchange the partials with the other half's wave
#pragma unroll
        for (int g = 0; g < 4; ++g) *(f32x4*)(xch_own + g * 1024) = (f32x4){sacc[4 * g], sacc[4 * g + 1], sacc[4 * g + 2], sacc[4 * g + 3]};
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 o = *(const f32x4*)(xch_peer + g * 1024);
            sacc[4 * g] += o[0]; sacc[4 * g + 1] += o[1]; sacc[4 * g + 2] += o[2]; sacc[4 * g + 3] += o[3];
        }
        if (!ST && (t + 1) * 32 > p.Nk) {   // ragged last tile (uniform branch)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kv = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (kv >= p.Nk) sacc[r] = -INFINITY;
            }
        }
        // ---- online softmax in fp32, deferred maximum (the reference point moves only when a row outgrows it by 2^8: P <= 2^8 is
        // as good a bf16 hi | lo pair as P <= 1, and the rescale of O is skipped on almost every tile)
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[r]);
        {
            const uint32_t mb = __builtin_bit_cast(uint32_t, mx);
            const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
            mx = fmaxf(__builtin_bit_cast(float, (uint32_t)sw[0]), __builtin_bit_cast(float, (uint32_t)sw[1]));
        }
        float alpha = 1.0f;
        const bool need = mx > m_run + 8.0f;   // true on the first tile (m_run = -inf)
        if (need) {
            alpha = __builtin_amdgcn_exp2f(m_run - mx);
            m_run = mx;
        }
        float rs = 0.f;
        bf16x8 ph[2], pl[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pv = __builtin_amdgcn_exp2f(sacc[r] - m_run);
            rs += pv;
            const bf16 hv = (bf16)pv;
            ph[r >> 3][r & 7] = hv;
            pl[r >> 3][r & 7] = (bf16)(pv - (float)hv);
        }
        l_run = l_run * alpha + rs;
        if (__builtin_expect(__any(need), 0)) {
            // O lives in the accumulator file.  Written as plain C++ (oacc *= alpha) hipcc pulls all 128 values into VGPRs at once, and
            // the pressure of that one cold block makes it park Q in AccVGPRs for the WHOLE loop (every Q fragment copied back in front of
            // its MFMA: 226 v_accvgpr_read per tile).  One element at a time through a scratch VGPR instead (attention_d512_body.inc);
            // the MFMA -> read hazard is covered by the 48 S-chain MFMAs since the last PV, the write -> MFMA hazard by the s_nop.
            asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#pragma unroll
            for (int dt = 0; dt < 8; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float x = oacc[dt][r], tmp;
                    asm volatile("v_accvgpr_read_b32 %1, %0\n\tv_mul_f32 %1, %1, %2\n\ts_nop 0\n\tv_accvgpr_write_b32 %0, %1\n\ts_nop 1"
                                 : "+a"(x), "=&v"(tmp)
                                 : "v"(alpha));
                    oacc[dt][r] = x;
                }
        }
        // ---- O^T[d][q] += X^T P^T over this wave's 8 d-blocks: 2 key steps x 3 terms each, fragments one pair ahead
        bf16x8 vl[3], vh[3];
        auto vreq = [&](int n, int slot) {   // n = 2 dt + s
            const int dtg = 8 * half + (n >> 1), s2 = n & 1;
            const int off = vbase[dtg & 3] + (dtg >> 2) * 256 + (16 * s2) * 1024;
            vl[slot] = read_vt(Il, off);
            vh[slot] = read_vt(Ih, off);
        };
        vreq(0, 0);
        vreq(1, 1);
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            if (n + 2 < 16) vreq(n + 2, (n + 2) % 3);
            oacc[n >> 1] = mma_bf16(vh[n % 3], pl[n & 1], oacc[n >> 1]);
            oacc[n >> 1] = mma_bf16(vl[n % 3], ph[n & 1], oacc[n >> 1]);
            oacc[n >> 1] = mma_bf16(vh[n % 3], ph[n & 1], oacc[n >> 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // tile t + 1 landed; everyone done with this stage and the exchange
        __builtin_amdgcn_sched_barrier(0);
    };
    int t = 0;
    for (; (t + 2) * 32 <= p.Nk; ++t) body(t, std::true_type{});
    for (; t < nt; ++t) body(t, std::false_type{});

    if (qrow >= p.Nq) return;
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    const int d0 = 256 * half;
    if (p.out_f32) {
        float* Ob = (float*)p.out + (int64_t)b * p.o_bs + (int64_t)qrow * p.o_ts + d0;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(f32x4*)(Ob + dt * 32 + 8 * g + 4 * lh) = (f32x4){oacc[dt][4 * g] * inv, oacc[dt][4 * g + 1] * inv, oacc[dt][4 * g + 2] * inv,
                                                                  oacc[dt][4 * g + 3] * inv};
    } else {
        bf16* Ob = (bf16*)p.out + (int64_t)b * p.o_bs + (int64_t)qrow * p.o_ts + d0;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 hv, lv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float o = oacc[dt][4 * g + e] * inv;
                    hv[e] = (bf16)o;
                    lv[e] = (bf16)(o - (float)hv[e]);
                }
                *(bf16x4*)(Ob + dt * 32 + 8 * g + 4 * lh) = lv;
                *(bf16x4*)(Ob + p.o_pl + dt * 32 + 8 * g + 4 * lh) = hv;
            }
    }
}

// w [R][Ctot] fp32 (R = Cout * taps) -> fp16 pairs [R][lo | hi]: hi = fp16(w), lo = fp16(w - hi) (RSVLD_F16W2; a weight beyond the fp16
// range saturates hi and leaves the rest to lo: not a case a trained network has)
__global__ __launch_bounds__(256) void pack_weight_pairs_kernel(const float* __restrict__ w, f16* __restrict__ o, int64_t items, int C8) {
    const int C = C8 * 8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / C8;
        const int c = (int)(i - row * C8) * 8;
        float f[8], l[8];
        ld8f(w + row * C + c, f);
        f16x8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            h[e] = (f16)sat_f16_keep_nan(f[e]);
            l[e] = f[e] - (float)h[e];
        }
        f16* d = o + row * (2 * (int64_t)C) + c;
        *(u32x4*)d = pack8<f16>(l);
        *(u32x4*)(d + C) = __builtin_bit_cast(u32x4, h);
    }
}

// fp32 rows -> RSVLD_F16Q8 rows (activations: lo part first) / weight rows (hi part first); 8 channels per thread
template <bool ACT>
__global__ __launch_bounds__(256) void to_hq8_kernel(const float* __restrict__ x, f16* __restrict__ o, int64_t items, int C8) {
    const int C = C8 * 8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / C8;
        const int c = (int)(i - row * C8) * 8;
        float f[8];
        ld8f(x + row * C + c, f);
        if (ACT) st_hq8<true, RSVLD_HQ8_SX_LO, RSVLD_HQ8_SX_HI>(o + row * (2 * (int64_t)C), C, c, f);
        else st_hq8<false, RSVLD_HQ8_SW_LO, RSVLD_HQ8_SW_HI>(o + row * (2 * (int64_t)C), C, c, f);
    }
}

static unsigned ew_blocks(int64_t items) {
    int64_t b = cdiv64(items, 256);
    return (unsigned)(b < 1 ? 1 : (b > 65536 ? 65536 : b));
}

}  // namespace

extern "C" int rsvld_split_planes(const float* x, void* planes, int64_t rows, int C, void* stream) {
    if (!x || !planes || rows < 1 || C < 8 || C % 8) return RSVLD_EINVAL;
    const int64_t items = rows * (C / 8);
    hipLaunchKernelGGL(split_planes_kernel, dim3(ew_blocks(items)), dim3(256), 0, (hipStream_t)stream, x, (bf16*)planes, items, C / 8);
    return rsvld_check_launch();
}

extern "C" int rsvld_merge_planes(const void* planes, float* x, int64_t rows, int C, void* stream) {
    if (!x || !planes || rows < 1 || C < 8 || C % 8) return RSVLD_EINVAL;
    const int64_t items = rows * (C / 8);
    hipLaunchKernelGGL(merge_planes_kernel, dim3(ew_blocks(items)), dim3(256), 0, (hipStream_t)stream, (const bf16*)planes, x, items, C / 8);
    return rsvld_check_launch();
}

extern "C" int rsvld_planes_to_f16(const void* planes, int64_t ld, int64_t plane_stride, void* out, int64_t out_ld, int64_t rows, int C, void* stream) {
    if (!planes || !out || rows < 1 || C < 8 || C % 8 || ld % 8 || plane_stride % 8 || out_ld % 8 || out_ld < C) return RSVLD_EINVAL;
    const int64_t items = rows * (C / 8);
    hipLaunchKernelGGL(planes_to_f16_kernel, dim3(ew_blocks(items)), dim3(256), 0, (hipStream_t)stream, (const bf16*)planes, (f16*)out, items, C / 8,
                       ld, plane_stride, out_ld);
    return rsvld_check_launch();
}

extern "C" int rsvld_f16_to_planes(const void* x, int64_t ld, void* planes, int64_t rows, int C, void* stream) {
    if (!x || !planes || rows < 1 || C < 8 || C % 8 || ld % 8 || ld < C) return RSVLD_EINVAL;
    const int64_t items = rows * (C / 8);
    hipLaunchKernelGGL(f16_to_planes_kernel, dim3(ew_blocks(items)), dim3(256), 0, (hipStream_t)stream, (const f16*)x, (bf16*)planes, items, C / 8, ld);
    return rsvld_check_launch();
}

extern "C" int rsvld_split_pack_weights(const float* w, void* w3, int64_t Cout, int taps, int Ctot, void* stream) {
    if (!w || !w3 || Cout < 1 || taps < 1 || Ctot < 8 || Ctot % 8) return RSVLD_EINVAL;
    const int64_t items = Cout * taps * (Ctot / 8);
    hipLaunchKernelGGL(split_pack_weights_kernel, dim3(ew_blocks(items)), dim3(256), 0, (hipStream_t)stream, w, (bf16*)w3, items, Ctot / 8);
    return rsvld_check_launch();
}

extern "C" int rsvld_pack_weight_pairs(const float* w, void* w2, int64_t Cout, int taps, int Ctot, void* stream) {
    if (!w || !w2 || Cout < 1 || taps < 1 || Ctot < 8 || Ctot % 8) return RSVLD_EINVAL;
    const int64_t items = Cout * taps * (Ctot / 8);
    hipLaunchKernelGGL(pack_weight_pairs_kernel, dim3(ew_blocks(items)), dim3(256), 0, (hipStream_t)stream, w, (f16*)w2, items, Ctot / 8);
    return rsvld_check_launch();
}

extern "C" int rsvld_pack_weight_hq8(const float* w, void* whq, int64_t Cout, int taps, int Ctot, void* stream) {
    if (!w || !whq || Cout < 1 || taps < 1 || Ctot < 32 || Ctot % 32) return RSVLD_EINVAL;
    const int64_t items = Cout * taps * (Ctot / 8);
    hipLaunchKernelGGL(to_hq8_kernel<false>, dim3(ew_blocks(items)), dim3(256), 0, (hipStream_t)stream, w, (f16*)whq, items, Ctot / 8);
    return rsvld_check_launch();
}

extern "C" int rsvld_split_hq8(const float* x, void* out, int64_t rows, int C, void* stream) {
    if (!x || !out || rows < 1 || C < 32 || C % 32) return RSVLD_EINVAL;
    const int64_t items = rows * (C / 8);
    hipLaunchKernelGGL(to_hq8_kernel<true>, dim3(ew_blocks(items)), dim3(256), 0, (hipStream_t)stream, x, (f16*)out, items, C / 8);
    return rsvld_check_launch();
}

extern "C" int rsvld_planes_to_triple(const void* planes, void* w3, int64_t rows, int64_t rows_p, int C, int64_t ld, void* stream) {
    if (!planes || !w3 || rows < 1 || rows_p < rows || C < 8 || C % 8 || ld < 2 * C || ld % 8) return RSVLD_EINVAL;
    const int64_t items = rows_p * (C / 8);
    hipLaunchKernelGGL(planes_to_triple_kernel, dim3(ew_blocks(items)), dim3(256), 0, (hipStream_t)stream, (const bf16*)planes, (bf16*)w3, rows,
                       items, C / 8, ld);
    return rsvld_check_launch();
}

extern "C" int rsvld_planes_transpose_triple(const void* planes, void* w3, int64_t rows, int64_t rows_p, int C, int64_t ld, void* stream) {
    if (!planes || !w3 || rows < 1 || rows_p < rows || rows_p % 8 || C < 8 || C % 8 || ld < 2 * C || ld % 8) return RSVLD_EINVAL;
    const int64_t gx = cdiv64(rows_p, 64);
    if (gx > 0x7fffffffLL) return RSVLD_EUNSUPPORTED;
    hipLaunchKernelGGL(planes_transpose_triple_kernel, dim3((unsigned)gx, (unsigned)((C + 63) / 64)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)planes, (uint16_t*)w3, rows, rows_p, C, ld);
    return rsvld_check_launch();
}

extern "C" int rsvld_softmax_rows_split(const float* s, void* p_planes, int64_t rows, int cols, int cols_p, int64_t ld, float scale,
                                        void* stream) {
    if (!s || !p_planes || rows < 1 || rows > 0x7fffffffLL || cols < 1 || cols_p < cols || cols_p % 8 || ld < cols_p || ld % 4) return RSVLD_EINVAL;
    hipLaunchKernelGGL(softmax_rows_split_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, s, (bf16*)p_planes, cols, cols_p, ld,
                       scale * 1.44269504088896340736f);
    return rsvld_check_launch();
}

extern "C" int rsvld_attention_split_d64(const void* q, const void* k, const void* v, void* out, int B, int heads, int Nq, int Nk,
                                         int64_t q_batch_stride, int64_t q_tok_stride, int64_t q_plane, int64_t k_batch_stride,
                                         int64_t k_tok_stride, int64_t k_plane, int64_t v_batch_stride, int64_t v_tok_stride,
                                         int64_t v_plane, int64_t o_batch_stride, int64_t o_tok_stride, int64_t o_plane, float scale,
                                         int out_f32, void* stream) {
    if (!q || !k || !v || !out || B < 1 || heads < 1 || Nq < 1 || Nk < 1) return RSVLD_EINVAL;
    if (heads > 65535 || B > 65535) return RSVLD_EUNSUPPORTED;
    if ((q_batch_stride | q_tok_stride | q_plane | k_batch_stride | k_tok_stride | k_plane | v_batch_stride | v_tok_stride | v_plane) % 8 != 0)
        return RSVLD_EINVAL;   // 16-byte vector accesses
    if ((o_batch_stride | o_tok_stride | o_plane) % 4 != 0) return RSVLD_EINVAL;
    // the kernel forms 32-bit lane offsets row * row_bytes with row <= 15 inside a 16-row piece group
    if (16 * k_tok_stride * 2 >= ((int64_t)1 << 32) || 16 * v_tok_stride * 2 >= ((int64_t)1 << 32)) return RSVLD_EUNSUPPORTED;
    AttnSplitArgs a;
    a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.out = out; a.Nq = Nq; a.Nk = Nk;
    a.q_bs = q_batch_stride; a.q_ts = q_tok_stride; a.q_pl = q_plane; a.k_bs = k_batch_stride; a.k_ts = k_tok_stride; a.k_pl = k_plane;
    a.v_bs = v_batch_stride; a.v_ts = v_tok_stride; a.v_pl = v_plane; a.o_bs = o_batch_stride; a.o_ts = o_tok_stride; a.o_pl = o_plane;
    a.scale_log2e = scale * 1.44269504088896340736f;
    a.out_f32 = (out_f32 & 1) ? 1 : 0;
    static const hipError_t attr = hipFuncSetAttribute((const void*)attn_split_d64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, AS_SMEM);
    static const hipError_t attr_pp = hipFuncSetAttribute((const void*)attn_split_d64pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ASP_SMEM);
    if (attr != hipSuccess || attr_pp != hipSuccess) return RSVLD_ELAUNCH;
    // out_f32 bit 2 (developer override, tests / A-B runs): the ping-pong form attn_split_d64pp_kernel.  NOT chosen by the library: on
    // MI355X it runs 342-354 effective TFLOP/s where this 4-wave kernel runs 391 (2 x 10 heads x 65 536 tokens).
    if ((out_f32 & 4) && Nk > 64 && 64 * k_tok_stride * 2 < ((int64_t)1 << 32) && 64 * v_tok_stride * 2 < ((int64_t)1 << 32)) {
        hipLaunchKernelGGL(attn_split_d64pp_kernel, dim3((unsigned)((Nq + 255) / 256), (unsigned)heads, (unsigned)B), dim3(512), ASP_SMEM,
                           (hipStream_t)stream, a);
        return rsvld_check_launch();
    }
    dim3 grid((unsigned)((Nq + 127) / 128), (unsigned)heads, (unsigned)B);
    hipLaunchKernelGGL(attn_split_d64_kernel, grid, dim3(256), AS_SMEM, (hipStream_t)stream, a);
    return rsvld_check_launch();
}

extern "C" int rsvld_attention_split_d512_shared(const void* q, const void* x, void* out, int B, int Nq, int Nk,
                                                 int64_t q_batch_stride, int64_t q_tok_stride, int64_t q_plane, int64_t x_batch_stride,
                                                 int64_t x_tok_stride, int64_t x_plane, int64_t o_batch_stride, int64_t o_tok_stride,
                                                 int64_t o_plane, float scale, int out_f32, void* stream) {
    if (!q || !x || !out || B < 1 || Nq < 1 || Nk < 1) return RSVLD_EINVAL;
    if (B > 65535) return RSVLD_EUNSUPPORTED;
    if ((q_batch_stride | q_tok_stride | q_plane | x_batch_stride | x_tok_stride | x_plane) % 8 != 0) return RSVLD_EINVAL;
    if ((o_batch_stride | o_tok_stride | o_plane) % 4 != 0) return RSVLD_EINVAL;
    if (x_tok_stride * 2 * 8 >= ((int64_t)1 << 32)) return RSVLD_EUNSUPPORTED;   // 32-bit lane offsets inside a piece group
    AttnSplit512Args a;
    a.q = (const bf16*)q; a.x = (const bf16*)x; a.out = out; a.Nq = Nq; a.Nk = Nk;
    a.q_bs = q_batch_stride; a.q_ts = q_tok_stride; a.q_pl = q_plane; a.x_bs = x_batch_stride; a.x_ts = x_tok_stride; a.x_pl = x_plane;
    a.o_bs = o_batch_stride; a.o_ts = o_tok_stride; a.o_pl = o_plane;
    a.scale_log2e = scale * 1.44269504088896340736f;
    a.out_f32 = out_f32 ? 1 : 0;
    static const hipError_t attr = hipFuncSetAttribute((const void*)attn_split_d512_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, A5S_SMEM);
    if (attr != hipSuccess) return RSVLD_ELAUNCH;
    hipLaunchKernelGGL(attn_split_d512_kernel, dim3((unsigned)((Nq + 63) / 64), (unsigned)B), dim3(256), A5S_SMEM, (hipStream_t)stream, a);
    return rsvld_check_launch();
}
