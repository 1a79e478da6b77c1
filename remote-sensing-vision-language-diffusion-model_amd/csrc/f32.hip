// f32.hip — the fp32-operand kernel family behind ``ae_dtype: fp32`` / ``diffusion_dtype: fp32`` (reference:
// models/SR_model.py:28-33 — no autocast; also what the reference's CPU path computes).  Everything the VAE and the Stage-2
// networks need, on fp32 NHWC tensors with fp32 weights:
//   conv_f32_kernel        implicit-GEMM convolution / 1x1 on v_mfma_f32_32x32x2_f32 (fp32 operands, fp32 accumulate)
//   gn_f32_*               GroupNorm statistics (fp64 partial sums, fixed merge order) and apply (+SiLU)
//   attn_f32_kernel        flash-style attention (scores never materialised), any head dim that is a multiple of 32 up to 512
//   nchw_to_nhwc_f32       layout conversion
//   layernorm / concat / axpby / absdiff   the remaining ops of the Stage-2 UNet + ControlNet under ``diffusion_dtype: fp32``
// This is the ACCURACY mode of the library, not the fast path: simple LDS tiling, no LDS-DMA pipelines.  The fp32 matrix
// rate of gfx950 is 1/16 of the 16-bit rate, so the reference's default (bf16 VAE) stays the benchmarked configuration.
//
// SPLIT instantiations (round 3; ``RSVLD_TUNE_F32_SPLIT`` / rsvld_attention_f32_split): the same kernels on the same fp32 tensors, but
// every matrix operand x is split on the fly into two bf16 numbers, hi = bf16(x) and lo = bf16(x - hi) (together 16 mantissa bits,
// fp32's exponent range), and each fp32 product a*b becomes THREE 16-bit MFMAs into the fp32 accumulator,
//     a_hi b_hi + a_lo b_hi + a_hi b_lo        (the dropped a_lo b_lo term is 2^-16 of the product),
// i.e. 6 v_mfma_f32_32x32x16_bf16 (192 cycles) per 32-deep chunk instead of 16 v_mfma_f32_32x32x2_f32 (1 024 cycles).  Relative
// error of a product ~1e-5: between the 16-bit kernels (1e-3) and the fp32 operands (1e-7); everything else (accumulation,
// softmax, GroupNorm statistics, epilogues) is the fp32 code of this file.
#include "rsvld_common.h"

namespace {

// split-operand helpers: x = hi + lo in bf16 (round to nearest even both times)
__device__ __forceinline__ void split2(float x, bf16& hi, bf16& lo) {
    hi = (bf16)x;
    lo = (bf16)(x - (float)hi);
}
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        bf16 h, l;
        split2(x[e], h, l);
        hi[e] = h;
        lo[e] = l;
    }
}
// acc += a b with a = ah + al, b = bh + bl (three 16-bit products, small terms first)
__device__ __forceinline__ f32x16 mfma_split(const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------- convolution
// out[m][n] = sum_k A[m][k] W[n][k];  m = (b, oy, ox), k = (tap, channel), gathered on the fly.
// Workgroup: 4 waves, 64 output pixels x 64 output channels, wave (wm, wn) owns 32 x 32.  K in chunks of 32 through LDS.
// MFMA 32x32x2 f32 operand map: lane l supplies A[row l&31][k = l>>5] and B[k = l>>5][col l&31];
// D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
struct ConvF32Args {
    const float* x; const float* x2; const float* w; const float* bias; const float* rowvec; const float* res; float* out;
    int B, H, W, Cin, Cin2, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo, ups, act, rv_stride;
    float alpha, beta;
    int Ktot;
    int64_t M;
};
constexpr int CF_BK = 32, CF_LD = CF_BK + 1;   // 33-float rows: the 32 rows one ds_read_b32 touches sit on 32 banks
constexpr int CS_LD = CF_BK + 8;               // SPLIT: 40 bf16 = 80-byte rows (16-byte aligned fragments)

template <bool SPLIT>
__global__ __launch_bounds__(256) void conv_f32_kernel(ConvF32Args p) {
    __shared__ __attribute__((aligned(16))) float As[SPLIT ? 1 : 64 * CF_LD];
    __shared__ __attribute__((aligned(16))) float Bs[SPLIT ? 1 : 64 * CF_LD];
    __shared__ __attribute__((aligned(16))) bf16 Sh[SPLIT ? 4 * 64 * CS_LD : 8];   // SPLIT: A hi | A lo | W hi | W lo
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
    const int64_t m0 = (int64_t)blockIdx.x * 64;
    const int n0 = blockIdx.y * 64;
    const int lr = tid >> 3, kq = (tid & 7) * 4;   // staging: rows lr and lr + 32, floats kq .. kq+3 of the chunk

    // the two A rows this thread stages: pixel coordinates
    int64_t xb[2];
    int oy[2], ox[2];
    bool mv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int64_t m = m0 + lr + 32 * i;
        mv[i] = m < p.M;
        const int64_t mm = mv[i] ? m : 0;
        const int64_t b = mm / ((int64_t)p.Ho * p.Wo);
        const int r = (int)(mm - b * (int64_t)p.Ho * p.Wo);
        oy[i] = r / p.Wo;
        ox[i] = r - oy[i] * p.Wo;
        xb[i] = b * (int64_t)p.H * p.W;
    }
    const int Hin = p.ups ? 2 * p.H : p.H, Win = p.ups ? 2 * p.W : p.W;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    for (int k0 = 0; k0 < p.Ktot; k0 += CF_BK) {
        const int k = k0 + kq;
        const bool kv = k < p.Ktot;          // Ktot % 8 == 0: a quad is inside or outside as a whole
        int ky = 0, kx = 0, c = 0;
        const float* src = p.x;
        int cs = p.Cin;
        if (kv) {
            const int ct = p.Cin + p.Cin2;
            const int tap = k / ct;
            c = k - tap * ct;
            ky = tap / p.KW;
            kx = tap - ky * p.KW;
            if (c >= p.Cin) { src = p.x2; c -= p.Cin; cs = p.Cin2; }   // [x | x2] channel concatenation, never materialised
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (kv && mv[i]) {
                int iy = oy[i] * p.stride + ky - p.pad_t, ix = ox[i] * p.stride + kx - p.pad_l;
                if (iy >= 0 && iy < Hin && ix >= 0 && ix < Win) {
                    if (p.ups) { iy >>= 1; ix >>= 1; }
                    v = *(const f32x4*)(src + ((xb[i] + (int64_t)iy * p.W + ix) * cs + c));
                }
            }
            f32x4 wv = {0.f, 0.f, 0.f, 0.f};
            const int n = n0 + lr + 32 * i;
            if (kv && n < p.Cout) wv = *(const f32x4*)(p.w + (int64_t)n * p.Ktot + k);
            if constexpr (SPLIT) {
                bf16x4 ah, al, wh, wl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bf16 h, l;
                    split2(v[e], h, l);
                    ah[e] = h; al[e] = l;
                    split2(wv[e], h, l);
                    wh[e] = h; wl[e] = l;
                }
                const int o = (lr + 32 * i) * CS_LD + kq;
                *(bf16x4*)(Sh + o) = ah;
                *(bf16x4*)(Sh + 64 * CS_LD + o) = al;
                *(bf16x4*)(Sh + 2 * 64 * CS_LD + o) = wh;
                *(bf16x4*)(Sh + 3 * 64 * CS_LD + o) = wl;
            } else {
                float* d = As + (lr + 32 * i) * CF_LD + kq;
                d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
                float* e = Bs + (lr + 32 * i) * CF_LD + kq;
                e[0] = wv[0]; e[1] = wv[1]; e[2] = wv[2]; e[3] = wv[3];
            }
        }
        __syncthreads();
        if constexpr (SPLIT) {   // lane: A row / W row (lane & 31), k-slots 8 (lane >> 5) .. +7 of each 16-deep step
            const int ao = (wm * 32 + (lane & 31)) * CS_LD + (lane >> 5) * 8, bo = (wn * 32 + (lane & 31)) * CS_LD + (lane >> 5) * 8;
#pragma unroll
            for (int ks = 0; ks < CF_BK / 16; ++ks)
                acc = mfma_split(*(const bf16x8*)(Sh + ao + ks * 16), *(const bf16x8*)(Sh + 64 * CS_LD + ao + ks * 16),
                                 *(const bf16x8*)(Sh + 2 * 64 * CS_LD + bo + ks * 16), *(const bf16x8*)(Sh + 3 * 64 * CS_LD + bo + ks * 16), acc);
        } else {
            const float* ar = As + (wm * 32 + (lane & 31)) * CF_LD + (lane >> 5);
            const float* br = Bs + (wn * 32 + (lane & 31)) * CF_LD + (lane >> 5);
#pragma unroll
            for (int s = 0; s < CF_BK / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[2 * s], br[2 * s], acc, 0, 0, 0);
        }
        __syncthreads();
    }

    const int n = n0 + wn * 32 + (lane & 31);
    const bool nv = n < p.Cout;            // Cout % 8 == 0: lanes 2j and 2j+1 are valid together (GEGLU pairs)
    const float bv = (nv && p.bias != nullptr) ? p.bias[n] : 0.f;
    const int64_t HoWo = (int64_t)p.Ho * p.Wo;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const bool ok = nv && m < p.M;
        float v = acc[r] + bv;
        if (ok && p.rowvec != nullptr) v += p.rowvec[(m / HoWo) * p.rv_stride + n];
        if (p.act == RSVLD_ACT_GEGLU) {    // weights packed value / gate interleaved: channel 2j = value, 2j+1 = gate
            const float gate = __shfl_xor(v, 1);
            if (ok && !(n & 1)) p.out[m * (p.Cout >> 1) + (n >> 1)] = p.alpha * v * (0.5f * gate * (1.0f + erff(gate * 0.70710678118654752f)));
            continue;
        }
        if (!ok) continue;
        if (p.act == RSVLD_ACT_SILU) v = v / (1.0f + expf(-v));
        v *= p.alpha;
        if (p.res != nullptr) v += p.beta * p.res[m * p.Cout + n];
        p.out[m * p.Cout + n] = v;
    }
}

// ------------------------------------------------------------------------------------------------- GroupNorm
// statistics: grid (chunks, groups, B); partial (sum, sum of squares) in fp64 -> ws[b][g][chunk][2]; merged in chunk order
__global__ __launch_bounds__(256) void gn_f32_partial_kernel(const float* __restrict__ x, double* __restrict__ ws, int64_t HW,
                                                             int C, int cpg, int64_t pix_per_chunk) {
    const int chunk = blockIdx.x, g = blockIdx.y, b = blockIdx.z, nchunk = gridDim.x, groups = gridDim.y;
    const int64_t p0 = chunk * pix_per_chunk, p1 = min(HW, p0 + pix_per_chunk);
    const float* xb = x + ((int64_t)b * HW) * C + (int64_t)g * cpg;
    double s = 0.0, ss = 0.0;
    const int64_t total = (p1 - p0) * cpg;
    for (int64_t e = threadIdx.x; e < total; e += 256) {
        const int64_t pix = p0 + e / cpg;
        const int c = (int)(e % cpg);
        const double v = (double)xb[pix * C + c];
        s += v;
        ss += v * v;
    }
    __shared__ double sh[2][256];
    sh[0][threadIdx.x] = s;
    sh[1][threadIdx.x] = ss;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double* o = ws + (((int64_t)b * groups + g) * nchunk + chunk) * 2;
        o[0] = sh[0][0];
        o[1] = sh[1][0];
    }
}

__global__ void gn_f32_finalize_kernel(const double* __restrict__ ws, float* __restrict__ mean_var, int nchunk, int total,
                                       double inv_count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // (b, g)
    if (i >= total) return;
    double s = 0.0, ss = 0.0;
    for (int c = 0; c < nchunk; ++c) {
        s += ws[((int64_t)i * nchunk + c) * 2];
        ss += ws[((int64_t)i * nchunk + c) * 2 + 1];
    }
    const double mean = s * inv_count;
    double var = ss * inv_count - mean * mean;
    if (var < 0.0) var = 0.0;
    mean_var[2 * i] = (float)mean;
    mean_var[2 * i + 1] = (float)var;
}

__global__ __launch_bounds__(256) void gn_f32_apply_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           const float* __restrict__ mean_var, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ mod_scale1p,
                                                           const float* __restrict__ mod_shift, int64_t mod_stride, int64_t HW, int C,
                                                           int cpg, int groups, float eps, int silu, int64_t total4) {
    const int C4 = C >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int64_t pix = i / C4;
        const int c = (int)(i - pix * C4) * 4;
        const int64_t b = pix / HW;
        const f32x4 v = *(const f32x4*)(x + i * 4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int g = (c + e) / cpg;
            const float mean = mean_var[(b * groups + g) * 2], var = mean_var[(b * groups + g) * 2 + 1];
            float t = (v[e] - mean) * (1.0f / sqrtf(var + eps)) * gamma[c + e] + beta[c + e];
            if (silu) t = t / (1.0f + expf(-t));
            if (mod_scale1p != nullptr) t = t * (1.0f + mod_scale1p[pix * mod_stride + c + e]) + mod_shift[pix * mod_stride + c + e];
            o[e] = t;
        }
        *(f32x4*)(y + i * 4) = o;
    }
}

// ------------------------------------------------------------------------------------------------- attention
// One workgroup = 32 query rows, 4 waves.  Key tiles of 32.  S^T = K Q^T (rows = keys, column = the lane's query row),
// contraction over d split across the waves by 32-wide blocks (wave w owns blocks w, w+4, ...), partial S^T summed through LDS so
// that every wave holds the SAME full tile and takes identical softmax decisions; O^T[d][q] accumulates per wave for its own
// d-blocks.  P^T is consumed straight from the score registers: the contraction index of the second MFMA is ordered like the
// score registers (kk = 2s + h  ->  key (s&3) + 8(s>>2) + 4h), so V rows are fetched in that order and no shuffle is needed.
struct AttnF32Args {
    const float* q; const float* k; const float* v; float* out;
    int Nq, Nk, D;
    int64_t q_bs, q_ts, k_bs, k_ts, v_bs, v_ts, o_bs, o_ts;
    float scale;
};
constexpr int AF_MAXBLK = 4;   // d-blocks per wave: D <= 4 waves x 4 x 32 = 512

template <bool SPLIT>
__global__ __launch_bounds__(256) void attn_f32_kernel(AttnF32Args p) {
    extern __shared__ float af_smem[];
    const int D = p.D, LDK = D + 1, nblk = D >> 5;
    float* Ks = af_smem;                 // [32][D + 1]
    float* red = af_smem + 32 * LDK;     // [4][16][64]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int q0 = blockIdx.x * 32, head = blockIdx.y, b = blockIdx.z;
    const float* Qb = p.q + (int64_t)b * p.q_bs + (int64_t)head * D;
    const float* Kb = p.k + (int64_t)b * p.k_bs + (int64_t)head * D;
    const float* Vb = p.v + (int64_t)b * p.v_bs + (int64_t)head * D;
    const int qrow = min(q0 + l31, p.Nq - 1);

    // Q fragments of this wave's d-blocks: B operand, B[k = d][col = q] = Q[q][d]
    // (SPLIT: per 16-deep step ks the lane holds d = 32 blk + 16 ks + 8 h + 0..7 as a hi and a lo bf16x8 -- the same 64 registers)
    float qf[SPLIT ? 1 : AF_MAXBLK][16];
    bf16x8 qh[SPLIT ? AF_MAXBLK : 1][2], ql[SPLIT ? AF_MAXBLK : 1][2];
#pragma unroll
    for (int j = 0; j < AF_MAXBLK; ++j) {
        const int blk = w + 4 * j;
        if constexpr (SPLIT) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                float x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = blk < nblk ? Qb[(int64_t)qrow * p.q_ts + blk * 32 + ks * 16 + 8 * h + e] : 0.f;
                split8(x, qh[j][ks], ql[j][ks]);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 16; ++s) qf[j][s] = blk < nblk ? Qb[(int64_t)qrow * p.q_ts + blk * 32 + 2 * s + h] : 0.f;
        }
    }
    f32x16 oacc[AF_MAXBLK];
#pragma unroll
    for (int j = 0; j < AF_MAXBLK; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[j][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int nt = (p.Nk + 31) >> 5;
    for (int t = 0; t < nt; ++t) {
        const int key0 = t * 32;
        // ---- K tile -> LDS (rows past Nk: zeros, their scores are masked below)
        for (int e = tid; e < 32 * (D >> 2); e += 256) {
            const int row = e / (D >> 2), c4 = (e - row * (D >> 2)) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (key0 + row < p.Nk) v = *(const f32x4*)(Kb + (int64_t)(key0 + row) * p.k_ts + c4);
            float* d = Ks + row * LDK + c4;
            d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        }
        __syncthreads();
        // ---- partial S^T over this wave's d-blocks
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
        for (int j = 0; j < AF_MAXBLK; ++j) {
            const int blk = w + 4 * j;
            if (blk < nblk) {
                if constexpr (SPLIT) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const float* kr = Ks + l31 * LDK + blk * 32 + ks * 16 + 8 * h;
                        float x[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) x[e] = kr[e];
                        bf16x8 kh, kl;
                        split8(x, kh, kl);
                        sacc = mfma_split(kh, kl, qh[j][ks], ql[j][ks], sacc);
                    }
                } else {
                    const float* kr = Ks + l31 * LDK + blk * 32 + h;
#pragma unroll
                    for (int s = 0; s < 16; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[2 * s], qf[j][s], sacc, 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(w * 16 + r) * 64 + lane] = sacc[r];
        __syncthreads();
        // ---- full tile in every wave (same summation order everywhere), online softmax in the lane's query column
        float sc[16];
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s = ((red[(0 * 16 + r) * 64 + lane] + red[(1 * 16 + r) * 64 + lane]) + red[(2 * 16 + r) * 64 + lane]) +
                      red[(3 * 16 + r) * 64 + lane];
            s *= p.scale;
            const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (key >= p.Nk) s = -INFINITY;
            sc[r] = s;
            mx = fmaxf(mx, s);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);            // finite: every tile holds at least one valid key
        const float alpha = expf(m_run - m_new);         // 0 on the first tile
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sc[r] = expf(sc[r] - m_new);
            rs += sc[r];
        }
        rs += __shfl_xor(rs, 32);
        l_run = l_run * alpha + rs;
        m_run = m_new;
        // ---- O^T[d][q] = alpha O^T + V^T P^T for this wave's d-blocks; V rows straight from global memory (coalesced along d)
        // (SPLIT: k-slot (h, e) of step ks is score register 8 ks + e of lane half h, i.e. key 16 ks + 8 (e >> 2) + 4 h + (e & 3):
        //  P is taken from its registers as it stands and V is fetched in that key order)
        bf16x8 ph[2], pl[2];
        if constexpr (SPLIT) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                float x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = sc[8 * ks + e];
                split8(x, ph[ks], pl[ks]);
            }
        }
#pragma unroll
        for (int j = 0; j < AF_MAXBLK; ++j) {
            const int blk = w + 4 * j;
            if (blk < nblk) {
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[j][r] *= alpha;
                if constexpr (SPLIT) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        float x[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int key = min(key0 + 16 * ks + 8 * (e >> 2) + 4 * h + (e & 3), p.Nk - 1);   // clamped rows carry P = 0
                            x[e] = Vb[(int64_t)key * p.v_ts + blk * 32 + l31];
                        }
                        bf16x8 vh, vl;
                        split8(x, vh, vl);
                        oacc[j] = mfma_split(vh, vl, ph[ks], pl[ks], oacc[j]);
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < 16; ++s) {
                        const int key = min(key0 + (s & 3) + 8 * (s >> 2) + 4 * h, p.Nk - 1);   // clamped rows carry P = 0
                        const float vv = Vb[(int64_t)key * p.v_ts + blk * 32 + l31];
                        oacc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, sc[s], oacc[j], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();   // Ks / red are rewritten by the next tile
    }

    if (q0 + l31 >= p.Nq) return;
    const float inv = 1.0f / l_run;
    float* Ob = p.out + (int64_t)b * p.o_bs + (int64_t)head * D + (int64_t)(q0 + l31) * p.o_ts;
#pragma unroll
    for (int j = 0; j < AF_MAXBLK; ++j) {
        const int blk = w + 4 * j;
        if (blk < nblk) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = oacc[j][4 * g + e] * inv;
                *(f32x4*)(Ob + blk * 32 + 8 * g + 4 * h) = o;   // rows (reg&3) + 8 (reg>>2) + 4 h of the block
            }
        }
    }
}

__global__ void nchw_to_nhwc_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int64_t HW, int Cdst,
                                        int c_off, int zero_pad, float scale, int64_t total_pix) {
    const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= total_pix) return;
    const int64_t b = pix / HW, r = pix - b * HW;
    float* d = dst + pix * Cdst;
    if (zero_pad) {
        for (int c = 0; c < Cdst; ++c) {
            const int cs = c - c_off;
            d[c] = (cs >= 0 && cs < C) ? src[(b * C + cs) * HW + r] * scale : 0.f;
        }
    } else {
        for (int c = 0; c < C; ++c) d[c_off + c] = src[(b * C + c) * HW + r] * scale;
    }
}

// LayerNorm over the last dim: one wave per row, two passes over the row (mean, then centred variance), fp32
__global__ __launch_bounds__(256) void layernorm_f32_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            int64_t rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
        const float* xr = x + row * C;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += xr[c];
        const float mean = wave_sum(s) / (float)C;
        float ss = 0.f;
        for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; ss += d * d; }
        const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)C + eps);
        for (int c = lane; c < C; c += 64) y[row * C + c] = (xr[c] - mean) * rstd * gamma[c] + beta[c];
    }
}

__global__ void concat_c_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, int64_t rows,
                                    int C1q, int C2q) {
    const int Cq = C1q + C2q;
    const int64_t total = rows * Cq;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / Cq;
        const int c = (int)(i - r * Cq);
        ((f32x4*)o)[i] = c < C1q ? ((const f32x4*)a)[r * C1q + c] : ((const f32x4*)b)[r * C2q + (c - C1q)];
    }
}

__global__ void axpby_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, int64_t n, float sa,
                                 float sb) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        o[i] = a[i] * sa + b[i] * sb;
}

// per row: (sum |a - b|, sum |a|), fp64 accumulation, one workgroup per row (fixed reduction tree: deterministic)
__global__ __launch_bounds__(1024) void absdiff_f32_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           float* __restrict__ out, int64_t n) {
    const int64_t row = blockIdx.x;
    double sd = 0.0, sa = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const float av = a[row * n + i];
        sd += (double)fabsf(av - b[row * n + i]);
        sa += (double)fabsf(av);
    }
    __shared__ double sh[2][1024];
    sh[0][threadIdx.x] = sd;
    sh[1][threadIdx.x] = sa;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[2 * row] = (float)sh[0][0];
        out[2 * row + 1] = (float)sh[1][0];
    }
}

static int gn_f32_chunks(int64_t HW) {
    int64_t n = (HW + 4095) / 4096;
    return (int)(n < 1 ? 1 : (n > 256 ? 256 : n));
}

}  // namespace

extern "C" int rsvld_conv2d_nhwc_f32(const rsvld_conv_desc* d, void* stream) {
    if (d == nullptr || d->x == nullptr || d->w == nullptr || d->out == nullptr) return RSVLD_EINVAL;
    if (d->dtype != RSVLD_F32) return RSVLD_EINVAL;
    if ((d->x2 != nullptr) != (d->Cin2 > 0) || d->Cin2 % 8 != 0) return RSVLD_EINVAL;
    if (d->act == RSVLD_ACT_GEGLU && (d->Cout % 16 != 0 || d->residual != nullptr)) return RSVLD_EINVAL;
    if (d->Cin % 8 != 0 || d->Cout % 8 != 0 || d->B < 1 || d->H < 1 || d->W < 1 || d->KH < 1 || d->KW < 1 || d->stride < 1)
        return RSVLD_EINVAL;
    if (d->Ho < 1 || d->Wo < 1) return RSVLD_EINVAL;   // taps outside the (up-sampled) image read zeros, whatever Ho / Wo say
    ConvF32Args a;
    a.x = (const float*)d->x; a.x2 = (const float*)d->x2; a.w = (const float*)d->w; a.bias = d->bias; a.rowvec = d->rowvec;
    a.res = (const float*)d->residual; a.out = (float*)d->out;
    a.rv_stride = d->rowvec_stride > 0 ? d->rowvec_stride : d->Cout;
    a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cin2 = d->Cin2; a.Cout = d->Cout; a.KH = d->KH; a.KW = d->KW; a.stride = d->stride;
    a.pad_t = d->pad_t; a.pad_l = d->pad_l; a.Ho = d->Ho; a.Wo = d->Wo; a.ups = d->upsample ? 1 : 0; a.act = d->act;
    a.alpha = d->alpha; a.beta = d->beta;
    a.Ktot = d->KH * d->KW * (d->Cin + d->Cin2);
    a.M = (int64_t)d->B * d->Ho * d->Wo;
    const int64_t gm = cdiv64(a.M, 64);
    if (gm > 0x7fffffffLL) return RSVLD_EINVAL;
    dim3 grid((unsigned)gm, (unsigned)((d->Cout + 63) / 64));
    if (d->tune & RSVLD_TUNE_F32_SPLIT) hipLaunchKernelGGL(conv_f32_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(conv_f32_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
    return rsvld_check_launch();
}

extern "C" int64_t rsvld_groupnorm_f32_ws_bytes(int B, int HW, int C, int groups) {
    (void)C;
    if (B < 1 || HW < 1 || groups < 1) return 0;
    return (int64_t)B * groups * gn_f32_chunks(HW) * 2 * (int64_t)sizeof(double);
}

extern "C" int rsvld_groupnorm_stats_f32(const float* x, float* mean_var, int B, int HW, int C, int groups, void* ws,
                                         void* stream) {
    if (x == nullptr || mean_var == nullptr || ws == nullptr || B < 1 || HW < 1 || C < 1 || groups < 1 || C % groups != 0)
        return RSVLD_EINVAL;
    const int nchunk = gn_f32_chunks(HW), cpg = C / groups;
    const int64_t ppc = cdiv64(HW, nchunk);
    hipLaunchKernelGGL(gn_f32_partial_kernel, dim3(nchunk, groups, B), dim3(256), 0, (hipStream_t)stream, x, (double*)ws,
                       (int64_t)HW, C, cpg, ppc);
    const int total = B * groups;
    hipLaunchKernelGGL(gn_f32_finalize_kernel, dim3((total + 127) / 128), dim3(128), 0, (hipStream_t)stream, (const double*)ws,
                       mean_var, nchunk, total, 1.0 / ((double)HW * cpg));
    return rsvld_check_launch();
}

extern "C" int rsvld_groupnorm_apply_f32(const float* x, float* y, const float* mean_var, const float* gamma, const float* beta,
                                         const float* mod_scale1p, const float* mod_shift, int mod_stride, int B, int HW, int C,
                                         int groups, float eps, int silu, void* stream) {
    if (x == nullptr || y == nullptr || mean_var == nullptr || gamma == nullptr || beta == nullptr || B < 1 || HW < 1 ||
        C % 4 != 0 || groups < 1 || C % groups != 0 || (mod_scale1p == nullptr) != (mod_shift == nullptr))
        return RSVLD_EINVAL;
    const int64_t total4 = (int64_t)B * HW * (C / 4);
    int64_t blocks = cdiv64(total4, 256);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(gn_f32_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, mean_var, gamma, beta,
                       mod_scale1p, mod_shift, (int64_t)(mod_stride > 0 ? mod_stride : C), (int64_t)HW, C, C / groups, groups, eps, silu,
                       total4);
    return rsvld_check_launch();
}

static int attention_f32_launch(bool split, const float* q, const float* k, const float* v, float* out, int B, int heads, int Nq, int Nk,
                                   int D, int64_t q_batch_stride, int64_t q_tok_stride, int64_t k_batch_stride,
                                   int64_t k_tok_stride, int64_t v_batch_stride, int64_t v_tok_stride, int64_t o_batch_stride,
                                   int64_t o_tok_stride, float scale, void* stream) {
    if (q == nullptr || k == nullptr || v == nullptr || out == nullptr || B < 1 || heads < 1 || Nq < 1 || Nk < 1) return RSVLD_EINVAL;
    if (D < 32 || D % 32 != 0 || D > 128 * AF_MAXBLK) return RSVLD_EUNSUPPORTED;
    if ((q_tok_stride | k_tok_stride | v_tok_stride | o_tok_stride | q_batch_stride | k_batch_stride | v_batch_stride |
         o_batch_stride) % 4 != 0)
        return RSVLD_EINVAL;   // 16-byte vector accesses
    AttnF32Args a;
    a.q = q; a.k = k; a.v = v; a.out = out; a.Nq = Nq; a.Nk = Nk; a.D = D;
    a.q_bs = q_batch_stride; a.q_ts = q_tok_stride; a.k_bs = k_batch_stride; a.k_ts = k_tok_stride;
    a.v_bs = v_batch_stride; a.v_ts = v_tok_stride; a.o_bs = o_batch_stride; a.o_ts = o_tok_stride;
    a.scale = scale;
    const int smem = (32 * (D + 1) + 4 * 16 * 64) * (int)sizeof(float);   // 81 KiB at D = 512
    static const hipError_t attr0 = hipFuncSetAttribute((const void*)attn_f32_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (32 * 513 + 4096) * 4);
    static const hipError_t attr1 = hipFuncSetAttribute((const void*)attn_f32_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (32 * 513 + 4096) * 4);
    if (attr0 != hipSuccess || attr1 != hipSuccess) return RSVLD_ELAUNCH;
    dim3 grid((unsigned)((Nq + 31) / 32), (unsigned)heads, (unsigned)B);
    if (split) hipLaunchKernelGGL(attn_f32_kernel<true>, grid, dim3(256), smem, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(attn_f32_kernel<false>, grid, dim3(256), smem, (hipStream_t)stream, a);
    return rsvld_check_launch();
}

extern "C" int rsvld_attention_f32(const float* q, const float* k, const float* v, float* out, int B, int heads, int Nq, int Nk,
                                   int D, int64_t q_batch_stride, int64_t q_tok_stride, int64_t k_batch_stride,
                                   int64_t k_tok_stride, int64_t v_batch_stride, int64_t v_tok_stride, int64_t o_batch_stride,
                                   int64_t o_tok_stride, float scale, void* stream) {
    return attention_f32_launch(false, q, k, v, out, B, heads, Nq, Nk, D, q_batch_stride, q_tok_stride, k_batch_stride, k_tok_stride,
                                v_batch_stride, v_tok_stride, o_batch_stride, o_tok_stride, scale, stream);
}

extern "C" int rsvld_attention_f32_split(const float* q, const float* k, const float* v, float* out, int B, int heads, int Nq, int Nk,
                                         int D, int64_t q_batch_stride, int64_t q_tok_stride, int64_t k_batch_stride,
                                         int64_t k_tok_stride, int64_t v_batch_stride, int64_t v_tok_stride, int64_t o_batch_stride,
                                         int64_t o_tok_stride, float scale, void* stream) {
    return attention_f32_launch(true, q, k, v, out, B, heads, Nq, Nk, D, q_batch_stride, q_tok_stride, k_batch_stride, k_tok_stride,
                                v_batch_stride, v_tok_stride, o_batch_stride, o_tok_stride, scale, stream);
}


extern "C" int rsvld_nchw_f32_to_nhwc_f32(const float* src, float* dst, int B, int C, int H, int W, int Cdst, int c_off,
                                          int zero_pad, float scale, void* stream) {
    if (src == nullptr || dst == nullptr || B < 1 || C < 1 || H < 1 || W < 1 || c_off < 0 || c_off + C > Cdst) return RSVLD_EINVAL;
    const int64_t total = (int64_t)B * H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_f32_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, C,
                       (int64_t)H * W, Cdst, c_off, zero_pad, scale, total);
    return rsvld_check_launch();
}

extern "C" int rsvld_layernorm_f32(const float* x, float* y, const float* gamma, const float* beta, int64_t rows, int C, float eps,
                                   void* stream) {
    if (x == nullptr || y == nullptr || gamma == nullptr || beta == nullptr || rows < 1 || C < 1) return RSVLD_EINVAL;
    int64_t blocks = cdiv64(rows, 4);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(layernorm_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, gamma, beta, rows, C, eps);
    return rsvld_check_launch();
}

extern "C" int rsvld_concat_c_f32(const float* a, const float* b, float* out, int64_t rows, int C1, int C2, void* stream) {
    if (a == nullptr || b == nullptr || out == nullptr || rows < 1 || C1 < 4 || C2 < 4 || C1 % 4 != 0 || C2 % 4 != 0) return RSVLD_EINVAL;
    int64_t blocks = cdiv64(rows * ((C1 + C2) / 4), 256);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(concat_c_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, out, rows, C1 / 4, C2 / 4);
    return rsvld_check_launch();
}

extern "C" int rsvld_axpby_f32(const float* a, const float* b, float* out, int64_t n, float sa, float sb, void* stream) {
    if (a == nullptr || b == nullptr || out == nullptr || n < 1) return RSVLD_EINVAL;
    int64_t blocks = cdiv64(n, 256);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(axpby_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, out, n, sa, sb);
    return rsvld_check_launch();
}

extern "C" int rsvld_absdiff_sums_f32(const float* a, const float* b, float* out, int rows, int64_t n_per_row, void* stream) {
    if (a == nullptr || b == nullptr || out == nullptr || rows < 1 || n_per_row < 1) return RSVLD_EINVAL;
    hipLaunchKernelGGL(absdiff_f32_kernel, dim3((unsigned)rows), dim3(1024), 0, (hipStream_t)stream, a, b, out, n_per_row);
    return rsvld_check_launch();
}
