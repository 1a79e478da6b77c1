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
#include <type_traits>

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

// Fused forms (round 5, rsvld_gemv_fused: the decode step of a Llama layer in five launches instead of ~26 small ones).  All three act on
// the activation row while it is staged into LDS, or on the result before its one store -- outside the weight-streaming loop:
//   norm_w != nullptr   x <- T( x * rsqrt(mean(x^2) + eps) * norm_w )         (the RMSNorm in front of q|k|v, gate|up and lm_head)
//   glu                 x has 2 K elements [gate | up]: x <- T( T(silu(gate)) * up )   (the SwiGLU in front of down_proj)
//   residual != nullptr y <- T( residual + T(W x + b) )                        (h + o_proj(...), h + down_proj(...))
// with the roundings of the unfused torch sequence (F.rms_norm, silu(g) * u, h + linear) kept where they were.
template <typename T, int KS>
__global__ __launch_bounds__(64 * GV_WAVES) void gemv_kernel(const T* __restrict__ W, const T* __restrict__ x, const T* __restrict__ bias,
                                                              T* __restrict__ y, int N, int K, const T* __restrict__ norm_w, float eps,
                                                              const T* __restrict__ residual, int glu) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // x: K elements (+ KS = 4: 4 x GV_RPW partial sums) + 16 floats of reduction scratch
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (glu) {
        for (int i = tid * 8; i < K; i += 64 * GV_WAVES * 8) {
            float g[8], u[8];
            unpack8<T>(*(const u32x4*)(x + i), g);
            unpack8<T>(*(const u32x4*)(x + K + i), u);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
#pragma clang fp contract(off)
                const T sg = (T)silu_f(g[e]);     // (silu rounded, THEN the product: two torch kernels)
                g[e] = (float)sg * u[e];
            }
            *(u32x4*)(smem + i * 2) = pack8<T>(g);
        }
    } else if (norm_w != nullptr) {
        float* red = (float*)(smem + (size_t)K * 2 + (KS == 4 ? 4 * GV_RPW * sizeof(float) : 0));
        float ss = 0.f;
        for (int i = tid * 8; i < K; i += 64 * GV_WAVES * 8) {
            float f[8];
            unpack8<T>(*(const u32x4*)(x + i), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) ss = __builtin_fmaf(f[e], f[e], ss);
        }
        ss = wave_sum(ss);
        if (lane == 0) red[w] = ss;
        __syncthreads();
        const float inv = rsqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)K + eps);
        for (int i = tid * 8; i < K; i += 64 * GV_WAVES * 8) {
            float f[8], g[8];
            unpack8<T>(*(const u32x4*)(x + i), f);
            unpack8<T>(*(const u32x4*)(norm_w + i), g);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = f[e] * inv * g[e];
            *(u32x4*)(smem + i * 2) = pack8<T>(f);
        }
    } else {
        for (int i = tid * 8; i < K; i += 64 * GV_WAVES * 8) *(u32x4*)(smem + i * 2) = *(const u32x4*)(x + i);
    }
    __syncthreads();
    auto store = [&](int row, float v) __attribute__((always_inline)) {
        T r = (T)(v + (bias != nullptr ? (float)bias[row] : 0.f));
        if (residual != nullptr) r = (T)((float)residual[row] + (float)r);
        y[row] = r;
    };
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
            store(row0 + tid, v);
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
        if (lane == 0 && row0 + r < N) store(row0 + r, v);
    }
}

// ---- one decode step of grouped-query attention over a static key / value cache (rsvld_llama_decode_attention): rotary embedding of the
// new q and k (half-split form), the cache write at ``pos``, scores, softmax and P V for the G = n_q / n_kv query heads that share a
// kv head.  Workgroup (kv head, chunk of DA_CH keys): every thread requests its pieces of the chunk's K and V rows FIRST (the cache of a
// 32-layer decoder does not stay in L2 between steps: one memory latency per launch, not one per key), the rows go to LDS, then scores,
// chunk maximum / exponentials / sum per head and the partial output [G][128] with its (m, l) into the workspace; da_combine_kernel
// folds the chunks.  pos is read
// from DEVICE memory (the step is replayed from a hipGraph): chunks beyond it write l = 0.  head_dim = 128.
constexpr int DA_CH = 128, DA_HD = 128, DA_GMAX = 8;   // keys per workgroup: 8 kv heads x 26 chunks = 208 workgroups at the caption's ~3 300 keys
constexpr int DA_ROW = DA_HD * 2 + 16;                 // LDS row stride in bytes: four rows read by one wave instruction fall into different banks
constexpr int DA_PIECES = DA_CH * 16 / 256;            // 16-byte pieces of the K (and of the V) chunk per thread

// sum over the 16 lanes of a DPP row, in every lane: four cross-lane VALU steps (quad swaps, half-row and row mirrors) where
// __shfl_xor would be four dependent ds_bpermute round trips -- 16 of them per key group and wave made the score phase the kernel's longest
__device__ __forceinline__ float da_row16_sum(float v) {
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, true));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});     // quad_perm [1, 0, 3, 2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});     // quad_perm [2, 3, 0, 1]
    v += dpp(v, std::integral_constant<int, 0x141>{});    // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{});    // row_mirror
    return v;
}

template <typename T>
__global__ __launch_bounds__(256) void da_kernel(const T* __restrict__ qkv, const T* __restrict__ cosv, const T* __restrict__ sinv,
                                                 const long long* __restrict__ pos_p, T* __restrict__ kc, T* __restrict__ vc,
                                                 float* __restrict__ ws, int n_q, int n_kv, int max_len, float scale) {
    __shared__ __attribute__((aligned(16))) char ks[DA_CH * DA_ROW], vs[DA_CH * DA_ROW];
    __shared__ float qs[DA_GMAX][DA_HD];        // rotated queries of this kv head's group
    __shared__ float sc[DA_GMAX][DA_CH];        // scores, then exp(s - m)
    __shared__ float ml[DA_GMAX][2];
    const int kvh = blockIdx.x, c = blockIdx.y, nchunk = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int G = n_q / n_kv, p = (int)min(max(*pos_p, 0ll), (long long)max_len - 1), half = DA_HD / 2;   // (a position outside the cache is clamped, never written past it)
    float* out = ws + ((size_t)(kvh * nchunk + c) * G) * (DA_HD + 2);
    const int j0 = c * DA_CH, nkeys = min(DA_CH, p + 1 - j0);
    const bool owner = p >= j0 && p < j0 + DA_CH;
    if (nkeys > 0) {
        // ---- the chunk's rows: all requests of a thread in flight at once (rows past the prefix re-read its last row: finite, masked below)
        u32x4 kr[DA_PIECES], vr[DA_PIECES];
#pragma unroll
        for (int i = 0; i < DA_PIECES; ++i) {
            const int pc = tid + 256 * i, row = min(pc >> 4, nkeys - 1), col = pc & 15;
            const size_t off = ((size_t)kvh * max_len + j0 + row) * DA_HD + col * 8;
            kr[i] = *(const u32x4*)(kc + off);
            vr[i] = *(const u32x4*)(vc + off);
        }
        // rotary embedding as the unfused sequence rounds it: T(x cos) + T(rot(x) sin) -> T
        auto rope = [&](const T* src, int i) {
#pragma clang fp contract(off)   // (three roundings, as three torch kernels leave them: hipcc otherwise demotes the sum to ONE 16-bit fma)
            const float xr = i < half ? -(float)src[i + half] : (float)src[i - half];
            const T a = (T)((float)src[i] * (float)cosv[i]);
            const T b = (T)(xr * (float)sinv[i]);
            return (float)(T)((float)a + (float)b);
        };
        for (int i = tid; i < G * DA_HD; i += 256) qs[i / DA_HD][i % DA_HD] = rope(qkv + (size_t)(kvh * G + i / DA_HD) * DA_HD, i % DA_HD);
#pragma unroll
        for (int i = 0; i < DA_PIECES; ++i) {
            const int pc = tid + 256 * i;
            *(u32x4*)(ks + (pc >> 4) * DA_ROW + (pc & 15) * 16) = kr[i];
            *(u32x4*)(vs + (pc >> 4) * DA_ROW + (pc & 15) * 16) = vr[i];
        }
        __syncthreads();
        if (owner && tid < DA_HD) {   // the new token's key (rotated) and value: into the cache AND over the (stale) row of the LDS image
            const T kn = (T)rope(qkv + (size_t)(n_q + kvh) * DA_HD, tid), vn = qkv[(size_t)(n_q + n_kv + kvh) * DA_HD + tid];
            kc[((size_t)kvh * max_len + p) * DA_HD + tid] = kn;
            vc[((size_t)kvh * max_len + p) * DA_HD + tid] = vn;
            ((T*)(ks + (p - j0) * DA_ROW))[tid] = kn;
            ((T*)(vs + (p - j0) * DA_ROW))[tid] = vn;
        }
        __syncthreads();
        // ---- scores: a wave takes four keys per step, 16 lanes x 16 bytes per key row
        const int sub = lane >> 4, l16 = lane & 15;
        for (int jj = wv * (DA_CH / 4); jj < (wv + 1) * (DA_CH / 4); jj += 4) {
            const int j = jj + sub;
            float kf[8];
            unpack8<T>(*(const u32x4*)(ks + j * DA_ROW + l16 * 16), kf);
            for (int g = 0; g < G; ++g) {
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) d = __builtin_fmaf(kf[e], qs[g][l16 * 8 + e], d);
                d = da_row16_sum(d);
                if (l16 == 0) sc[g][j] = j < nkeys ? (float)(T)d * scale : -INFINITY;    // (the unfused path holds the scores in T)
            }
        }
        __syncthreads();
        for (int g = wv; g < G; g += 4) {          // chunk maximum, exponentials, sum: one wave per head
            constexpr int PER = DA_CH / 64;
            float v[PER], m = -INFINITY;
#pragma unroll
            for (int i = 0; i < PER; ++i) { v[i] = sc[g][lane + 64 * i]; m = fmaxf(m, v[i]); }
            m = wave_max(m);
            float l = 0.f;
#pragma unroll
            for (int i = 0; i < PER; ++i) { v[i] = __expf(v[i] - m); l += v[i]; sc[g][lane + 64 * i] = v[i]; }
            l = wave_sum(l);
            if (lane == 0) { ml[g][0] = m; ml[g][1] = l; }
        }
        __syncthreads();
        // ---- P V: wave = head (groups of four); four value rows per LDS instruction, each lane its eight channels over the keys of its quarter
        for (int g = wv; g < G; g += 4) {
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
            for (int jb = 0; jb < nkeys; jb += 4) {
                const int j = jb + sub;
                float vf[8];
                unpack8<T>(*(const u32x4*)(vs + j * DA_ROW + l16 * 16), vf);     // (rows past nkeys: copies of the last row, weight exp(-inf) = 0)
                const float pw = sc[g][j];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = __builtin_fmaf(pw, vf[e], acc[e]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                acc[e] += __shfl_xor(acc[e], 16);
                acc[e] += __shfl_xor(acc[e], 32);
            }
            float* o = out + (size_t)g * (DA_HD + 2);
            if (sub == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[l16 * 8 + e] = acc[e];
            }
            if (lane == 0) { o[DA_HD] = ml[g][0]; o[DA_HD + 1] = ml[g][1]; }
        }
    } else {                                   // nothing of this chunk is visible yet
        for (int i = tid; i < G * (DA_HD + 2); i += 256) out[i] = (i % (DA_HD + 2)) == DA_HD ? -INFINITY : 0.f;
    }
}

// folds the chunks of one query head: out = sum_c exp(m_c - M) acc_c / sum_c exp(m_c - M) l_c.  (A second launch, not a "last workgroup"
// ticket inside da_kernel: the partials of one head are written on several XCDs, and making them visible to one another inside a launch
// costs every workgroup an L2 write-back -- measured 41 us per call against 21 + 13 for two launches.)  The chunk loop is unrolled so that
// its loads are in flight together: one L2 latency, not one per chunk.
template <typename T>
__global__ __launch_bounds__(DA_HD) void da_combine_kernel(const float* __restrict__ ws, T* __restrict__ out, int n_q, int n_kv, int nchunk) {
    __shared__ float wsh[64];
    const int h = blockIdx.x, G = n_q / n_kv, kvh = h / G, g = h % G, d = threadIdx.x;
    const float* base = ws + ((size_t)kvh * nchunk * G + g) * (DA_HD + 2);
    const size_t cs = (size_t)G * (DA_HD + 2);
    float l = 0.f, acc = 0.f;
    for (int c0 = 0; c0 < nchunk; c0 += 64) {              // (<= 64 chunks per round: 8 192 keys)
        const int nc = min(64, nchunk - c0);
        float mc = -INFINITY, lc = 0.f;
        if (d < nc) { mc = base[(c0 + d) * cs + DA_HD]; lc = base[(c0 + d) * cs + DA_HD + 1]; }
        // running maximum across rounds is not needed for <= 64 chunks; for more, rescale
        const float Mr = wave_max(d < 64 ? mc : -INFINITY);
        __shared__ float Msh, Lsh, Ash;
        if (d == 0) { Msh = Mr; }
        __syncthreads();
        const float wgt = (d < nc && mc != -INFINITY) ? __expf(mc - Msh) : 0.f;
        if (d < 64) wsh[d] = wgt;
        const float lsum = wave_sum(d < 64 ? wgt * lc : 0.f);
        if (d == 0) Lsh = lsum;
        __syncthreads();
        float a = 0.f;
#pragma unroll 8
        for (int c = 0; c < nc; ++c) a = __builtin_fmaf(wsh[c], base[(c0 + c) * cs + d], a);
        if (c0 == 0) { acc = a; l = Lsh; Ash = Msh; }
        else {   // a later round: bring both to the larger maximum
            const float Mo = Ash, Mn = fmaxf(Mo, Msh), so = __expf(Mo - Mn), sn = __expf(Msh - Mn);
            acc = acc * so + a * sn; l = l * so + Lsh * sn;
            __syncthreads();
            if (d == 0) Ash = Mn;
        }
        __syncthreads();
    }
    out[(size_t)h * DA_HD + d] = (T)(acc / l);
}

}  // namespace

static int gemv_launch(const void* w, const void* x, const void* bias, const void* norm_w, float eps, const void* residual, int glu, void* y,
                       int N, int K, int dtype, void* stream) {
    if (!w || !x || !y || N <= 0 || K <= 0) return RSVLD_EINVAL;
    if (dtype != RSVLD_F16 && dtype != RSVLD_BF16) return RSVLD_EINVAL;
    if (K % 8 != 0 || (size_t)K * 2 + 4 * GV_RPW * sizeof(float) + 16 * sizeof(float) > 65536) return RSVLD_EUNSUPPORTED;   // 16-byte pieces; x + scratch within the 64 KiB of dynamic LDS a launch gets without an opt-in
    if (((uintptr_t)w | (uintptr_t)x | (uintptr_t)norm_w) & 15) return RSVLD_EINVAL;
    if (glu && (norm_w != nullptr || (((uintptr_t)x + (uintptr_t)K * 2) & 15))) return RSVLD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    // few rows (fewer than ~4 row-split workgroups per CU of a 256-CU chip): the four waves of a workgroup split K instead
    const bool ksplit = K % 2048 == 0 && (N + GV_WAVES * GV_RPW - 1) / (GV_WAVES * GV_RPW) < 1024;
    const dim3 grid(ksplit ? (unsigned)((N + GV_RPW - 1) / GV_RPW) : (unsigned)((N + GV_WAVES * GV_RPW - 1) / (GV_WAVES * GV_RPW)));
    const size_t smem = (size_t)K * 2 + (ksplit ? 4 * GV_RPW * sizeof(float) : 0) + 16 * sizeof(float);
    const f16* wh = (const f16*)w; const f16* xh = (const f16*)x; const f16* bh = (const f16*)bias;
    const bf16* wb = (const bf16*)w; const bf16* xb = (const bf16*)x; const bf16* bb = (const bf16*)bias;
    if (dtype == RSVLD_F16) {
        if (ksplit) hipLaunchKernelGGL((gemv_kernel<f16, 4>), grid, dim3(64 * GV_WAVES), smem, s, wh, xh, bh, (f16*)y, N, K, (const f16*)norm_w, eps, (const f16*)residual, glu);
        else hipLaunchKernelGGL((gemv_kernel<f16, 1>), grid, dim3(64 * GV_WAVES), smem, s, wh, xh, bh, (f16*)y, N, K, (const f16*)norm_w, eps, (const f16*)residual, glu);
    } else {
        if (ksplit) hipLaunchKernelGGL((gemv_kernel<bf16, 4>), grid, dim3(64 * GV_WAVES), smem, s, wb, xb, bb, (bf16*)y, N, K, (const bf16*)norm_w, eps, (const bf16*)residual, glu);
        else hipLaunchKernelGGL((gemv_kernel<bf16, 1>), grid, dim3(64 * GV_WAVES), smem, s, wb, xb, bb, (bf16*)y, N, K, (const bf16*)norm_w, eps, (const bf16*)residual, glu);
    }
    return rsvld_check_launch();
}

extern "C" int rsvld_gemv(const void* w, const void* x, const void* bias, void* y, int N, int K, int dtype, void* stream) {
    return gemv_launch(w, x, bias, nullptr, 0.f, nullptr, 0, y, N, K, dtype, stream);
}

extern "C" int rsvld_gemv_fused(const void* w, const void* x, const void* bias, const void* norm_w, float norm_eps, const void* residual, int glu,
                                void* y, int N, int K, int dtype, void* stream) {
    return gemv_launch(w, x, bias, norm_w, norm_eps, residual, glu ? 1 : 0, y, N, K, dtype, stream);
}

extern "C" size_t rsvld_llama_decode_attention_ws_bytes(int n_q, int n_kv, int max_len) {
    if (n_q <= 0 || n_kv <= 0 || max_len <= 0) return 0;
    return (size_t)n_q * ((max_len + DA_CH - 1) / DA_CH) * (DA_HD + 2) * sizeof(float);
}

extern "C" int rsvld_llama_decode_attention(const void* qkv, const void* cosv, const void* sinv, const int64_t* pos, void* kcache, void* vcache,
                                            void* out, float* ws, int n_q, int n_kv, int head_dim, int max_len, float scale, int dtype,
                                            void* stream) {
    if (!qkv || !cosv || !sinv || !pos || !kcache || !vcache || !out || !ws) return RSVLD_EINVAL;
    if (dtype != RSVLD_F16 && dtype != RSVLD_BF16) return RSVLD_EINVAL;
    if (n_q <= 0 || n_kv <= 0 || max_len <= 0) return RSVLD_EINVAL;
    if (head_dim != DA_HD || n_q % n_kv != 0 || n_q / n_kv > DA_GMAX) return RSVLD_EUNSUPPORTED;
    if (((uintptr_t)kcache | (uintptr_t)vcache | (uintptr_t)qkv) & 15) return RSVLD_EINVAL;
    const int nchunk = (max_len + DA_CH - 1) / DA_CH;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == RSVLD_F16) {
        hipLaunchKernelGGL((da_kernel<f16>), dim3(n_kv, nchunk), dim3(256), 0, s, (const f16*)qkv, (const f16*)cosv, (const f16*)sinv,
                           (const long long*)pos, (f16*)kcache, (f16*)vcache, ws, n_q, n_kv, max_len, scale);
        hipLaunchKernelGGL((da_combine_kernel<f16>), dim3(n_q), dim3(DA_HD), 0, s, ws, (f16*)out, n_q, n_kv, nchunk);
    } else {
        hipLaunchKernelGGL((da_kernel<bf16>), dim3(n_kv, nchunk), dim3(256), 0, s, (const bf16*)qkv, (const bf16*)cosv, (const bf16*)sinv,
                           (const long long*)pos, (bf16*)kcache, (bf16*)vcache, ws, n_q, n_kv, max_len, scale);
        hipLaunchKernelGGL((da_combine_kernel<bf16>), dim3(n_q), dim3(DA_HD), 0, s, ws, (bf16*)out, n_q, n_kv, nchunk);
    }
    return rsvld_check_launch();
}
