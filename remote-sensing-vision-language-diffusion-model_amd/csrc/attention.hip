// attention.hip — flash-style attention for gfx950: out = softmax(scale * Q K^T) V with an
// fp32 online softmax; the N x N score matrix never exists in HBM.
//
// D = 512 (single head: SR3 SelfAttention, VAE mid-block attention)
//   workgroup = 4 waves, 64 query rows x 32-key tiles.  Head dim 512 is too wide for one
//   wave's accumulators, so the head dimension is split: wave w owns d in [128w, 128w+128).
//     S phase : each wave computes a PARTIAL S^T = K_w Q_w^T over its d-slice (16 MFMAs,
//               Q fragments live in registers for the whole kernel), partials go to LDS;
//     softmax : 4 threads per query row sum the 4 partials, online max / sum (wave shuffles),
//               write P (16-bit) and the per-row rescale factor to LDS;
//     PV phase: wave w accumulates O^T[d-slice][q] += V^T P^T (16 MFMAs); the query row sits on
//               the lane, so the online-softmax rescale is one per-lane scalar.
//   K is staged [kv][d] (XOR-swizzled 1-KiB rows), V is transposed through registers into
//   V^T [d][kv] so both MFMA operands are ds_read_b128 along the contraction dim.
//   Next tile's K/V global loads are issued before the S phase (register prefetch).
#include "rsvld_common.h"

namespace {

struct AttnArgs {
    const void* q;
    const void* k;
    const void* v;
    void* out;
    int B, heads, Nq, Nk, D;
    int64_t q_bs, q_ts, k_bs, k_ts, v_bs, v_ts, o_bs, o_ts;
    float scale_log2e;
};

constexpr int A5_KS = 0;
constexpr int A5_VT = 32768;
constexpr int A5_SP = 65536;
constexpr int A5_SP_STRIDE = 36;  // floats per (wave,q) row: 32 + 4 pad -> conflict-free b128 writes
constexpr int A5_PS = A5_SP + 4 * 64 * A5_SP_STRIDE * 4;
constexpr int A5_AL = A5_PS + 64 * 64;
constexpr int A5_SMEM = A5_AL + 2 * 64 * 4 + 2 * 16;   // + two rescale flags (ping-pong)
constexpr float A5_DEFER_LOG2 = 8.0f;   // rescale O only when a row's running max grows by more than 2^8

template <typename T>
__global__ __launch_bounds__(512) void attn_d512_kernel(AttnArgs p) {
    // 8 waves: wave = (query half qh, d-slice dw).  Two waves share each SIMD, so one wave's LDS / barrier waits
    // are covered by the other's MFMAs; per wave: 32 query rows x 128 head dims of O (64 accumulator registers).
    constexpr int D = 512;
    typedef typename Mfma<T>::v8 v8;
    typedef typename Mfma<T>::v4 v4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem + A5_KS;
    char* VTs = smem + A5_VT;
    float* Sp = (float*)(smem + A5_SP);
    char* Ps = smem + A5_PS;
    float* alpha_s = (float*)(smem + A5_AL);
    float* l_s = alpha_s + 64;
    int* resc_flag = (int*)(l_s + 64);   // [2]: any row of this tile rescaled? (ping-pong so the reset never races)

    const int tid = threadIdx.x, lane = tid & 63, w8 = tid >> 6;
    const int dw = w8 & 3, qh = w8 >> 2;
    const int l31 = lane & 31, lh = lane >> 5;
    const int q0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const T* Qb = (const T*)p.q + (int64_t)b * p.q_bs + (int64_t)h * D;
    const T* Kb = (const T*)p.k + (int64_t)b * p.k_bs + (int64_t)h * D;
    const T* Vb = (const T*)p.v + (int64_t)b * p.v_bs + (int64_t)h * D;
    T* Ob = (T*)p.out + (int64_t)b * p.o_bs + (int64_t)h * D;

    // Q fragments (MFMA B operand: col = query row on the lane, k = d)
    v8 qf[8];
    const int qrow = q0 + qh * 32 + l31;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (qrow < p.Nq) v = *(const u32x4*)(Qb + (int64_t)qrow * p.q_ts + dw * 128 + ks * 16 + lh * 8);
        qf[ks] = __builtin_bit_cast(v8, v);
    }
    f32x16 oacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;

    // softmax role: row sq, 4 keys starting at 4*part (8 threads per row)
    const int sq = tid >> 3, part = tid & 7;
    float m_run = -INFINITY, l_run = 0.f;
    if (tid < 2) resc_flag[tid] = 0;

    // staging roles.  K: 4 x 16 B per thread.  V: a 4(kv) x 8(d) micro-block per thread, transposed in registers.
    const int kvb = tid & 7, db = tid >> 3;
    u32x4 rk[4], rv[4];
    auto load_k = [&](int t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 512 * i;
            const int row = idx >> 6, ch = idx & 63;
            const int kv = t * 32 + row;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (kv < p.Nk) v = *(const u32x4*)(Kb + (int64_t)kv * p.k_ts + ch * 8);
            rk[i] = v;
        }
    };
    auto load_v = [&](int t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kv = t * 32 + kvb * 4 + r;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (kv < p.Nk) v = *(const u32x4*)(Vb + (int64_t)kv * p.v_ts + db * 8);
            rv[r] = v;
        }
    };
    auto store_k = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 512 * i;
            const int row = idx >> 6, ch = idx & 63;
            *(u32x4*)(Ks + row * 1024 + ((ch ^ (row & 15)) << 4)) = rk[i];
        }
    };
    auto store_v = [&]() {
        v8 vin[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) vin[r] = __builtin_bit_cast(v8, rv[r]);
#pragma unroll
        for (int dd = 0; dd < 8; ++dd) {
            v4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = vin[r][dd];
            const int d = db * 8 + dd;
            *(v4*)(VTs + d * 64 + (((kvb >> 1) ^ ((d >> 2) & 3)) << 4) + (kvb & 1) * 8) = o;
        }
    };

    // Prefetch distances: the K registers are free as soon as a K tile has been written to LDS (after barrier A),
    // so K(t+2) is requested a full tile ahead; V(t+1) is requested at the top of tile t and consumed after barrier C.
    const int ntiles = (p.Nk + 31) / 32;
    load_k(0);
    load_v(0);
    store_k();
    store_v();
    __syncthreads();
    if (ntiles > 1) load_k(1);

    for (int t = 0; t < ntiles; ++t) {
        const bool more = t + 1 < ntiles;
        if (more) load_v(t + 1);

        // ---- S phase: partial S^T[kv][q] over this wave's d-slice
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int ch = dw * 16 + ks * 2 + lh;
            const v8 kf = *(const v8*)(Ks + l31 * 1024 + ((ch ^ (l31 & 15)) << 4));
            sacc = Mfma<T>::mma(kf, qf[ks], sacc);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int qq = qh * 32 + l31;
            const int kvc = 8 * g + 4 * lh;
            f32x4 v = {sacc[4 * g], sacc[4 * g + 1], sacc[4 * g + 2], sacc[4 * g + 3]};
            *(f32x4*)(Sp + (dw * 64 + qq) * A5_SP_STRIDE + kvc) = v;
        }
        __syncthreads();  // (A) partial scores visible; K tile free
        if (more) {
            store_k();
            if (t + 2 < ntiles) load_k(t + 2);
        }

        // ---- softmax over the 32 keys of this tile (8 threads per row, 4 keys each)
        {
            float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) {
                const f32x4 a = *(const f32x4*)(Sp + (ww * 64 + sq) * A5_SP_STRIDE + part * 4);
                s[0] += a[0]; s[1] += a[1]; s[2] += a[2]; s[3] += a[3];
            }
            float mx = -INFINITY;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kv = t * 32 + part * 4 + e;
                s[e] = kv < p.Nk ? s[e] * p.scale_log2e : -INFINITY;
                mx = fmaxf(mx, s[e]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 1));
            mx = fmaxf(mx, __shfl_xor(mx, 2));
            mx = fmaxf(mx, __shfl_xor(mx, 4));
            // Deferred max (cdna_hip_programming.md T13): the reference point m_run of a row only moves when the
            // tile's max exceeds it by more than 2^8, so the rescale of O (accumulators live in AGPRs: every VALU
            // touch costs accvgpr read+write) is skipped on almost every tile.  Softmax is invariant to the
            // reference point; P <= 2^8 stays exact enough in 16-bit, l and O are fp32.
            float a = 1.0f;
            if (mx > m_run + A5_DEFER_LOG2) {     // also true on the first tile (m_run = -inf)
                a = __builtin_amdgcn_exp2f(m_run - mx);
                m_run = mx;
                resc_flag[t & 1] = 1;
            }
            float pr[4], rs = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { pr[e] = __builtin_amdgcn_exp2f(s[e] - m_run); rs += pr[e]; }
            rs += __shfl_xor(rs, 1);
            rs += __shfl_xor(rs, 2);
            rs += __shfl_xor(rs, 4);
            l_run = l_run * a + rs;
            v4 pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = (T)pr[e];
            *(v4*)(Ps + sq * 64 + (((part >> 1) ^ ((sq >> 2) & 3)) << 4) + (part & 1) * 8) = pk;
            if (part == 0) alpha_s[sq] = a;
            if (tid == 0) resc_flag[(t + 1) & 1] = 0;   // next tile's flag; nobody reads it before barrier (B) of t+1
        }
        __syncthreads();  // (B) P and rescale factors visible

        // ---- PV phase: O^T[d][q] = alpha*O^T + V^T P^T over this wave's d-slice
        {
            if (resc_flag[t & 1] != 0) {   // workgroup-uniform (LDS word written before barrier B)
                const float a0 = alpha_s[qh * 32 + l31];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[dt][r] *= a0;
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int ch = 2 * ks + lh;
                const int qq = qh * 32 + l31;
                const v8 pf = *(const v8*)(Ps + qq * 64 + ((ch ^ ((qq >> 2) & 3)) << 4));
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const int d = dw * 128 + dt * 32 + l31;
                    const v8 vf = *(const v8*)(VTs + d * 64 + ((ch ^ ((d >> 2) & 3)) << 4));
                    oacc[dt] = Mfma<T>::mma(vf, pf, oacc[dt]);
                }
            }
        }
        __syncthreads();  // (C) V^T / P free
        if (more) store_v();
    }

    if (part == 0) l_s[sq] = l_run;
    __syncthreads();
    if (qrow < p.Nq) {
        const float inv = 1.0f / l_s[qh * 32 + l31];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                v4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (T)(oacc[dt][4 * g + e] * inv);
                const int d = dw * 128 + dt * 32 + 8 * g + 4 * lh;
                *(v4*)(Ob + (int64_t)qrow * p.o_ts + d) = o;
            }
    }
}

}  // namespace

namespace {

// ---------------------------------------------------------------------------------------
// D = 64 (multi-head: sgm CrossAttention / MemoryEfficientCrossAttention, ZeroCrossAttn)
//   workgroup = 4 waves, each wave owns QT x 32 query rows of one head; 64-key tiles of K and
//   V^T are double-buffered in LDS and shared by the 4 waves (one barrier per tile).
//   S^T = K Q^T is computed "swapped" (keys on the accumulator registers, the query row on the
//   lane), so the online softmax is register-local: a row's 64 scores sit in the registers of
//   lanes l and l^32 (one shuffle for max; the row sums are merged once at the end).
//   P never touches LDS: accumulator registers 8s..8s+7 are converted to 16-bit and used
//   directly as the B operand of k-step s of O^T += V^T P^T (cdna_hip_programming.md §3 "An
//   accumulator tile as the next MFMA's operand"); their k order is permuted
//   (k = 16s + 8(j>>2) + 4h + (j&3)), so V^T is STORED in LDS with that permutation and the A
//   operand stays one ds_read_b128.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int a6_off(int row, int c) { return row * 128 + ((c ^ ((row >> 1) & 7)) << 4); }

template <typename T, int QT>
__global__ __launch_bounds__(256) void attn_d64_kernel(AttnArgs p) {
    constexpr int D = 64;
    typedef typename Mfma<T>::v8 v8;
    typedef typename Mfma<T>::v4 v4;
    __shared__ __attribute__((aligned(16))) char smem[4 * 8192];  // K[2] | VT[2], 8 KiB each

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int q0 = (blockIdx.x * 4 + w) * (QT * 32), h = blockIdx.y, b = blockIdx.z;
    const T* Qb = (const T*)p.q + (int64_t)b * p.q_bs + (int64_t)h * D;
    const T* Kb = (const T*)p.k + (int64_t)b * p.k_bs + (int64_t)h * D;
    const T* Vb = (const T*)p.v + (int64_t)b * p.v_bs + (int64_t)h * D;
    T* Ob = (T*)p.out + (int64_t)b * p.o_bs + (int64_t)h * D;

    v8 qf[QT][4];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int row = q0 + qt * 32 + l31;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (row < p.Nq) v = *(const u32x4*)(Qb + (int64_t)row * p.q_ts + ks * 16 + lh * 8);
            qf[qt][ks] = __builtin_bit_cast(v8, v);
        }
    f32x16 oacc[QT][2];
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        m_run[qt] = -INFINITY;
        l_run[qt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[qt][dt][r] = 0.f;
    }

    // staging roles.  K: two 16-B chunks per thread.  V: a 4(kv) x 4(d) micro-block per thread,
    // transposed in registers and written as four 8-B pieces of V^T.
    const int kc = tid & 7, kr = tid >> 3;        // chunk kc of rows kr, kr+32
    const int dq = tid & 15, kvq = tid >> 4;      // d = 4dq.., kv = 4kvq..
    u32x4 rk[2];
    u32x2 rv[4];
    auto load_kv = [&](int t) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int kv = t * 64 + kr + 32 * i;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (kv < p.Nk) v = *(const u32x4*)(Kb + (int64_t)kv * p.k_ts + kc * 8);
            rk[i] = v;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kv = t * 64 + kvq * 4 + r;
            u32x2 v = {0u, 0u};
            if (kv < p.Nk) v = *(const u32x2*)(Vb + (int64_t)kv * p.v_ts + dq * 4);
            rv[r] = v;
        }
    };
    auto store_kv = [&](int buf) {
        char* Ks = smem + buf * 8192;
        char* VTs = smem + 16384 + buf * 8192;
#pragma unroll
        for (int i = 0; i < 2; ++i) *(u32x4*)(Ks + a6_off(kr + 32 * i, kc)) = rk[i];
        v4 vin[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) vin[r] = __builtin_bit_cast(v4, rv[r]);
        // kv = 4*kvq + r  ->  16-group g, within it 8a + 4hh + r  ->  stored position 8hh + 4a + r
        const int g = kvq >> 2, a = (kvq >> 1) & 1, hh = kvq & 1;
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
            v4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = vin[r][dd];
            const int d = dq * 4 + dd;
            *(v4*)(VTs + a6_off(d, 2 * g + hh) + a * 8) = o;
        }
    };

    const int ntiles = (p.Nk + 63) / 64;
    load_kv(0);
    store_kv(0);
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const bool more = t + 1 < ntiles;
        if (more) load_kv(t + 1);
        const char* Ks = smem + (t & 1) * 8192;
        const char* VTs = smem + 16384 + (t & 1) * 8192;

        // ---- S^T[kv][q] for both 32-key halves
        f32x16 sacc[QT][2];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[qt][kt][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const v8 kf = *(const v8*)(Ks + a6_off(kt * 32 + l31, 2 * ks + lh));
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) sacc[qt][kt] = Mfma<T>::mma(kf, qf[qt][ks], sacc[qt][kt]);
            }

        // ---- online softmax, register-local per query row (lane) + its partner lane^32
        v8 pf[QT][4];
        if ((t + 1) * 64 > p.Nk) {   // ragged last tile only (uniform branch): mask keys beyond Nk
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kv = t * 64 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (kv >= p.Nk) sacc[qt][kt][r] = -INFINITY;
                    }
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float sv = sacc[qt][kt][r] * p.scale_log2e;
                    sacc[qt][kt][r] = sv;
                    mx = fmaxf(mx, sv);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            // deferred max (T13): move a row's reference point only when the tile max exceeds it by > 2^8; the
            // rescale of O (and its accumulator-file traffic) is skipped unless some row of the wave needs it
            const bool need = mx > m_run[qt] + 8.0f;          // true on the first tile (m_run = -inf)
            float alpha = 1.0f;
            if (need) {
                alpha = __builtin_amdgcn_exp2f(m_run[qt] - mx);
                m_run[qt] = mx;
            }
            const float mref = m_run[qt];
            float rs = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(sacc[qt][kt][r] - mref);   // raw v_exp_f32: inputs <= 8, no denormal fix-up needed
                    sacc[qt][kt][r] = pv;
                    rs += pv;
                }
            l_run[qt] = l_run[qt] * alpha + rs;
            if (__any(need)) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[qt][dt][r] *= alpha;
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                v8 f;
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = (T)sacc[qt][s4 >> 1][8 * (s4 & 1) + j];
                pf[qt][s4] = f;
            }
        }

        // ---- O^T[d][q] += V^T P^T
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const v8 vf = *(const v8*)(VTs + a6_off(dt * 32 + l31, 2 * s4 + lh));
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) oacc[qt][dt] = Mfma<T>::mma(vf, pf[qt][s4], oacc[qt][dt]);
            }

        if (more) store_kv((t + 1) & 1);
        __syncthreads();
    }

#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int row = q0 + qt * 32 + l31;
        const float l_tot = l_run[qt] + __shfl_xor(l_run[qt], 32);
        if (row >= p.Nq) continue;
        const float inv = 1.0f / l_tot;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                v4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (T)(oacc[qt][dt][4 * g + e] * inv);
                *(v4*)(Ob + (int64_t)row * p.o_ts + dt * 32 + 8 * g + 4 * lh) = o;
            }
    }
}

}  // namespace

extern "C" int rsvld_attention(const void* q, const void* k, const void* v, void* out, int B, int heads, int Nq, int Nk,
                               int D, int64_t q_batch_stride, int64_t q_tok_stride, int64_t k_batch_stride,
                               int64_t k_tok_stride, int64_t v_batch_stride, int64_t v_tok_stride,
                               int64_t o_batch_stride, int64_t o_tok_stride, float scale, int dtype, void* stream) {
    if (!q || !k || !v || !out || B <= 0 || heads <= 0 || Nq <= 0 || Nk <= 0) return RSVLD_EINVAL;
    if (dtype != RSVLD_F16 && dtype != RSVLD_BF16) return RSVLD_EINVAL;
    // 16-byte vector access along d: strides must keep rows 8-element aligned
    if ((q_tok_stride | k_tok_stride | v_tok_stride | o_tok_stride | q_batch_stride | k_batch_stride | v_batch_stride |
         o_batch_stride) & 7)
        return RSVLD_EINVAL;
    if (heads > 65535 || B > 65535) return RSVLD_EINVAL;
    AttnArgs a;
    a.q = q; a.k = k; a.v = v; a.out = out;
    a.B = B; a.heads = heads; a.Nq = Nq; a.Nk = Nk; a.D = D;
    a.q_bs = q_batch_stride; a.q_ts = q_tok_stride; a.k_bs = k_batch_stride; a.k_ts = k_tok_stride;
    a.v_bs = v_batch_stride; a.v_ts = v_tok_stride; a.o_bs = o_batch_stride; a.o_ts = o_tok_stride;
    a.scale_log2e = scale * 1.4426950408889634f;
    hipStream_t s = (hipStream_t)stream;
    if (D == 512) {
        dim3 grid((unsigned)((Nq + 63) / 64), (unsigned)heads, (unsigned)B);
        if (dtype == RSVLD_F16) {
            static bool set = false;
            if (!set) {
                if (hipFuncSetAttribute((const void*)attn_d512_kernel<f16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        A5_SMEM) != hipSuccess)
                    return RSVLD_ELAUNCH;
                set = true;
            }
            hipLaunchKernelGGL(attn_d512_kernel<f16>, grid, dim3(512), A5_SMEM, s, a);
        } else {
            static bool set = false;
            if (!set) {
                if (hipFuncSetAttribute((const void*)attn_d512_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        A5_SMEM) != hipSuccess)
                    return RSVLD_ELAUNCH;
                set = true;
            }
            hipLaunchKernelGGL(attn_d512_kernel<bf16>, grid, dim3(512), A5_SMEM, s, a);
        }
        return rsvld_check_launch();
    }
    if (D == 64) {
        // 2 query tiles per wave (256 rows per workgroup) once the grid still fills the chip
        const bool big = (int64_t)((Nq + 255) / 256) * heads * B >= 512;
        const int rows = big ? 256 : 128;
        dim3 grid((unsigned)((Nq + rows - 1) / rows), (unsigned)heads, (unsigned)B);
        if (dtype == RSVLD_F16) {
            if (big) hipLaunchKernelGGL((attn_d64_kernel<f16, 2>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((attn_d64_kernel<f16, 1>), grid, dim3(256), 0, s, a);
        } else {
            if (big) hipLaunchKernelGGL((attn_d64_kernel<bf16, 2>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((attn_d64_kernel<bf16, 1>), grid, dim3(256), 0, s, a);
        }
        return rsvld_check_launch();
    }
    return RSVLD_EUNSUPPORTED;
}
