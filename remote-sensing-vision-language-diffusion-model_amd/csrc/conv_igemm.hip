// conv_igemm.hip — implicit-GEMM convolution / linear for gfx950 on v_mfma_f32_32x32x16.
//
//   OutT[co][m] = sum_k Wp[co][k] * A[m][k],   m = (b,oy,ox) output pixel, k = (ky,kx,ci)
//
// The MFMA "A" operand is the weight tile (rows = output channels), the "B" operand is the
// im2col activation tile (columns = output pixels), so each lane ends up owning ONE pixel
// and, per accumulator register quad, FOUR CONSECUTIVE output channels: the epilogue can
// then move 16 B per lane into the LDS out-tile and leave HBM in full 16-byte NHWC rows.
//
// Data movement, two staging variants selected at run time (A/B-able on hardware):
//  GLDS (default): HBM --global_load_lds_dwordx4 (LDS-DMA; im2col gather through the per-lane SOURCE
//       address; padded taps read a zero page)--> LDS, no VGPR round trip.  The LDS image is
//       lane-linear per wave (8 rows x 128 B), so the XOR swizzle is applied to the source chunk
//       each lane fetches (cdna_hip_programming.md §5.4 rule 21).  Motivation: with register staging
//       the ds_write_b128 path (~79 B/clk/CU) carries 2 KiB per 32-cycle MFMA slot = ~80 % busy.
//  REG : HBM --global_load_dwordx4 (zero-fill by predicate)--> VGPR --ds_write_b128--> LDS.
// Both: double-buffered BK = 64 tiles, next tile's loads issued BEFORE the current tile's MFMAs,
// one barrier per K-step; ds_read_b128 of XOR-swizzled 128-B rows is conflict-free.
// Workgroup -> tile order is XCD-aware: the 8 XCDs get contiguous runs of M tiles (same weight
// tile, adjacent image rows) so halo rows and weights hit that XCD's private L2.
//
// Fusions: nearest x2 up-sampling of the input (index >>1), channel concatenation of two
// inputs, bias, per-image row vector (time embedding), residual add, SiLU / GEGLU.
//
// dtype RSVLD_SPLIT (round 4, the split-operand precision on this tiling; see gemm.hip): x / x2 are bf16 PLANES [.., lo(C) | hi(C)]
// of fp32 activations, the weights are packed per tap as the triple [W_hi | W_lo | W_hi] over the concatenated channels, and the
// kernel runs as a bf16 convolution over 3 (Cin + Cin2) logical channels per tap whose third segment re-reads the hi planes:
// x_lo W_hi + x_hi W_lo + x_hi W_hi.  Residual fp32; output fp32 (out_f32 = 1), planes (0) or fp16 (2).
// dtype RSVLD_F16W2 (round 5): fp16 activations, weights per tap as the fp16 pair [W_lo | W_hi]: an fp16 convolution over 2 (Cin + Cin2)
// logical channels per tap whose second segment re-reads the activation.  Residual fp32; output fp32 (out_f32 = 1) or fp16 (0).
#include <stdlib.h>
#include <string.h>

#include "rsvld_common.h"
#include <type_traits>

namespace {

struct ConvArgs {
    const void* x;
    const void* x2;
    const void* w;
    const float* bias;
    const float* rowvec;
    const void* residual;
    void* out;
    const void* zero;   // 16 zero bytes in device memory: source of padded / out-of-range chunks for the LDS-DMA path
    int B, H, W, Cin, Cin2, Cout;
    int KH, KW, stride, pad_t, pad_l, Ho, Wo, upsample;
    int out_f32, act;
    float alpha, beta;
    int M;       // B*Ho*Wo
    int HoWo;
    int C1_8;    // Cin/8
    int Ctot8;   // (Cin+Cin2)/8
    int KC;      // KH*KW*Ctot8 : K in 8-element chunks
    int nk;      // K-steps of 64 elements
    int Cout_out;  // channels of the stored tensor (Cout, or Cout/2 for GEGLU)
    int rv_stride; // row stride of rowvec
    int M_plan;    // rows the launch plan is made for (M / plan_div)
    int tune;      // RSVLD_TUNE_*
    int seg;       // K segments per tap: 1 plain; 3 RSVLD_SPLIT (planes in, triple weights); 2 RSVLD_F16W2 (fp16 in, pair weights).
                   // Ctot8 / KC count the seg x (Cin + Cin2) LOGICAL channels
    int out_kind;  // seg > 1: 0 = fp16 out, 1 = fp32 out, 2 = bf16 planes out (seg == 1: out_f32 decides)
    int Cseg8;     // (Cin + Cin2) / 8, the chunks of one segment
    int C2_8;      // Cin2 / 8
};

constexpr int BK_BYTES = 128;  // 64 x 16-bit per LDS row

__device__ uint4 g_zero16;  // zero page: source of padded / out-of-range chunks for the LDS-DMA path

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// byte offset of 16-byte chunk `c` (0..7) of LDS row `row` (128-B rows).  Two rows share
// one 256-B bank row; XOR with (row>>1)&7 makes every ds_read_b128 lane group hit 16
// distinct 16-B slots (MI355X_MICROARCH.md §LDS).
__device__ __forceinline__ int lds_off(int row, int c) { return row * BK_BYTES + ((c ^ ((row >> 1) & 7)) << 4); }

// counted wait: all but the newest `n` LDS-DMA (vector-memory) operations of this wave have landed
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// KS = intra-workgroup split of K: KS groups of 4 waves each own every KS-th K-step of the SAME output tile (own
// LDS ring, shared barriers) and their accumulators are summed through LDS before the epilogue.  For small-M layers
// whose grid is one workgroup per CU this doubles the waves per SIMD without partial sums in HBM.
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, bool GLDS, int STAGES, int KS, int SEG = 1>
__global__ __launch_bounds__(256 * KS) void conv_igemm_kernel(ConvArgs p) {
    constexpr bool SPLIT = SEG == 3;
    static_assert(GLDS ? (STAGES >= 2 && STAGES <= 4) : STAGES == 2, "register staging is double-buffered");
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per K group");
    static_assert(KS == 1 || (GLDS && STAGES > 2), "split K runs on the LDS-DMA ring");
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int A_LOADS = BM / 32, B_LOADS = BN / 32;
    constexpr int A_BYTES = BM * BK_BYTES, B_BYTES = BN * BK_BYTES;
    constexpr int STAGE = A_BYTES + B_BYTES;
    typedef typename Mfma<T>::v8 v8;

    extern __shared__ __attribute__((aligned(16))) char smem_all[];

    const int kg = KS > 1 ? (int)(threadIdx.x >> 8) : 0;   // K group (wave-uniform)
    const int tid = threadIdx.x & 255;                      // thread inside its K group
    char* smem = smem_all + kg * (STAGES * STAGE);
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    // XCD-aware tile order: linear workgroup id -> (XCD label, slot); each XCD walks a contiguous run of
    // tiles with m fastest (bijective for any grid size, cdna_hip_programming.md §5 "XCD swizzle")
    int tile_m, tile_n;
    {
        const int nmt = gridDim.x, nwg = gridDim.x * gridDim.y;
        const int lid = blockIdx.x + blockIdx.y * nmt;
        const int q = nwg >> 3, r = nwg & 7, xcd = lid & 7, slot = lid >> 3;
        const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
        tile_n = t / nmt;
        tile_m = t - tile_n * nmt;
    }
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    // ---- staging roles: thread -> 16-byte chunk of rows r0, r0+32, ...  With LDS-DMA the lane's LDS
    // slot is fixed (row r0, physical chunk tid&7), so it fetches the LOGICAL chunk that the swizzle
    // maps there: c = (tid&7) ^ ((r0>>1)&7)  ((row>>1)&7 is the same for r0 and r0+32i).
    const int r0 = tid >> 3;
    const int c = GLDS ? ((tid & 7) ^ ((r0 >> 1) & 7)) : (tid & 7);

    int a_pix[A_LOADS], a_iy0[A_LOADS], a_ix0[A_LOADS];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        const int m = m0 + r0 + 32 * i;
        if (m < p.M) {
            const int n = m / p.HoWo;
            const int rem = m - n * p.HoWo;
            const int oy = rem / p.Wo;
            const int ox = rem - oy * p.Wo;
            a_pix[i] = n * p.H * p.W;
            a_iy0[i] = oy * p.stride - p.pad_t;
            a_ix0[i] = ox * p.stride - p.pad_l;
        } else {
            a_pix[i] = 0;
            a_iy0[i] = -(1 << 28);  // never valid
            a_ix0[i] = 0;
        }
    }
    const int Hlim = p.upsample ? 2 * p.H : p.H;
    const int Wlim = p.upsample ? 2 * p.W : p.W;
    const int ush = p.upsample ? 1 : 0;

    // K position of this thread's chunk: (ky, kx, ci) advanced by 8 chunks per K-step
    int ci = c + 8 * kg, ky = 0, kx = 0;
    auto normalize = [&]() {
        while (ci >= p.Ctot8) {
            ci -= p.Ctot8;
            if (++kx == p.KW) { kx = 0; ++ky; }
        }
    };
    normalize();

    const T* __restrict__ X1 = (const T*)p.x;
    const T* __restrict__ X2 = (const T*)p.x2;
    const T* __restrict__ Wp = (const T*)p.w;
    const int64_t Kel = (int64_t)p.KC * 8;

    u32x4 ra[A_LOADS], rb[B_LOADS];

    // logical 8-channel chunk -> (source tensor, chunk inside a pixel's row, elements per pixel).  Split: segment 0 reads the lo
    // planes, segments 1 and 2 the hi planes; a pixel's row holds lo | hi.  Written as selects on values (the chunk index is per
    // LANE): with assignments under nested branches hipcc built a two-entry pointer table on the stack and indexed it with a scratch
    // load inside the K loop -- whose vmcnt wait drained the LDS-DMA ring (conv_igemm_64x128: 586 -> 437 TFLOP/s until noticed).
    auto src_of = [&](int ch, const T*& src, int& cc, int& Cs) __attribute__((always_inline)) {
        int hi = 0;
        if (SPLIT) {                                  // (compile-time)
            const bool s1 = ch >= p.Cseg8;
            ch -= s1 ? p.Cseg8 : 0;
            ch -= ch >= p.Cseg8 ? p.Cseg8 : 0;
            hi = s1 ? 1 : 0;
        } else if (SEG == 2) {                        // the pair form reads the one activation twice
            ch -= ch >= p.Cseg8 ? p.Cseg8 : 0;
        }
        const bool first = ch < p.C1_8;
        src = first ? X1 : X2;
        const int c1 = first ? p.C1_8 : p.C2_8;         // 8-channel chunks of one plane of the chosen source
        cc = (first ? ch : ch - p.C1_8) + hi * c1;
        Cs = (SPLIT ? 16 : 8) * c1;
    };

    auto load_tile = [&](int kt) {
        const bool kvalid = ky < p.KH;
        const T* src;
        int cc, Cs;
        src_of(ci, src, cc, Cs);
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
            const bool valid = kvalid && (unsigned)iy < (unsigned)Hlim && (unsigned)ix < (unsigned)Wlim;
            iy >>= ush; ix >>= ush;
            const int64_t off = ((int64_t)(a_pix[i] + iy * p.W + ix)) * Cs + cc * 8;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (valid) v = *(const u32x4*)(src + off);
            ra[i] = v;
        }
        const int q = kt * 8 + c;
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            const int n = n0 + r0 + 32 * i;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (n < p.Cout && q < p.KC) v = *(const u32x4*)(Wp + (int64_t)n * Kel + (int64_t)q * 8);
            rb[i] = v;
        }
    };
    // LDS-DMA: one wave instruction fills 8 consecutive rows (1 KiB) of the tile
    auto dma_tile = [&](int kt, int buf) {
        char* a_s = smem + buf * STAGE;
        char* b_s = a_s + A_BYTES;
        const bool kvalid = ky < p.KH;
        const T* src;
        int cc, Cs;
        src_of(ci, src, cc, Cs);
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
            const bool valid = kvalid && (unsigned)iy < (unsigned)Hlim && (unsigned)ix < (unsigned)Wlim;
            iy >>= ush; ix >>= ush;
            const int64_t off = ((int64_t)(a_pix[i] + iy * p.W + ix)) * Cs + cc * 8;
            const void* g = valid ? (const void*)(src + off) : p.zero;
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(a_s + (wave * 8 + 32 * i) * BK_BYTES), 16, 0, 0);
        }
        const int q = kt * 8 + c;
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            const int n = n0 + r0 + 32 * i;
            const void* g = (n < p.Cout && q < p.KC) ? (const void*)(Wp + (int64_t)n * Kel + (int64_t)q * 8)
                                                     : p.zero;
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(b_s + (wave * 8 + 32 * i) * BK_BYTES), 16, 0, 0);
        }
    };
    auto store_tile = [&](int buf) {
        char* a_s = smem + buf * STAGE;
        char* b_s = a_s + A_BYTES;
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) *(u32x4*)(a_s + lds_off(r0 + 32 * i, c)) = ra[i];
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) *(u32x4*)(b_s + lds_off(r0 + 32 * i, c)) = rb[i];
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;
    auto compute_tile = [&](int buf) {
        const char* a_s = smem + buf * STAGE;
        const char* b_s = a_s + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int ch = 2 * ks + lh;
            v8 fa[TN], fb[TM];
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) fa[ni] = *(const v8*)(b_s + lds_off(wn * WTN + ni * 32 + l31, ch));
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) fb[mi] = *(const v8*)(a_s + lds_off(wm * WTM + mi * 32 + l31, ch));
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) acc[ni][mi] = Mfma<T>::mma(fa[ni], fb[mi], acc[ni][mi]);
        }
    };

    if (GLDS && STAGES > 2) {
        // ---- LDS-DMA ring, STAGES buffers, STAGES-1 tiles in flight (cdna_hip_programming.md §5
        // "Pipelining across barriers"): per K-step  counted vmcnt -> raw s_barrier -> issue tile
        // kt+STAGES-1 into the buffer read in step kt-1 -> MFMAs of tile kt.  The barrier both publishes
        // tile kt (every wave waited for its own DMA pieces) and retires the reads of tile kt-1.
        // With KS groups, group kg walks the global K-steps kg, kg+KS, ...; every wave executes the same
        // number of barriers (the step count of group 0), a group that has run out only waits.
        constexpr int L = A_LOADS + B_LOADS;   // DMA instructions per wave per tile
        const int nk_g = (p.nk - kg + KS - 1) / KS;      // this group's K-steps
        const int nk_0 = (p.nk + KS - 1) / KS;           // barriers of the loop
        int issued = 0;
        for (; issued < STAGES - 1 && issued < nk_g; ++issued) {
            if (issued > 0) { ci += 8 * KS; normalize(); }
            dma_tile(issued * KS + kg, issued);
        }
        for (int kt = 0; kt < nk_0; ++kt) {
            const int ahead = issued - 1 - kt;   // tiles issued after tile kt: -1 (group ran out) .. STAGES-2
            if (ahead >= 2) wait_vmcnt<2 * L>();
            else if (ahead == 1) wait_vmcnt<L>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (issued < nk_g) {
                ci += 8 * KS;
                normalize();
                dma_tile(issued * KS + kg, issued % STAGES);
                ++issued;
            }
            if (KS == 1 || kt < nk_g) compute_tile(kt % STAGES);
        }
        __builtin_amdgcn_s_barrier();          // all waves done with the ring before the epilogue reuses it
    } else {
        if (GLDS) {
            dma_tile(0, 0);
        } else {
            load_tile(0);
            store_tile(0);
        }
        __syncthreads();
        for (int kt = 0; kt < p.nk; ++kt) {
            const bool more = kt + 1 < p.nk;
            if (more) {
                ci += 8;
                normalize();
                if (GLDS) dma_tile(kt + 1, (kt + 1) & 1);
                else load_tile(kt + 1);
            }
            compute_tile(kt & 1);
            if (!GLDS && more) store_tile((kt + 1) & 1);
            __syncthreads();   // with LDS-DMA in flight this also waits vmcnt(0): next tile has landed
        }
    }

    // ---- epilogue: accumulators -> LDS out tile Ct[pixel][cout] (fp32, row stride BN+4)
    constexpr int CT_STRIDE = BN + 4;
    float* Ct = (float*)smem_all;
    constexpr int CPR = BN / 8;               // 8-channel chunks per tile row
    constexpr int RPP = 256 * KS / CPR;       // rows per pass
    constexpr int RPT = (BM + RPP - 1) / RPP; // rows each thread stores
    const int cc = (int)threadIdx.x % CPR;
    const int rr = (int)threadIdx.x / CPR;
    const int n = n0 + cc * 8;
    // the residual pieces this thread will add on the way out are requested before the staging pass (their HBM round trip
    // runs under it; same change as in gemm.hip / conv_halo.hip)
    u32x4 rres[RPT];
    if (p.residual != nullptr && n < p.Cout && p.act != RSVLD_ACT_GEGLU && SEG == 1) {
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const int row = rr + j * RPP;
            const int64_t m = m0 + row;
            rres[j] = (row < BM && m < p.M) ? *(const u32x4*)((const T*)p.residual + m * p.Cout_out + n) : u32x4{0u, 0u, 0u, 0u};
        }
    }
#pragma unroll
    for (int g2 = 0; g2 < KS; ++g2) {    // K groups add their partial tiles one after the other (fixed order)
        if (kg == g2) {
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int row = wm * WTM + mi * 32 + l31;
                        const int col = wn * WTN + ni * 32 + 8 * g + 4 * lh;
                        f32x4 v = {acc[ni][mi][4 * g], acc[ni][mi][4 * g + 1], acc[ni][mi][4 * g + 2], acc[ni][mi][4 * g + 3]};
                        if (g2 > 0) v += *(const f32x4*)(Ct + row * CT_STRIDE + col);
                        *(f32x4*)(Ct + row * CT_STRIDE + col) = v;
                    }
        }
        __syncthreads();
    }

    if (n >= p.Cout) return;
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = (p.bias != nullptr && n + e < p.Cout) ? p.bias[n + e] : 0.f;

#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        const int row = rr + j * RPP;
        const int m = m0 + row;
        if (row >= BM || m >= p.M) break;
        const f32x4 v0 = *(const f32x4*)(Ct + row * CT_STRIDE + cc * 8);
        const f32x4 v1 = *(const f32x4*)(Ct + row * CT_STRIDE + cc * 8 + 4);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bv[e];
        if (p.rowvec != nullptr) {
            const int img = m / p.HoWo;
            const float* rv = p.rowvec + (int64_t)img * p.rv_stride + n;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rv[e];
        }
        if (SEG > 1) {   // fp32 residual; fp32, planes (lo | hi per row) or fp16 out
            if (p.act == RSVLD_ACT_GEGLU) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = p.alpha * v[2 * e] * gelu_erf_f(v[2 * e + 1]);
                if (p.out_kind == 1) {
                    *(f32x4*)((float*)p.out + (int64_t)m * p.Cout_out + (n >> 1)) = (f32x4){o[0], o[1], o[2], o[3]};
                } else if (p.out_kind == 0) {
                    f16x4 ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[e] = (f16)o[e];
                    *(f16x4*)((f16*)p.out + (int64_t)m * p.Cout_out + (n >> 1)) = ov;
                } else {
                    typename Mfma<T>::v4 oh, ol;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { oh[e] = (T)o[e]; ol[e] = (T)(o[e] - (float)oh[e]); }
                    T* ob = (T*)p.out + (int64_t)m * (2 * p.Cout_out) + (n >> 1);
                    *(typename Mfma<T>::v4*)ob = ol;
                    *(typename Mfma<T>::v4*)(ob + p.Cout_out) = oh;
                }
                continue;
            }
            if (p.act == RSVLD_ACT_SILU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
            if (p.residual != nullptr) {   // the residual has the output's type: fp32, or fp16 beside the pair form's fp16 output
                if (SEG == 2 && p.out_kind == 0) {
                    float rf[8];
                    unpack8<f16>(*(const u32x4*)((const f16*)p.residual + (int64_t)m * p.Cout_out + n), rf);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += p.beta * rf[e];
                } else {
                    const float* r = (const float*)p.residual + (int64_t)m * p.Cout_out + n;
                    const f32x4 r0 = *(const f32x4*)r, r1 = *(const f32x4*)(r + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += p.beta * r0[e]; v[4 + e] += p.beta * r1[e]; }
                }
            }
            if (p.out_kind == 1) {
                float* o = (float*)p.out + (int64_t)m * p.Cout_out + n;
                *(f32x4*)o = (f32x4){v[0], v[1], v[2], v[3]};
                *(f32x4*)(o + 4) = (f32x4){v[4], v[5], v[6], v[7]};
            } else if (p.out_kind == 0) {
                *(u32x4*)((f16*)p.out + (int64_t)m * p.Cout_out + n) = pack8<f16>(v);
            } else {
                float lo[8];
                typename Mfma<T>::v8 hv;
#pragma unroll
                for (int e = 0; e < 8; ++e) { hv[e] = (T)v[e]; lo[e] = v[e] - (float)hv[e]; }
                T* ob = (T*)p.out + (int64_t)m * (2 * p.Cout_out) + n;
                *(u32x4*)ob = pack8<T>(lo);
                *(u32x4*)(ob + p.Cout_out) = __builtin_bit_cast(u32x4, hv);
            }
            continue;
        }
        if (p.act == RSVLD_ACT_GEGLU) {
            // channels are (value, gate) interleaved: 4 outputs per 8 accumulators
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = p.alpha * v[2 * e] * gelu_erf_f(v[2 * e + 1]);
            typename Mfma<T>::v4 ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[e] = (T)o[e];
            *(typename Mfma<T>::v4*)((T*)p.out + (int64_t)m * p.Cout_out + (n >> 1)) = ov;
            continue;
        }
        if (p.act == RSVLD_ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
        if (p.residual != nullptr) {
            float rf[8];
            unpack8<T>(rres[j], rf);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += p.beta * rf[e];
        }
        if (p.out_f32) {
            float* o = (float*)p.out + (int64_t)m * p.Cout_out + n;
            *(f32x4*)o = (f32x4){v[0], v[1], v[2], v[3]};
            *(f32x4*)(o + 4) = (f32x4){v[4], v[5], v[6], v[7]};
        } else {
            *(u32x4*)((T*)p.out + (int64_t)m * p.Cout_out + n) = pack8<T>(v);
        }
    }
}

// one-time registration of a kernel's dynamic LDS size: a function-local static of an instantiation keyed on the KERNEL (a generic
// lambda's static would be shared by every kernel of one function type: the plain and the multi-segment instantiations of a tile)
template <auto KERN> hipError_t conv_smem_once(int smem) {
    static const hipError_t attr = hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    return attr;
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, bool GLDS, int STAGES = 2, int KS = 1>
int launch_conv(const ConvArgs& a, hipStream_t s) {
    constexpr int stage = KS * STAGES * (BM + BN) * BK_BYTES;
    constexpr int epi = BM * (BN + 4) * 4;
    constexpr int smem = stage > epi ? stage : epi;
    dim3 grid((unsigned)((a.M + BM - 1) / BM), (unsigned)((a.Cout + BN - 1) / BN));
    auto go = [&](auto kern_c) -> int {
        constexpr auto kern = decltype(kern_c)::value;
        if (conv_smem_once<kern>(smem) != hipSuccess) return RSVLD_ELAUNCH;   // one-time, thread-safe (C++11 magic static), per KERNEL
        hipLaunchKernelGGL(kern, grid, dim3(256 * KS), smem, s, a);
        return rsvld_check_launch();
    };
#define CONV_K(SEG) std::integral_constant<void (*)(ConvArgs), &conv_igemm_kernel<T, BM, BN, WAVES_M, WAVES_N, GLDS, STAGES, KS, SEG>>{}
    if constexpr (__is_same(T, bf16) && GLDS) {   // RSVLD_SPLIT: its own instantiation (bf16, LDS-DMA staging), so that the 16-bit kernels stay as they were
        if (a.seg == 3) return go(CONV_K(3));
        if (a.seg != 1) return RSVLD_EUNSUPPORTED;
    } else if constexpr (__is_same(T, f16) && GLDS) {   // RSVLD_F16W2 likewise (fp16, LDS-DMA staging)
        if (a.seg == 2) return go(CONV_K(2));
        if (a.seg != 1) return RSVLD_EUNSUPPORTED;
    } else {
        if (a.seg != 1) return RSVLD_EUNSUPPORTED;
    }
    return go(CONV_K(1));
#undef CONV_K
}

// Tile choice.  Cout <= 32: 256x32.  Cout <= 64: 128x64 (48 KiB LDS -> 3 workgroups per CU: these layers
// have few K-steps per tile, so co-resident workgroups hide each other's prologue / epilogue).  Otherwise
// 128x128, except when that grid would leave CUs idle (< 256 workgroups): then 64x128 doubles the grid.
// All grid-size tests use M_plan (the rows of ONE of the plan_div stacked units): the plan, and with it the
// order of every accumulation, does not depend on how many units share the launch.
template <typename T, bool GLDS>
int dispatch_conv2(const ConvArgs& a, hipStream_t s) {
    if (a.Cout <= 32) return launch_conv<T, 256, 32, 4, 1, GLDS>(a, s);
    const int ov = a.tune & RSVLD_TUNE_TILE_MASK;
    const int st = GLDS ? ((a.tune >> RSVLD_TUNE_STAGES_SHIFT) & 7) : 2;
    const bool ksplit = !(a.tune & RSVLD_TUNE_NO_KSPLIT);
    if (a.Cout <= 64) {
        if (ov == 1) return launch_conv<T, 256, 64, 4, 1, GLDS>(a, s);
        if constexpr (GLDS) {   // measured: 3 WGs/CU x 1 tile in flight (48 KiB) beats 2 WGs/CU x 2 tiles (72 KiB)
            if (st == 3) return launch_conv<T, 128, 64, 4, 1, true, 3>(a, s);
            if (st == 4) return launch_conv<T, 128, 64, 4, 1, true, 4>(a, s);
        }
        return launch_conv<T, 128, 64, 4, 1, GLDS>(a, s);
    }
    const int64_t wg128 = (int64_t)((a.M_plan + 127) / 128) * ((a.Cout + 127) / 128);
    const int64_t wg64x128 = (int64_t)((a.M_plan + 63) / 64) * ((a.Cout + 127) / 128);
    if constexpr (GLDS) {
        // small-M linears (Stage-2 transformer blocks at 16x16 / 32x32 tokens): even 64x128 leaves most CUs idle and
        // the K loop is a latency-bound weight stream.  64x64 tiles double the grid again; 4 x 16 KiB stages keep
        // 3 tiles in flight per workgroup at 2 workgroups per CU.
        if (ov == 0 && wg64x128 < 256) {
            const int64_t wg64 = (int64_t)((a.M_plan + 63) / 64) * ((a.Cout + 63) / 64);
            if (ksplit && st == 0 && wg64 <= 256 && a.nk >= 16) return launch_conv<T, 64, 64, 2, 2, true, 4, 2>(a, s);
            return launch_conv<T, 64, 64, 2, 2, true, 4>(a, s);
        }
    }
    if (ov == 4 || (ov == 0 && wg128 < 256)) {
        if constexpr (GLDS) {
            // one 64x128 workgroup per CU at most and a long K loop: two K groups (8 waves) per tile
            if (ksplit && st == 0 && wg64x128 <= 256 && a.nk >= 16) return launch_conv<T, 64, 128, 2, 2, true, 3, 2>(a, s);
            if (st == 3 || st == 0) return launch_conv<T, 64, 128, 2, 2, true, 3>(a, s);   // 72 KiB: 2 WGs / CU
            if (st == 4) return launch_conv<T, 64, 128, 2, 2, true, 4>(a, s);
        }
        return launch_conv<T, 64, 128, 2, 2, GLDS>(a, s);
    }
    if constexpr (GLDS) {   // measured: the 96 KiB ring drops to 1 WG/CU and loses 30 % to 2 x 64 KiB double buffers
        if (st == 3) return launch_conv<T, 128, 128, 2, 2, true, 3>(a, s);
        if (st == 4) return launch_conv<T, 128, 128, 2, 2, true, 4>(a, s);
    }
    return launch_conv<T, 128, 128, 2, 2, GLDS>(a, s);
}

template <typename T>
int dispatch_conv(const ConvArgs& a, hipStream_t s) {
    return (a.tune & RSVLD_TUNE_REG_STAGING) ? dispatch_conv2<T, false>(a, s) : dispatch_conv2<T, true>(a, s);
}

}  // namespace

int rsvld_gemm256_try(const rsvld_conv_desc* d, void* stream);   // gemm.hip: large 1x1 / Linear layers

extern "C" int rsvld_conv2d_nhwc(const rsvld_conv_desc* d, void* stream) {
    if (d == nullptr || d->x == nullptr || d->w == nullptr || d->out == nullptr) return RSVLD_EINVAL;
    if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Ho <= 0 || d->Wo <= 0) return RSVLD_EINVAL;
    if (d->Cin <= 0 || d->Cin % 8 != 0 || d->Cout <= 0 || d->Cout % 8 != 0) return RSVLD_EINVAL;
    if (d->Cin2 < 0 || d->Cin2 % 8 != 0 || ((d->Cin2 > 0) != (d->x2 != nullptr))) return RSVLD_EINVAL;
    if (d->KH <= 0 || d->KW <= 0 || d->stride <= 0) return RSVLD_EINVAL;
    // RSVLD_F16W1: the weight-pair kernels (SEG = 2: fp16 in, fp32 out + fp32 residual) over ONE K segment -- their second-segment wrap of the
    // channel index never triggers when Ctot = Cseg, and the weight rows are K long
    const bool split = d->dtype == RSVLD_SPLIT, w1 = d->dtype == RSVLD_F16W1, w2 = d->dtype == RSVLD_F16W2 || w1;
    const int seg = split ? 3 : w2 ? 2 : 1;
    if (d->dtype != RSVLD_F16 && d->dtype != RSVLD_BF16 && seg == 1) return RSVLD_EINVAL;
    if (w1 && (d->out_f32 != 1 || d->KH != 1 || d->KW != 1)) return RSVLD_EINVAL;
    if (d->out_f32 < 0 || d->out_f32 > 2 || (d->out_f32 == 2 && !split)) return RSVLD_EINVAL;
    if (seg == 1 && d->out_f32 && (d->Cout > 32 || d->act == RSVLD_ACT_GEGLU || d->residual != nullptr)) return RSVLD_EUNSUPPORTED;
    if (split && d->out_f32 != 1 && d->residual != nullptr) return RSVLD_EINVAL;   // planes / fp16 out of RSVLD_SPLIT: no residual (the stream stays fp32)
    if (d->act == RSVLD_ACT_GEGLU && (d->Cout % 16 != 0 || d->residual != nullptr)) return RSVLD_EINVAL;
    if ((int64_t)d->B * d->Ho * d->Wo >= (int64_t)1 << 31) return RSVLD_EUNSUPPORTED;
    if ((int64_t)d->B * d->H * d->W >= (int64_t)1 << 31) return RSVLD_EUNSUPPORTED;
    // every output pixel must read inside the (optionally up-sampled) padded input: shape sanity
    {
        const int Hin = d->upsample ? 2 * d->H : d->H, Win = d->upsample ? 2 * d->W : d->W;
        if ((d->Ho - 1) * d->stride - d->pad_t >= Hin || (d->Wo - 1) * d->stride - d->pad_l >= Win) return RSVLD_EINVAL;
    }
    {
        const int rc = rsvld_gemm256_try(d, stream);   // the dedicated GEMM takes the shapes it is built for
        if (rc != RSVLD_EUNSUPPORTED) return rc;
    }
    ConvArgs a;
    a.x = d->x; a.x2 = d->x2; a.w = d->w; a.bias = d->bias; a.rowvec = d->rowvec;
    a.residual = d->residual; a.out = d->out;
    {   // the zero page's address is a kernel argument: taking &g_zero16 in device code reloads it through the GOT
        // inside the K loop, and an outstanding scalar load forces every LDS wait there to lgkmcnt(0)
        static const void* const zero = [] {   // one-time, thread-safe
            void* z = nullptr;
            return hipGetSymbolAddress(&z, HIP_SYMBOL(g_zero16)) == hipSuccess ? (const void*)z : (const void*)nullptr;
        }();
        if (zero == nullptr) return RSVLD_ELAUNCH;
        a.zero = zero;
    }
    a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cin2 = d->Cin2; a.Cout = d->Cout;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad_t = d->pad_t; a.pad_l = d->pad_l;
    a.Ho = d->Ho; a.Wo = d->Wo; a.upsample = d->upsample ? 1 : 0;
    a.out_f32 = d->out_f32 == 1 ? 1 : 0; a.act = d->act; a.alpha = d->alpha; a.beta = d->beta;
    a.M = d->B * d->Ho * d->Wo;
    a.M_plan = d->plan_div > 1 ? (a.M + d->plan_div - 1) / d->plan_div : a.M;
    a.tune = d->tune;
    a.HoWo = d->Ho * d->Wo;
    a.C1_8 = d->Cin / 8;
    a.C2_8 = d->Cin2 / 8;
    a.seg = seg;
    a.out_kind = d->out_f32 == 1 ? 1 : (split && d->out_f32 == 0) ? 2 : 0;
    a.Cseg8 = (d->Cin + d->Cin2) / 8;
    a.Ctot8 = (w1 ? 1 : seg) * a.Cseg8;
    a.KC = d->KH * d->KW * a.Ctot8;
    a.nk = (a.KC + 7) / 8;
    a.Cout_out = d->act == RSVLD_ACT_GEGLU ? d->Cout / 2 : d->Cout;
    a.rv_stride = d->rowvec_stride > 0 ? d->rowvec_stride : d->Cout;
    hipStream_t s = (hipStream_t)stream;
    return (d->dtype == RSVLD_F16 || w2) ? dispatch_conv<f16>(a, s) : dispatch_conv<bf16>(a, s);   // RSVLD_SPLIT runs the bf16 kernels, RSVLD_F16W2 the fp16 ones
}
