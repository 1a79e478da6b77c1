// conv_halo.hip — 3x3 / stride-1 / pad-1 convolution with an LDS-resident input halo patch and an
// optional fused GroupNorm(+SiLU) prologue, for gfx950.
//
// Why a second conv kernel: the implicit-GEMM kernel (conv_igemm.hip) gathers the im2col operand from
// L2 for every tap, i.e. each activation byte crosses the L2->LDS path 9 times; rocprofv3 showed
// ~7x the algorithmic HBM-side bytes and a kernel bound by bytes-in-flight / latency.
// Here a workgroup owns an 8 x 32 pixel tile of ONE image:
//   per 64-channel chunk:  the (8+2) x (32+2) x 64ch input patch is loaded ONCE (16-byte vectors),
//                          normalised and activated in registers  y = silu(a[b,c]*x + b[b,c])  when a
//                          GroupNorm precedes the conv (zero padding is applied AFTER the activation,
//                          as nn.Conv2d does), and written to LDS (XOR-swizzled 128-B rows);
//   per tap (9 per chunk): the [BN x 64] weight slice streams in by LDS-DMA (double buffered) while
//                          the MFMAs of the previous tap run; the pixel operand of tap (ky,kx) is the
//                          SAME patch read at a shifted row offset -- no data movement at all.
// L2->LDS bytes per 256 pixels x 64 channels: 43.5 KiB patch + 9 x BN x 128 B weights, vs
// 9 x (256+BN) x 128 B for the gather.  MFMA work per barrier doubles (32 MFMAs per wave per tap).
#include "rsvld_common.h"
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace {

struct HaloArgs {
    const void* x;
    const void* x2;
    const void* w;
    const float* bias;
    const float* rowvec;
    const void* residual;
    void* out;
    const float* ab;     // [B][Cin+Cin2][2] = (scale, shift) of the fused GroupNorm, or nullptr
    float* stats;        // [B][tiles_y*tiles_x][Cout][2] per-tile per-channel (sum, sumsq) of the OUTPUT, or nullptr
    int B, H, W, Cin, Cin2, Cout;   // H, W: OUTPUT map (= input map, or 2x the input when ush = 1)
    int Hs, Ws, ush;                // source map and the nearest-x2 shift
    int out_f32, act, norm_silu;
    float alpha, beta;
    int Ctot, nchunks;   // channels, 64-channel chunks
    int tiles_x, tiles_y;   // workgroup tiles (tiles_y counts rows of the kernel's own tile height)
    int tiles_y8;           // rows of the 8x32-pixel grid of the statistics partials
    int rv_stride, Cout_out;
    int B_plan, tune;       // batch rows the launch plan is made for (B / plan_div); RSVLD_TUNE_*
    // dtype RSVLD_SPLIT (round 4; see gemm.hip): x / x2 are bf16 planes [.., lo(C) | hi(C)], the weights the per-tap triple
    // [W_hi | W_lo | W_hi]; Ctot = 3 Cseg logical channels whose third segment re-reads the hi planes; fp32 residual, fp32 / planes out
    // dtype RSVLD_F16W2 (round 5): x / x2 fp16, the weights the per-tap pair [W_lo | W_hi]; Ctot = 2 Cseg logical channels whose second
    // segment re-reads the activation (and re-applies the fused GroupNorm: `ab` holds Cseg rows per image); residual and output fp32
    // (out_f32) or fp16
    // dtype RSVLD_F16Q8 (round 6, seg = 5): x / x2 rows [fp16(x) (C) | C / 32 e4m3 blocks], weight rows per tap likewise; Ctot = 2 Cseg logical
    // channels: a body = the fp16 chunk of 32 channels (two fp16 MFMAs per tap and tile) + their e4m3 block (ONE scaled MFMA over K = 64:
    // lane half 0 x_lo w_hi, lane half 1 x_hi w_lo); fp32 out, fp32 residual
    int seg, Cseg;
};

constexpr int TH = 8, TW = 32, PW = TW + 2, PROWS = (TH + 2) * PW;   // 340 patch pixels
constexpr int PATCH_BYTES = PROWS * 128;
constexpr int PLOADS = (PROWS * 8 + 255) / 256;                        // 11 16-byte pieces per thread

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ int swz_off(int row, int c) { return row * 128 + ((c ^ ((row >> 1) & 7)) << 4); }
// Patch image: pixel (py, px) of the (TH+2) x PW patch owns the 128-B row py*PW + px; its 16-B channel chunk c sits
// at slot c ^ ((px>>1)&7).  The swizzle depends on the patch COLUMN only (PW is even, so a row of the patch starts
// on a 256-B bank row), which keeps every ds_read_b128 lane group on 16 distinct slots for any tap shift AND makes the
// per-lane part of a tap's read address a function of (kx, k-step) alone: 12 precomputed registers, the rest is an
// immediate offset -- no address arithmetic inside the tap loop.
__device__ __forceinline__ int patch_off(int py, int px, int c) { return (py * PW + px) * 128 + ((c ^ ((px >> 1) & 7)) << 4); }

// logical channel ch0 (start of a 32- or 64-channel chunk) -> source tensor (0: x, 1: x2), channel inside a pixel's row, elements per
// pixel.  Split: segment 0 = lo planes, segments 1, 2 = hi planes; a pixel's row holds lo | hi.
template <int SEG>
__device__ __forceinline__ void halo_src_of(const HaloArgs& p, int ch0, int& which, int& Cs, int& coff) {
    if (SEG != 3 && SEG != 5) {   // (SEG 5: the fp16 part sits where the lo plane does, the e4m3 blocks where the hi plane does)
        if (SEG == 2 && ch0 >= p.Cseg) ch0 -= p.Cseg;   // the pair form reads the one activation twice
        if (ch0 < p.Cin) { which = 0; Cs = p.Cin; coff = ch0; } else { which = 1; Cs = p.Cin2; coff = ch0 - p.Cin; }
        return;
    }
    int hi = 0;
    if (ch0 >= p.Cseg) { ch0 -= p.Cseg; hi = 1; }
    if (ch0 >= p.Cseg) ch0 -= p.Cseg;
    if (ch0 < p.Cin) { which = 0; Cs = 2 * p.Cin; coff = ch0 + hi * p.Cin; } else { which = 1; Cs = 2 * p.Cin2; coff = ch0 - p.Cin + hi * p.Cin2; }
}

// ---- epilogue shared by the halo kernels: accumulators -> LDS (fp32) -> bias / row vector / SiLU / residual ->
// 16-byte NHWC stores, plus the per-channel (sum, sumsq) partials of every 8x32-pixel sub-tile for the next GroupNorm.
// NW waves own 64*NW pixels (8 rows of 32 per 4 waves); passes of EPI_ROWS pixels; the staging tile aliases the (dead)
// operand buffers.
template <typename T, int BN, int TM, int TN, int NW = 4, int SEG = 1>
__device__ __forceinline__ void halo_epilogue(const HaloArgs& p, char* smem, f32x16 (&acc)[TN][TM], int tid, int wm, int wn,
                                              int l31, int lh, int x0, int y0, int n0, int img, int tx, int ty) {
    constexpr bool SPLIT = SEG == 3;
    // the residual has the output's type: fp32 beside an fp32 / planes output of the multi-segment forms, 16-bit otherwise
    const bool res32 = SEG > 1 && (SPLIT || p.out_f32);
    constexpr int NT = 64 * NW, PIX = 64 * NW;
    constexpr int EPI_ROWS = BN > 64 ? 128 : 256;
    constexpr int EPI_PASSES = PIX / EPI_ROWS;
    constexpr int SUB = 256 / EPI_ROWS;        // passes per 8x32 sub-tile
    constexpr int CT_STRIDE = BN + 4;
    float* Ct = (float*)smem;
    constexpr int CPR = BN / 8, RPP = NT / CPR;
    const int cc = tid % CPR, rr = tid / CPR;
    const int n = n0 + cc * 8;
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = (p.bias != nullptr && n + e < p.Cout) ? p.bias[n + e] : 0.f;
    // statistics of the stored tensor for the NEXT GroupNorm: this thread's 8 channels, summed over its pixels
    float st_s[8], st_q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { st_s[e] = 0.f; st_q[e] = 0.f; }
    constexpr int RPT = EPI_ROWS / RPP;        // rows each thread stores per pass
#pragma unroll
    for (int pass = 0; pass < EPI_PASSES; ++pass) {
        // The residual pieces of this pass are requested BEFORE the staging writes and their barrier, so that the HBM round trip
        // runs under them (one row at a time every store waited for its own residual load; same change as in gemm.hip).
        u32x4 rres[RPT];
        if (p.residual != nullptr && n < p.Cout && !res32) {
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                const int prow = pass * EPI_ROWS + rr + j * RPP;
                const int y = y0 + (prow >> 5), x = x0 + (prow & 31);
                rres[j] = (y < p.H && x < p.W) ? *(const u32x4*)((const T*)p.residual + (((int64_t)img * p.H + y) * p.W + x) * p.Cout_out + n)
                                               : u32x4{0u, 0u, 0u, 0u};
            }
        }
        if (pass > 0) __syncthreads();
        if ((wm * TM * 32) / EPI_ROWS == pass) {
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int row = (wm * TM + mi) * 32 + l31 - pass * EPI_ROWS;
                        const int col = wn * (TN * 32) + ni * 32 + 8 * g + 4 * lh;
                        f32x4 v = {acc[ni][mi][4 * g], acc[ni][mi][4 * g + 1], acc[ni][mi][4 * g + 2], acc[ni][mi][4 * g + 3]};
                        *(f32x4*)(Ct + row * CT_STRIDE + col) = v;
                    }
        }
        __syncthreads();
        if (n < p.Cout) {
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                const int row = rr + j * RPP;
                const int prow = pass * EPI_ROWS + row;
                const int y = y0 + (prow >> 5), x = x0 + (prow & 31);
                if (y >= p.H || x >= p.W) continue;
                const int64_t m = ((int64_t)img * p.H + y) * p.W + x;
                const f32x4 v0 = *(const f32x4*)(Ct + row * CT_STRIDE + cc * 8);
                const f32x4 v1 = *(const f32x4*)(Ct + row * CT_STRIDE + cc * 8 + 4);
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += bv[e];
                if (p.rowvec != nullptr) {
                    const float* rv = p.rowvec + (int64_t)img * p.rv_stride + n;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += rv[e];
                }
                if (p.act == RSVLD_ACT_SILU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
                if (p.residual != nullptr) {
                    float rf[8];
                    if (res32) {   // fp32 residual
                        const float* r = (const float*)p.residual + m * p.Cout_out + n;
                        const f32x4 r0 = *(const f32x4*)r, r1 = *(const f32x4*)(r + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { rf[e] = r0[e]; rf[4 + e] = r1[e]; }
                    } else {
                        unpack8<T>(rres[j], rf);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += p.beta * rf[e];
                }
                if (p.stats != nullptr) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { st_s[e] += v[e]; st_q[e] += v[e] * v[e]; }
                }
                if (SPLIT && !p.out_f32) {   // planes: lo | hi per pixel row
                    float lo[8];
                    typename Mfma<T>::v8 hv;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { hv[e] = (T)v[e]; lo[e] = v[e] - (float)hv[e]; }
                    T* ob = (T*)p.out + m * (2 * p.Cout_out) + n;
                    *(u32x4*)ob = pack8<T>(lo);
                    *(u32x4*)(ob + p.Cout_out) = __builtin_bit_cast(u32x4, hv);
                } else if (p.out_f32) {
                    float* o = (float*)p.out + m * p.Cout_out + n;
                    *(f32x4*)o = (f32x4){v[0], v[1], v[2], v[3]};
                    *(f32x4*)(o + 4) = (f32x4){v[4], v[5], v[6], v[7]};
                } else {
                    *(u32x4*)((T*)p.out + m * p.Cout_out + n) = pack8<T>(v);
                }
            }
        }
        if (p.stats != nullptr && (pass % SUB) == SUB - 1) {   // workgroup-uniform: an 8x32 sub-tile is complete
            __syncthreads();        // Ct is dead: reuse it as [RPP][BN][2]
            float* red = (float*)smem;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[((rr * BN) + cc * 8 + e) * 2] = st_s[e];
                red[((rr * BN) + cc * 8 + e) * 2 + 1] = st_q[e];
                st_s[e] = 0.f;
                st_q[e] = 0.f;
            }
            __syncthreads();
            const int ty8 = ty * (PIX / 256) + pass / SUB;   // row of the 8x32 partial grid
            if (tid < BN && n0 + tid < p.Cout && ty8 < p.tiles_y8) {
                float a = 0.f, q = 0.f;
                for (int r = 0; r < RPP; ++r) { a += red[(r * BN + tid) * 2]; q += red[(r * BN + tid) * 2 + 1]; }
                const int64_t tile = (int64_t)ty8 * p.tiles_x + tx;
                float* o = p.stats + (((int64_t)img * p.tiles_x * p.tiles_y8 + tile) * p.Cout + n0 + tid) * 2;
                o[0] = a;
                o[1] = q;
            }
        }
    }
}

// TPS = taps of weights staged per pipeline step (per barrier): 1 for BN = 128, 2 for BN = 64, so that every
// step carries a 16 KiB weight slice and 32 MFMAs per wave
template <typename T, int BN, int WAVES_M, int TPS, int SEG = 1>
__global__ __launch_bounds__(256) void conv_halo_kernel(HaloArgs p) {
    constexpr bool SPLIT = SEG == 3;
    constexpr int WAVES_N = 4 / WAVES_M;
    constexpr int SPC = (9 + TPS - 1) / TPS;       // steps per 64-channel chunk
    constexpr int TM = TH / WAVES_M;               // 32-pixel rows per wave
    constexpr int TN = BN / (32 * WAVES_N);
    constexpr int W_BYTES = TPS * BN * 128;        // one step's weight stage
    constexpr int W_LOADS = BN / 32;
    typedef typename Mfma<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* patch = smem;
    char* wbuf = smem + PATCH_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l31 = lane & 31, lh = lane >> 5;

    // XCD-aware tile order (same scheme as conv_igemm): linear id -> contiguous run per XCD, x fastest
    int tx, ty, img, tile_n;
    {
        const int nwg = gridDim.x;
        const int lid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = lid & 7, slot = lid >> 3;
        int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
        tx = t % p.tiles_x; t /= p.tiles_x;
        ty = t % p.tiles_y; t /= p.tiles_y;
        img = t % p.B;
        tile_n = t / p.B;
    }
    const int x0 = tx * TW, y0 = ty * TH, n0 = tile_n * BN;

    // ---- patch roles: piece i of this thread = patch pixel (tid>>3) + 32 i, chunk c = tid & 7
    const int c = tid & 7;
    int poff[PLOADS];     // pixel offset inside the image, -1 = outside (zero padding) or beyond the patch
#pragma unroll
    for (int i = 0; i < PLOADS; ++i) {
        const int pp = (tid >> 3) + 32 * i;
        const int py = pp / PW, px = pp - py * PW;
        const int y = y0 - 1 + py, x = x0 - 1 + px;
        // with ush = 1 the conv runs on the nearest-x2 up-sampled map: output-grid pixel (y, x) reads source (y>>1, x>>1)
        poff[i] = (pp < PROWS && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) ? (y >> p.ush) * p.Ws + (x >> p.ush) : -1;
    }
    const T* __restrict__ X1 = (const T*)p.x + (int64_t)img * p.Hs * p.Ws * (SPLIT ? 2 * p.Cin : p.Cin);
    const T* __restrict__ X2 = p.x2 ? (const T*)p.x2 + (int64_t)img * p.Hs * p.Ws * (SPLIT ? 2 * p.Cin2 : p.Cin2) : nullptr;
    const T* __restrict__ Wp = (const T*)p.w;
    const int64_t Kel = (int64_t)9 * p.Ctot;

    // PAIRED (round 5, the weight-pair form): the two 64-channel K chunks of one channel block are ADJACENT -- chunk 2 c = channels 64 c
    // against W_lo, chunk 2 c + 1 = the same channels against W_hi -- so the second runs on the patch its partner left in LDS: no global
    // load, no normalisation, no LDS write.  (The per-tap weight row is [W_lo(C) | W_hi(C)].)
    constexpr bool PAIRED = SEG == 2;
    auto chan0 = [&](int kc) { return PAIRED ? (kc >> 1) * 64 : kc * 64; };                                   // first logical channel of the chunk's patch
    auto wchan0 = [&](int kc) { return PAIRED ? (kc & 1) * p.Cseg + (kc >> 1) * 64 : kc * 64; };              // ... of its weight slice
    u32x4 rp[PLOADS];
    auto load_patch = [&](int kc) {
        int which, Cs, coff;
        halo_src_of<PAIRED ? 1 : SEG>(p, chan0(kc), which, Cs, coff);
        const T* src = which ? X2 : X1;
        coff += c * 8;
#pragma unroll
        for (int i = 0; i < PLOADS; ++i) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (poff[i] >= 0) v = *(const u32x4*)(src + (int64_t)poff[i] * Cs + coff);
            rp[i] = v;
        }
    };
    auto store_patch = [&](int kc) {
        float sa[8], sb[8];
        if (p.ab != nullptr) {
            const float* ab = p.ab + ((int64_t)img * p.Cseg + chan0(kc) + c * 8) * 2;   // (PAIRED: SEG == 2 is always paired here)
#pragma unroll
            for (int e = 0; e < 8; ++e) { sa[e] = ab[2 * e]; sb[e] = ab[2 * e + 1]; }
        }
#pragma unroll
        for (int i = 0; i < PLOADS; ++i) {
            const int pp = (tid >> 3) + 32 * i;
            if (pp >= PROWS) continue;
            const int py = pp / PW, px = pp - py * PW;
            u32x4 v = rp[i];
            if (p.ab != nullptr && poff[i] >= 0) {
                float f[8];
                unpack8<T>(v, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float t = f[e] * sa[e] + sb[e];
                    f[e] = p.norm_silu ? silu_f(t) : t;
                }
                v = pack8<T>(f);
            }
            *(u32x4*)(patch + patch_off(py, px, c)) = v;
        }
    };
    // weights of (chunk kc, tap): rows n0.. , k = tap*Ctot + kc*64 ..+64 ; lane-linear LDS image, source-side swizzle
    const int wr0 = tid >> 3;
    const int wc = (tid & 7) ^ ((wr0 >> 1) & 7);
    auto dma_w = [&](int kc, int step, int buf) {   // taps step*TPS .. of chunk kc
#pragma unroll
        for (int j = 0; j < TPS; ++j) {
            const int tap = step * TPS + j;
            if (tap >= 9) break;
            char* dst = wbuf + buf * W_BYTES + j * (BN * 128);
            const int64_t koff = (int64_t)tap * p.Ctot + wchan0(kc) + wc * 8;
#pragma unroll
            for (int i = 0; i < W_LOADS; ++i) {
                // rows past Cout re-read the last row: their accumulators are never stored
                const int n = min(n0 + wr0 + 32 * i, p.Cout - 1);
                __builtin_amdgcn_global_load_lds((gptr_t)(Wp + (int64_t)n * Kel + koff), (lptr_t)(dst + (wave * 8 + 32 * i) * 128), 16, 0, 0);
            }
        }
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

    // Loop structure.  hipcc drains vmcnt(0) in front of the first ds_read of a loop body whenever ordinary
    // (VGPR-destination) global loads are live across iterations next to LDS-DMA (cdna_hip_programming.md §5
    // "Three .s-level traps" (b)); measured here as a full weight-DMA round trip exposed on EVERY tap.  So the tap
    // loop contains LDS-DMA only; the register-staged patch phase (load -> normalise -> ds_write) runs between
    // chunks, while the first tap's weight slice of that chunk is already in flight.
    // per-lane part of the operand addresses: patch [kx][k-step], weights [k-step]
    int fb_off[3][4], fa_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) fb_off[kx][ks] = patch_off(wm * TM, l31 + kx, 2 * ks + lh);
        fa_off[ks] = swz_off(wn * (TN * 32) + l31, 2 * ks + lh);
    }
    const int nsteps = p.nchunks * SPC;
    int s = 0;
    dma_w(0, 0, 0);
    for (int kc = 0; kc < p.nchunks; ++kc) {
        if (!PAIRED || !(kc & 1)) {
            load_patch(kc);
            store_patch(kc);
        }
        __syncthreads();   // patch visible; weights of (kc, step 0) landed (vmcnt(0) + barrier)
#pragma unroll
        for (int st = 0; st < SPC; ++st, ++s) {   // unrolled: the tap of every read is a compile-time constant
            if (s + 1 < nsteps) {   // next step's weights stream in behind this step's MFMAs
                if (st == SPC - 1) dma_w(kc + 1, 0, (s + 1) & 1);
                else dma_w(kc, st + 1, (s + 1) & 1);
            }
            const char* w_st = wbuf + (s & 1) * W_BYTES;
#pragma unroll
            for (int j = 0; j < TPS; ++j) {
                const int tap = st * TPS + j;
                if (tap >= 9) break;
                const int ky = tap / 3, kx = tap - ky * 3;
                const char* w_s = w_st + j * (BN * 128);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    v8 fa[TN], fb[TM];
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) fa[ni] = *(const v8*)(w_s + fa_off[ks] + ni * (32 * 128));
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi)   // shifted window of the same patch
                        fb[mi] = *(const v8*)(patch + fb_off[kx][ks] + (mi + ky) * (PW * 128));
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                        for (int mi = 0; mi < TM; ++mi) acc[ni][mi] = Mfma<T>::mma(fa[ni], fb[mi], acc[ni][mi]);
                }
            }
            __syncthreads();   // next weights landed; everyone done with this step's weights (last step: with the patch)
        }
    }

    halo_epilogue<T, BN, TM, TN, 4, SEG>(p, smem, acc, tid, wm, wn, l31, lh, x0, y0, n0, img, tx, ty);
}

// ---------------------------------------------------------------------------------------------------------------
// conv_halo32: the same halo-patch convolution with 32-channel K chunks and the patch DOUBLE-BUFFERED in LDS.
//
// In conv_halo_kernel the patch phase of every 64-channel chunk (global load -> GroupNorm+SiLU -> ds_write) sits between
// two tap loops with nothing to hide it but the other workgroup of the CU (measured: the same GEMM shape runs at
// 660-800 TFLOP/s with the fused norm and 980-1120 without).  Holding the next 64-channel patch in registers costs 44
// VGPRs on top of 128 accumulators -- over the 2-waves-per-SIMD budget.  With 32-channel chunks a patch buffer is
// 21.25 KiB, two of them fit where one 64-channel patch was, the prefetch needs 24 VGPRs, and the normalised pieces
// are written straight into the idle buffer a few at a time BESIDE the MFMAs of the same wave:
//
//   body = two chunks A (buffer 0), B (buffer 1) = 18 taps = 9 steps of 2 taps (32 MFMAs per wave per barrier)
//   step 0 : ...........................  end: B's loads (issued in step 8 of the previous body) are waited for
//   step 1-3: normalise + write B, 2 pieces per step -> buffer 1 (idle since step 8 of the previous body)
//   step 4 : taps A8, B0; request A' = first chunk of the next body (asm loads, left in flight over the barrier)
//   step 5 : ...........................  end: A' landed
//   step 6-8: normalise + write A' -> buffer 0 (idle since step 4);  step 8 then requests B'
//
// The tap loop has no load hipcc tracks: patch loads are inline asm (cdna_hip_programming.md §5.7 form ii), weights
// and the (scale, shift) rows are LDS-DMA, waits are counted by hand (vmcnt(6) leaves exactly the 6 patch loads,
// issued after the step's DMAs, in flight) and the barrier is raw.  NORM / NEXT are compile-time so that a step is
// one basic block and the scheduler can place the VALU between the MFMAs.
// ---------------------------------------------------------------------------------------------------------------
#ifndef HALO_ABL
#define HALO_ABL 0   // diagnostic builds (timing only, results wrong; every buffer is filled once so that operands stay real data):
                     // 1 no weight DMA in the loop, 2 no patch prefetch/normalise in the loop, 4 half the fragment reads
#endif
constexpr int AB32_BYTES = 1024;        // one LDS-DMA wave-instruction: 512 B of (scale, shift) + 512 B duplicate

// pixel (py, px) owns a 64-B row; 16-B slot s sits at s ^ ((px>>2)&3): conflict-free ds_read_b128 for every tap shift
// (brute-forced over the lane groups of MI355X_MICROARCH.md), ky stays an immediate offset
__device__ __forceinline__ int p32_off(int py, int px, int slot) { return (py * PW + px) * 64 + ((slot ^ ((px >> 2) & 3)) << 4); }
__device__ __forceinline__ int w32_off(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }

// compile-time loop: f(std::integral_constant<int, i>) for i = 0..N-1
template <typename F, int... I> __device__ __forceinline__ void halo_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void halo_static_for(F&& f) {
    halo_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <int N> __device__ __forceinline__ void halo_wait_barrier() {
    // DMA pieces older than the N youngest vector-memory operations have landed, this wave's LDS writes are done
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// NORM: 0 none, 1 scale/shift, 2 scale/shift + SiLU.  NW = 4: 8x32-pixel tile, two workgroups per CU.  NW = 8: 16x32-pixel
// tile, one workgroup per CU (same 8 waves per CU): the per-tap weight slice is staged once per CU instead of twice and
// the halo overhead of the patch drops (612 instead of 680 patch pixels per 512 outputs).  Ablation builds
// (tools/ablate_halo.sh) put the issue cost of the weight LDS-DMAs at 25 % of the 4-wave kernel's time and the patch
// pipeline at 22-36 %, the fragment reads at ~1 %: per-wave DMA / staging instructions are what to cut.
// PAIRED bodies (round 5, the multi-segment forms; H32_PAIRED = 0 restores the linear K' order): a body is ONE 32-channel chunk of the
// activation under the two weight column blocks that multiply it -- step st = tap st with the [BN x 32] slices of both -- instead of two
// consecutive chunks of the concatenated K'.  Weight pairs (K' = [X | X] against [W_lo | W_hi]): every body.  Split (K' = [X_lo | X_hi | X_hi]
// against [W_hi | W_lo | W_hi]): the hi planes, after the lo planes have gone through in the linear order.  The patch of such a chunk is
// loaded, normalised and written to LDS once instead of twice (the patch pipeline is 13-22 % of the weight-pair form's time by the
// -DHALO_ABL=2 builds, tools/bench_halo_seg.py), its fragments are read once per tap for both blocks, and the two patch buffers alternate
// by body.  Same products, another summation order (per tap instead of per segment).
#ifndef H32_PAIRED
#define H32_PAIRED 1
#endif
template <typename T, int BN, int NORM, int NW, int SEG = 1>
__global__ __launch_bounds__(64 * NW, 2) void conv_halo32_kernel(HaloArgs p) {
    constexpr bool SPLIT = SEG == 3;
    constexpr bool HQ = SEG == 5;            // RSVLD_F16Q8: chunk A of a body = fp16 channels, chunk B = their e4m3 cross-term block
    constexpr bool TWO_PART = SPLIT || HQ;   // a pixel's row holds two parts of C 16-bit elements each
    constexpr int NT = 64 * NW;
    constexpr int WAVES_N = 2, WAVES_M = NW / 2;
    constexpr int TM = 4;                            // 32-pixel rows per wave
    constexpr int THT = TM * WAVES_M;                // tile rows: 8 or 16
    constexpr int TN = BN / (32 * WAVES_N);
    constexpr int PR = (THT + 2) * PW;               // patch pixels
    constexpr int PB = PR * 64;                      // one 32-channel patch buffer
    constexpr int PL = (PR * 4 + NT - 1) / NT;       // 16-byte pieces per thread: 6 (NW = 4) or 5 (NW = 8)
    constexpr int TAP_BYTES = BN * 64;               // [BN x 32 ch] weight slice of one tap
    constexpr int W_BYTES = 2 * TAP_BYTES;           // a step stages two taps
    constexpr int W_LOADS = BN / (16 * NW);          // LDS-DMA instructions per thread per tap
    // weight stages: the 4-wave kernel has LDS for two (the slice of step s+1 is requested in step s and must land within
    // it); the 8-wave kernel has room for three and requests two steps ahead -- the ablation builds charge most of the
    // "weight DMA" cost to exactly that wait
    constexpr int WST = NW == 8 ? 3 : 2, AHEAD = WST - 1;
    static_assert(PL == 5 || PL == 6, "wait_patch names PL register quads");
    static_assert(BN % (16 * NW) == 0, "whole wave-instructions per tap");
    typedef typename Mfma<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* patch = smem;                      // [2][PB]
    char* wbuf = smem + 2 * PB;              // [WST][W_BYTES]
    char* abuf = wbuf + WST * W_BYTES;       // [2][AB32_BYTES]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l31 = lane & 31, lh = lane >> 5;

    int tx, ty, img, tile_n;
    {
        const int nwg = gridDim.x;
        const int lid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = lid & 7, slot = lid >> 3;
        int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
        tx = t % p.tiles_x; t /= p.tiles_x;
        ty = t % p.tiles_y; t /= p.tiles_y;
        img = t % p.B;
        tile_n = t / p.B;
    }
    const int x0 = tx * TW, y0 = ty * THT, n0 = tile_n * BN;

    // ---- patch roles: piece i of this thread = patch pixel (tid>>2) + 16 NW i, 16-byte slot c4 = tid & 3
    const int c4 = tid & 3;
    int poff[PL];   // source pixel offset, -1 = zero padding / past the patch
    int pdst[PL];   // LDS byte offset inside a patch buffer; pieces past the patch write a dead LDS word
#pragma unroll
    for (int i = 0; i < PL; ++i) {
        const int pp = (tid >> 2) + 16 * NW * i;
        const int py = pp / PW, px = pp - py * PW;
        const int y = y0 - 1 + py, x = x0 - 1 + px;
        poff[i] = (pp < PR && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) ? (y >> p.ush) * p.Ws + (x >> p.ush) : -1;
        // dead target: the duplicate half of (scale, shift) buffer 0, never read
        pdst[i] = pp < PR ? p32_off(py, px, c4) : (int)(abuf - patch) + 512 + (tid & 31) * 16;
    }
    const T* __restrict__ X1 = (const T*)p.x + (int64_t)img * p.Hs * p.Ws * (TWO_PART ? 2 * p.Cin : p.Cin);
    const T* __restrict__ X2 = p.x2 ? (const T*)p.x2 + (int64_t)img * p.Hs * p.Ws * (TWO_PART ? 2 * p.Cin2 : p.Cin2) : nullptr;
    const T* __restrict__ Wp = (const T*)p.w;
    const int64_t Kel = (int64_t)9 * p.Ctot;

    u32x4 rp[PL];
    auto issue_patch = [&](int ch0, u32x4 (&r)[PL]) {   // ch0: first K' channel of the 32-channel chunk
        int which, Cs, coff;
        halo_src_of<SEG>(p, ch0, which, Cs, coff);
        const T* src = which ? X2 : X1;
        coff += c4 * 8;
#pragma unroll
        for (int i = 0; i < PL; ++i) {
            const T* ptr = src + (int64_t)(poff[i] < 0 ? 0 : poff[i]) * Cs + coff;   // padding reads a valid dummy, zeroed later
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[i]) : "v"(ptr) : "memory");
        }
    };
    // names every destination: no consumer is scheduled above it; the weight DMAs issued after the patch loads (the
    // N youngest operations) may stay in flight
    auto wait_patch = [&](u32x4 (&r)[PL], auto n_c) {
        constexpr int N = decltype(n_c)::value;
        if constexpr (PL == 6)
            asm volatile("s_waitcnt vmcnt(%6)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[PL - 1]) : "n"(N) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(%5)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]) : "n"(N) : "memory");
    };
    // normalise piece i and write it into patch buffer P; ab16 = (scale, shift) of this thread's 8 channels.  Branch-free.
    auto norm_write = [&](u32x4 v, int i, const float (&ab16)[16], char* P) {
        if (NORM != 0) {
            float f[8];
            unpack8<T>(v, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = f[e] * ab16[2 * e] + ab16[2 * e + 1];
                f[e] = NORM == 2 ? silu_f(t) : t;
            }
            v = pack8<T>(f);
        }
        const bool inside = poff[i] >= 0;   // zero padding is applied AFTER the activation, as nn.Conv2d does
        v = (u32x4){inside ? v[0] : 0u, inside ? v[1] : 0u, inside ? v[2] : 0u, inside ? v[3] : 0u};
        *(u32x4*)(P + pdst[i]) = v;
    };
    auto lds_ab = [&](const char* row, float (&ab16)[16]) {
        if (NORM != 0) {
            const f32x4* a4 = (const f32x4*)row;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = a4[q];
                ab16[4 * q] = v[0]; ab16[4 * q + 1] = v[1]; ab16[4 * q + 2] = v[2]; ab16[4 * q + 3] = v[3];
            }
        }
    };
    // weights of flat tap g (0..17) of body b: chunk 2b + (g >= 9), tap g % 9; lane-linear LDS image, source-side swizzle.
    // Address = uniform base (SGPR pair: tap, chunk) + 32-bit per-lane offset (row, swizzled slot): W_LOADS VGPRs in all --
    // with 64-bit per-lane pointers hipcc hoists one pointer per (tap, row group) out of the loop and spills them.
    const int wr0 = tid >> 2;
    const int wsl = (tid & 3) ^ ((wr0 >> 2) & 3);
    uint32_t wvoff[W_LOADS];
#pragma unroll
    for (int i = 0; i < W_LOADS; ++i)   // rows past Cout re-read the last row; their accumulators are never stored
        wvoff[i] = (uint32_t)(((int64_t)min(n0 + wr0 + 16 * NW * i, p.Cout - 1) * Kel + wsl * 8) * (int64_t)sizeof(T));
    // Body kinds (compile-time).  KIND 0: two consecutive 32-channel chunks of K' (chunk A -> patch buffer 0, B -> buffer 1; taps A0..A8, B0..B8,
    // two per step).  KIND 1 (PAIRED): ONE chunk of the activation under two weight column blocks Cseg apart -- step st = tap st under both --
    // patch buffers alternating by body.  idx = the body's index within its kind.  xch = K' channel of a body's chunk (for halo_src_of),
    // wcol = its weight column.  PAIR_X0 / PAIR_W0: where the paired chunks start (the split form pairs its hi-plane segments 1 and 2).
    const int PAIR_X0 = SEG == 3 ? p.Cseg : 0;
    auto xch = [&](auto kind_c, int idx, int which) {
        return decltype(kind_c)::value ? PAIR_X0 + idx * 32 : HQ ? (which ? p.Cseg : 0) + idx * 32 : (2 * idx + which) * 32;
    };
    auto dma_w = [&](auto kind_c, int idx, int st, int buf) {
        constexpr int KIND = decltype(kind_c)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int g = 2 * st + j;
            const int tap = KIND ? st : (g >= 9 ? g - 9 : g);
            const int col = KIND ? PAIR_X0 + j * p.Cseg + idx * 32 : HQ ? (g >= 9 ? p.Cseg : 0) + idx * 32 : (2 * idx + (g >= 9 ? 1 : 0)) * 32;
            char* dst = wbuf + buf * W_BYTES + j * TAP_BYTES;
            const char* base = (const char*)(Wp + (int64_t)tap * p.Ctot + col);   // wave-uniform
#pragma unroll
            for (int i = 0; i < W_LOADS; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t)(base + wvoff[i]), (lptr_t)(dst + (wave * 16 + 16 * NW * i) * 64), 16, 0, 0);
        }
    };
    auto dma_ab = [&](auto kind_c, int idx, int buf) {   // (scale, shift) of the body's 64 (KIND 1: 32, written twice) channels: 512 B, lanes 32..63 duplicate it
        constexpr int KIND = decltype(kind_c)::value;
        if (NORM != 0) {   // every wave issues the same piece (identical bytes): no branch in the step, uniform vmcnt
            const int bch = KIND ? idx * 32 : idx * 64 - ((SEG == 2 && idx * 64 >= p.Cseg) ? p.Cseg : 0);   // the channels the body holds (wave-uniform)
            const float* src = p.ab + ((int64_t)img * p.Cseg + bch) * 2 + (lane & (KIND ? 15 : 31)) * 4;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(abuf + buf * AB32_BYTES), 16, 0, 0);
        }
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

    // E8M0 scale bytes of the scaled MFMA, per lane half: (w 2^SW_HI)(x_lo 2^SX_LO) in half 0, (w_lo 2^SW_LO)(x 2^SX_HI) in half 1
    const int q_scale_w = 127 - (lh ? RSVLD_HQ8_SW_LO : RSVLD_HQ8_SW_HI), q_scale_x = 127 - (lh ? RSVLD_HQ8_SX_HI : RSVLD_HQ8_SX_LO);
    int fb_off[3][2], fa_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) fb_off[kx][ks] = p32_off(wm * TM, l31 + kx, 2 * ks + lh);
        fa_off[ks] = w32_off(wn * (TN * 32) + l31, 2 * ks + lh);
    }

    // ---- the body sequence: U bodies of KIND 0, then Q of KIND 1.  Plain GEMM-K order: all KIND 0.  Weight pairs: all KIND 1.  Split: the lo
    // planes (segment 0, against W_hi) as KIND 0, then every hi-plane chunk ONCE under W_lo and W_hi (segments 1 and 2) as KIND 1.
    typedef std::integral_constant<int, 0> K0;
    typedef std::integral_constant<int, 1> K1;
    constexpr bool ALL_PAIRED = SEG == 2 && H32_PAIRED, PHASED = SEG == 3 && H32_PAIRED;
    const int nb = p.nchunks;   // bodies = 64 channels of K' each
    const int U = ALL_PAIRED ? 0 : PHASED ? p.Cseg / 64 : nb, Q = nb - U;
    typedef std::integral_constant<int, ALL_PAIRED ? 1 : 0> KF;   // the first body's kind

    // ---- prologue: chunk 0 is the one exposed load -> normalise -> write; chunk 1 (KIND 0) is requested with it
#pragma unroll
    for (int a = 0; a < ((HALO_ABL & 1) ? WST : AHEAD); ++a) dma_w(KF{}, 0, a, a);   // steps 0 .. AHEAD-1 of body 0 (every body has 9 steps); ablation 1: every stage once, so that the operands stay real data
    dma_ab(KF{}, 0, 0);
    {
        u32x4 ra[PL];
        issue_patch(xch(KF{}, 0, 0), ra);
        if (!KF::value) issue_patch(xch(K0{}, 0, 1), rp);
        else {
#pragma unroll
            for (int i = 0; i < PL; ++i) rp[i] = (u32x4){0u, 0u, 0u, 0u};
        }
        float ab16[16];
        if (NORM != 0) {
            const float* ab = p.ab + ((int64_t)img * p.Cseg + c4 * 8) * 2;
#pragma unroll
            for (int e = 0; e < 16; ++e) ab16[e] = ab[e];
        }
        if constexpr (PL == 6)
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[PL - 1]), "+v"(rp[0]), "+v"(rp[1]),
                           "+v"(rp[2]), "+v"(rp[3]), "+v"(rp[4]), "+v"(rp[PL - 1])
                         :
                         : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(rp[0]), "+v"(rp[1]), "+v"(rp[2]),
                           "+v"(rp[3]), "+v"(rp[4])
                         :
                         : "memory");
#pragma unroll
        for (int i = 0; i < PL; ++i) norm_write(ra[i], i, ab16, patch);
        if (HALO_ABL & 2) {   // ablation 2: buffer 1 is filled once (real data), the loop never refreshes the patch
#pragma unroll
            for (int i = 0; i < PL; ++i) norm_write(rp[i], i, ab16, patch + PB);
        }
    }
    halo_wait_barrier<0>();

    int s = 0;    // global step: weight stage parity
    int gb = 0;   // global body: parity of the (scale, shift) buffers
    // NEXT: 0 = the last body, 1 = a KIND 0 body follows, 2 = a KIND 1 body follows (behind a KIND 0 body: the first of its kind)
    auto body = [&](auto kind_c, auto next_c, int idx) {
        constexpr int KIND = decltype(kind_c)::value, NEXT = decltype(next_c)::value;
        typedef std::integral_constant<int, NEXT == 2 ? 1 : 0> KN;   // the next body's kind
        const int nidx = (NEXT == 2 && KIND == 0) ? 0 : idx + 1;
        halo_static_for<9>([&](auto st_c) {
            constexpr int st = decltype(st_c)::value;   // compile-time: taps, wait counts and piece indices depend on it
            constexpr bool w_issue = (st + AHEAD < 9) || NEXT != 0;   // the slice of step s + AHEAD exists
            if constexpr (w_issue && !(HALO_ABL & 1)) {
                if constexpr (st + AHEAD < 9) dma_w(kind_c, idx, st + AHEAD, (s + AHEAD) % WST);
                else dma_w(KN{}, nidx, st + AHEAD - 9, (s + AHEAD) % WST);
            }
            if constexpr (st == 4 && NEXT != 0 && !(HALO_ABL & 2)) {
                dma_ab(KN{}, nidx, (gb + 1) & 1);
                issue_patch(xch(KN{}, nidx, 0), rp);
            }
            if constexpr (st >= 1 && st <= 3 && !(HALO_ABL & 2) && KIND == 0) {          // B of this body -> buffer 1
                float ab16[16];
                lds_ab(abuf + (gb & 1) * AB32_BYTES + 256 + c4 * 64, ab16);
                norm_write(rp[2 * (st - 1)], 2 * (st - 1), ab16, patch + PB);
                if constexpr (2 * (st - 1) + 1 < PL) norm_write(rp[2 * (st - 1) + 1], 2 * (st - 1) + 1, ab16, patch + PB);
            }
            if constexpr (st >= 6 && NEXT != 0 && !(HALO_ABL & 2)) {   // the next body's (first) chunk -> buffer 0; KIND 1 -> KIND 1: the buffer this body does not read
                float ab16[16];
                lds_ab(abuf + ((gb + 1) & 1) * AB32_BYTES + c4 * 64, ab16);
                char* const Pn = (KIND == 1 && NEXT == 2) ? patch + (nidx & 1) * PB : patch;
                norm_write(rp[2 * (st - 6)], 2 * (st - 6), ab16, Pn);
                if constexpr (2 * (st - 6) + 1 < PL) norm_write(rp[2 * (st - 6) + 1], 2 * (st - 6) + 1, ab16, Pn);
            }
            if constexpr (st == 8 && NEXT == 1 && !(HALO_ABL & 2)) issue_patch(xch(K0{}, nidx, 1), rp);
            const char* w_st = wbuf + (s % WST) * W_BYTES;
            if constexpr (KIND == 1) {   // tap st of this body's chunk under both weight blocks: one set of patch fragments per k-step
                constexpr int ky = st / 3, kx = st - ky * 3;
                const char* P = patch + (idx & 1) * PB;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    v8 fb[TM];
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi) {
                        if ((HALO_ABL & 4) && (mi & 1)) fb[mi] = fb[mi - 1];
                        else fb[mi] = *(const v8*)(P + fb_off[kx][ks] + (mi + ky) * (PW * 64));
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        v8 fa[TN];
#pragma unroll
                        for (int ni = 0; ni < TN; ++ni) fa[ni] = *(const v8*)(w_st + j * TAP_BYTES + fa_off[ks] + ni * (32 * 64));
#pragma unroll
                        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                            for (int mi = 0; mi < TM; ++mi) acc[ni][mi] = Mfma<T>::mma(fa[ni], fb[mi], acc[ni][mi]);
                    }
                }
            } else
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int g = 2 * st + j;
                const int par = g >= 9 ? 1 : 0, tap = g >= 9 ? g - 9 : g;
                const int ky = tap / 3, kx = tap - ky * 3;
                const char* P = patch + par * PB;
                const char* w_s = w_st + j * TAP_BYTES;
                if (HQ && par) {   // the e4m3 block of the body's 32 channels: ONE scaled MFMA over K = 64 per tile (lane half 0: x_lo w_hi, 1: x_hi w_lo)
                    typedef int i32x8 __attribute__((ext_vector_type(8)));
                    __builtin_amdgcn_sched_barrier(0);
                    i32x8 qa[TN];
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) {
                        const u32x4 a0 = *(const u32x4*)(w_s + fa_off[0] + ni * (32 * 64)), a1 = *(const u32x4*)(w_s + fa_off[1] + ni * (32 * 64));
                        qa[ni] = (i32x8){(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
                    }
                    // pixel fragments one tile row ahead of their MFMAs; the scheduler may not hoist further (it otherwise lifts every read of
                    // the step to its top and spills 1.3 KiB of fragments: 128 accumulators + 24 prefetch registers leave ~70)
                    auto ldq = [&](int mi) {
                        const u32x4 b0 = *(const u32x4*)(P + fb_off[kx][0] + (mi + ky) * (PW * 64)), b1 = *(const u32x4*)(P + fb_off[kx][1] + (mi + ky) * (PW * 64));
                        return (i32x8){(int)b0[0], (int)b0[1], (int)b0[2], (int)b0[3], (int)b1[0], (int)b1[1], (int)b1[2], (int)b1[3]};
                    };
                    i32x8 qb = ldq(0);
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi) {
                        i32x8 qn = qb;
                        if (mi + 1 < TM) qn = ldq(mi + 1);
#pragma unroll
                        for (int ni = 0; ni < TN; ++ni) {
                            acc[ni][mi] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(qa[ni], qb, acc[ni][mi], 0, 0, 0, q_scale_w, 0, q_scale_x);
                            // pins the product in front of the step's barrier: in the LAST body nothing else consumes the accumulators before the
                            // epilogue, and hipcc sank all 72 scaled MFMAs behind the last barrier with their operands spilled (no instruction emitted)
                            asm volatile("" : "+v"(acc[ni][mi]));
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        qb = qn;
                    }
                    continue;
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    v8 fa[TN], fb[TM];
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) fa[ni] = *(const v8*)(w_s + fa_off[ks] + ni * (32 * 64));
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi) {
                        if ((HALO_ABL & 4) && (mi & 1)) fb[mi] = fb[mi - 1];
                        else fb[mi] = *(const v8*)(P + fb_off[kx][ks] + (mi + ky) * (PW * 64));
                    }
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                        for (int mi = 0; mi < TM; ++mi) acc[ni][mi] = Mfma<T>::mma(fa[ni], fb[mi], acc[ni][mi]);
                }
            }
            // end of step: the weight slice of the NEXT step must have landed (with three stages it was requested a step
            // ago: everything issued in this step may stay in flight); patch loads requested in this step always stay in
            // flight, those requested in the previous step (st 5, st 0) are waited for here
            constexpr int n_w = (w_issue && !(HALO_ABL & 1)) ? 2 * W_LOADS : 0;
            constexpr int n_ab = (st == 4 && NEXT != 0 && NORM != 0 && !(HALO_ABL & 2)) ? 1 : 0;
            constexpr int n_p = (((st == 4 && NEXT != 0) || (st == 8 && NEXT == 1)) && !(HALO_ABL & 2)) ? PL : 0;
            constexpr int keep = (AHEAD == 2 ? n_w + n_ab : 0) + n_p;
            if constexpr (st == 5 || (st == 0 && KIND == 0)) wait_patch(rp, std::integral_constant<int, (AHEAD == 2 ? n_w : 0)>{});
            halo_wait_barrier<keep>();
            ++s;
        });
        ++gb;
    };
    typedef std::integral_constant<int, 0> N0;
    typedef std::integral_constant<int, 1> N1;
    typedef std::integral_constant<int, 2> N2;
    if constexpr (!ALL_PAIRED) {
        for (int u = 0; u + 1 < U; ++u) body(K0{}, N1{}, u);
        if constexpr (PHASED) body(K0{}, N2{}, U - 1);
        else body(K0{}, N0{}, U - 1);
    }
    if constexpr (ALL_PAIRED || PHASED) {
        for (int q = 0; q + 1 < Q; ++q) body(K1{}, N2{}, q);
        body(K1{}, N0{}, Q - 1);
    }

    halo_epilogue<T, BN, TM, TN, NW, SEG>(p, smem, acc, tid, wm, wn, l31, lh, x0, y0, n0, img, tx, ty);
}

// one-time registration of a kernel's dynamic LDS size, keyed on the KERNEL (see conv_igemm.hip: a generic lambda's static is shared by
// every kernel of one function type -- here the NORM and the K-segment variants of a tile)
template <auto KERN> hipError_t halo_smem_once(int smem) {
    static const hipError_t attr = hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    return attr;
}
#define HALO_K(...) std::integral_constant<void (*)(HaloArgs), &__VA_ARGS__>{}

template <typename T, int BN, int NW>
int launch_halo32(HaloArgs a, hipStream_t s) {
    constexpr int THT = 2 * NW;
    constexpr int stage = 2 * ((THT + 2) * PW * 64) + (NW == 8 ? 3 : 2) * (2 * BN * 64) + 2 * AB32_BYTES;
    constexpr int epi = (BN > 64 ? 128 : 256) * (BN + 4) * 4;
    constexpr int red = (64 * NW / (BN / 8)) * BN * 2 * 4;
    constexpr int smem = stage > epi ? (stage > red ? stage : red) : (epi > red ? epi : red);
    const int norm = a.ab == nullptr ? 0 : (a.norm_silu ? 2 : 1);
    a.tiles_y = (a.H + THT - 1) / THT;
    const int64_t nwg = (int64_t)a.tiles_x * a.tiles_y * a.B * ((a.Cout + BN - 1) / BN);
    if (nwg >= ((int64_t)1 << 31)) return RSVLD_EUNSUPPORTED;
    auto go = [&](auto kern_c) -> int {
        constexpr auto kern = decltype(kern_c)::value;
        if (halo_smem_once<kern>(smem) != hipSuccess) return RSVLD_ELAUNCH;   // one-time, thread-safe, per KERNEL
        hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(64 * NW), smem, s, a);
        return rsvld_check_launch();
    };
    if constexpr (__is_same(T, bf16)) {   // RSVLD_SPLIT: planes in, no fused norm (checked by the entry point)
        if (a.seg == 3) return go(HALO_K(conv_halo32_kernel<T, BN, 0, NW, 3>));
    } else {                              // RSVLD_F16W2: fp16 in, weight pairs, the fused norm available
        if (a.seg == 2) {
            if (norm == 0) return go(HALO_K(conv_halo32_kernel<T, BN, 0, NW, 2>));
            if (norm == 1) return go(HALO_K(conv_halo32_kernel<T, BN, 1, NW, 2>));
            return go(HALO_K(conv_halo32_kernel<T, BN, 2, NW, 2>));
        }
        if (a.seg == 5) return norm == 0 ? go(HALO_K(conv_halo32_kernel<T, BN, 0, NW, 5>)) : RSVLD_EUNSUPPORTED;   // RSVLD_F16Q8
    }
    if (a.seg != 1) return RSVLD_EUNSUPPORTED;
    if (norm == 0) return go(HALO_K(conv_halo32_kernel<T, BN, 0, NW, 1>));
    if (norm == 1) return go(HALO_K(conv_halo32_kernel<T, BN, 1, NW, 1>));
    return go(HALO_K(conv_halo32_kernel<T, BN, 2, NW, 1>));
}

template <typename T, int BN, int WAVES_M, int TPS>
int launch_halo(const HaloArgs& a, hipStream_t s) {
    constexpr int stage = PATCH_BYTES + 2 * TPS * BN * 128;
    constexpr int epi = (256 / (BN > 64 ? 2 : 1)) * (BN + 4) * 4;
    constexpr int smem = stage > epi ? stage : epi;
    const int64_t nwg = (int64_t)a.tiles_x * a.tiles_y * a.B * ((a.Cout + BN - 1) / BN);
    if (nwg >= ((int64_t)1 << 31)) return RSVLD_EUNSUPPORTED;
    auto go = [&](auto kern_c) -> int {
        constexpr auto kern = decltype(kern_c)::value;
        if (halo_smem_once<kern>(smem) != hipSuccess) return RSVLD_ELAUNCH;
        hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), smem, s, a);
        return rsvld_check_launch();
    };
    if constexpr (__is_same(T, bf16)) {
        if (a.seg == 3) return go(HALO_K(conv_halo_kernel<T, BN, WAVES_M, TPS, 3>));
    } else {
        if (a.seg == 2) return go(HALO_K(conv_halo_kernel<T, BN, WAVES_M, TPS, 2>));
    }
    if (a.seg != 1) return RSVLD_EUNSUPPORTED;
    return go(HALO_K(conv_halo_kernel<T, BN, WAVES_M, TPS, 1>));
}

template <typename T>
int dispatch_halo(const HaloArgs& a, hipStream_t s) {
    if (a.Cout <= 64) return launch_halo<T, 64, 4, 2>(a, s);
    // 16x32-pixel tiles (one 8-wave workgroup per CU) once that grid still covers most of the chip
    const int64_t wg16 = (int64_t)a.tiles_x * ((a.H + 15) / 16) * a.B_plan * ((a.Cout + 127) / 128);
    // measured (tools/bench_halo.py, one box): +3..13 % from 192 channels of K up, -3 % at 128 (longer pipeline fill)
    const bool use8 = (a.tune & RSVLD_TUNE_HALO_NW8) ? true : (a.tune & RSVLD_TUNE_HALO_NW4) ? false
                                                             : (wg16 >= 192 && a.Ctot >= 192);
    if (use8) return launch_halo32<T, 128, 8>(a, s);
    return launch_halo32<T, 128, 4>(a, s);
}

}  // namespace

// 3x3 / stride 1 / pad 1, no up-sampling, every source a multiple of 64 channels
extern "C" int rsvld_conv3x3_halo_supported(const rsvld_conv_desc* d) {
    if (d == nullptr) return 0;
    if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad_t != 1 || d->pad_l != 1) return 0;
    const int up = d->upsample ? 2 : 1;
    if (d->Ho != up * d->H || d->Wo != up * d->W) return 0;
    if (d->Cin % 64 != 0 || d->Cin2 % 64 != 0) return 0;
    if (d->act == RSVLD_ACT_GEGLU) return 0;
    if (d->out_f32 && d->Cout > 32 && d->dtype != RSVLD_SPLIT && d->dtype != RSVLD_F16W2 && d->dtype != RSVLD_F16Q8) return 0;
    if (d->dtype == RSVLD_F16Q8 && (d->Cout <= 64 || !d->out_f32 || d->upsample || d->Cin2 != 0)) return 0;   // the 128-wide kernels only; fp32 out; ONE source
                                                                                                              // (the GroupNorm apply pass that writes q8 rows has already concatenated)
    if (d->Wo < 16 || d->Ho < 4) return 0;   // tiny maps: the 8x32 tile would be mostly padding
    return 1;
}

extern "C" int rsvld_conv3x3_halo_nhwc(const rsvld_conv_desc* d, const float* norm_scale_shift, int norm_silu,
                                       float* out_stats_partials, void* stream) {
    if (!rsvld_conv3x3_halo_supported(d)) return RSVLD_EUNSUPPORTED;
    if (d->x == nullptr || d->w == nullptr || d->out == nullptr) return RSVLD_EINVAL;
    if (d->B <= 0 || d->Cout <= 0 || d->Cout % 8 != 0) return RSVLD_EINVAL;
    if ((d->Cin2 > 0) != (d->x2 != nullptr)) return RSVLD_EINVAL;
    const bool split = d->dtype == RSVLD_SPLIT, w2 = d->dtype == RSVLD_F16W2, hq = d->dtype == RSVLD_F16Q8;
    const int seg = split ? 3 : w2 ? 2 : hq ? 5 : 1;
    if (d->dtype != RSVLD_F16 && d->dtype != RSVLD_BF16 && seg == 1) return RSVLD_EINVAL;
    if (d->out_f32 < 0 || d->out_f32 > 1) return RSVLD_EINVAL;   // (an fp16 output of RSVLD_SPLIT exists for the Linear layers only)
    if ((split || hq) && norm_scale_shift != nullptr) return RSVLD_EUNSUPPORTED;   // the normalised tensor is split by its own kernel (rsvld_groupnorm_apply_split)
    if (split && !d->out_f32 && d->residual != nullptr) return RSVLD_EINVAL;
    if ((int64_t)d->Ho * d->Wo >= ((int64_t)1 << 31)) return RSVLD_EUNSUPPORTED;
    if (norm_scale_shift != nullptr && d->upsample) return RSVLD_EUNSUPPORTED;   // no GroupNorm sits before an Upsample conv
    HaloArgs a;
    a.x = d->x; a.x2 = d->x2; a.w = d->w; a.bias = d->bias; a.rowvec = d->rowvec; a.residual = d->residual; a.out = d->out;
    a.ab = norm_scale_shift;
    a.stats = (d->out_f32 && seg == 1) ? nullptr : out_stats_partials;
    a.seg = seg;
    a.Cseg = d->Cin + d->Cin2;
    a.B = d->B; a.H = d->Ho; a.W = d->Wo; a.Cin = d->Cin; a.Cin2 = d->Cin2; a.Cout = d->Cout;
    a.B_plan = d->plan_div > 1 ? (d->B + d->plan_div - 1) / d->plan_div : d->B;
    a.tune = d->tune;
    a.Hs = d->H; a.Ws = d->W; a.ush = d->upsample ? 1 : 0;
    a.out_f32 = d->out_f32 ? 1 : 0; a.act = d->act; a.norm_silu = norm_silu ? 1 : 0;
    a.alpha = d->alpha; a.beta = d->beta;
    a.Ctot = (hq ? 2 : seg) * (d->Cin + d->Cin2);   // logical K' channels per tap (16-bit elements of a weight row)
    a.nchunks = a.Ctot / 64;
    a.tiles_x = (d->Wo + TW - 1) / TW;
    a.tiles_y = (d->Ho + TH - 1) / TH;
    a.tiles_y8 = a.tiles_y;
    a.rv_stride = d->rowvec_stride > 0 ? d->rowvec_stride : d->Cout;
    a.Cout_out = d->Cout;
    hipStream_t s = (hipStream_t)stream;
    return (d->dtype == RSVLD_F16 || w2 || hq) ? dispatch_halo<f16>(a, s) : dispatch_halo<bf16>(a, s);   // RSVLD_SPLIT runs the bf16 kernels, RSVLD_F16W2 the fp16 ones
}
