// conv_igemm.hip — implicit-GEMM convolution / linear for gfx950 on v_mfma_f32_32x32x16.
//
//   OutT[co][m] = sum_k Wp[co][k] * A[m][k],   m = (b,oy,ox) output pixel, k = (ky,kx,ci)
//
// The MFMA "A" operand is the weight tile (rows = output channels), the "B" operand is the
// im2col activation tile (columns = output pixels), so each lane ends up owning ONE pixel
// and, per accumulator register quad, FOUR CONSECUTIVE output channels: the epilogue can
// then move 16 B per lane into the LDS out-tile and leave HBM in full 16-byte NHWC rows.
//
// Data movement (cdna_hip_programming.md §5 "glds vs register staging", T14):
//   HBM --global_load_dwordx4 (im2col gather, zero-fill by predicate)--> VGPR
//       --ds_write_b128 (XOR-swizzled 128-B rows)--> LDS (double buffered, BK = 64)
//       --ds_read_b128 (conflict-free)--> MFMA operands.
// The next K-tile's global loads are issued BEFORE the current tile's MFMAs and written to
// the other LDS buffer after them (one barrier per K-step).
//
// Fusions: nearest x2 up-sampling of the input (index >>1), channel concatenation of two
// inputs, bias, per-image row vector (time embedding), residual add, SiLU / GEGLU.
#include "rsvld_common.h"

namespace {

struct ConvArgs {
    const void* x;
    const void* x2;
    const void* w;
    const float* bias;
    const float* rowvec;
    const void* residual;
    void* out;
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
};

constexpr int BK_BYTES = 128;  // 64 x 16-bit per LDS row

// byte offset of 16-byte chunk `c` (0..7) of LDS row `row` (128-B rows).  Two rows share
// one 256-B bank row; XOR with (row>>1)&7 makes every ds_read_b128 lane group hit 16
// distinct 16-B slots (MI355X_MICROARCH.md §LDS).
__device__ __forceinline__ int lds_off(int row, int c) { return row * BK_BYTES + ((c ^ ((row >> 1) & 7)) << 4); }

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs p) {
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int A_LOADS = BM / 32, B_LOADS = BN / 32;
    constexpr int A_BYTES = BM * BK_BYTES, B_BYTES = BN * BK_BYTES;
    constexpr int STAGE = A_BYTES + B_BYTES;
    typedef typename Mfma<T>::v8 v8;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    // ---- staging roles: thread -> 16-byte chunk c of rows r0, r0+32, ...
    const int c = tid & 7;
    const int r0 = tid >> 3;

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
    int ci = c, ky = 0, kx = 0;
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

    auto load_tile = [&](int kt) {
        const bool kvalid = ky < p.KH;
        const T* src;
        int cc, Cs;
        if (ci < p.C1_8) { src = X1; cc = ci; Cs = p.Cin; } else { src = X2; cc = ci - p.C1_8; Cs = p.Cin2; }
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

    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int l31 = lane & 31, lh = lane >> 5;
    for (int kt = 0; kt < p.nk; ++kt) {
        const bool more = kt + 1 < p.nk;
        if (more) {
            ci += 8;
            normalize();
            load_tile(kt + 1);
        }
        const char* a_s = smem + (kt & 1) * STAGE;
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
        if (more) store_tile((kt + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: accumulators -> LDS out tile Ct[pixel][cout] (fp32, row stride BN+4)
    constexpr int CT_STRIDE = BN + 4;
    float* Ct = (float*)smem;
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int row = wm * WTM + mi * 32 + l31;
                const int col = wn * WTN + ni * 32 + 8 * g + 4 * lh;
                f32x4 v = {acc[ni][mi][4 * g], acc[ni][mi][4 * g + 1], acc[ni][mi][4 * g + 2], acc[ni][mi][4 * g + 3]};
                *(f32x4*)(Ct + row * CT_STRIDE + col) = v;
            }
    __syncthreads();

    constexpr int CPR = BN / 8;          // 8-channel chunks per tile row
    constexpr int RPP = 256 / CPR;       // rows per pass
    const int cc = tid % CPR;
    const int rr = tid / CPR;
    const int n = n0 + cc * 8;
    if (n >= p.Cout) return;
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = (p.bias != nullptr && n + e < p.Cout) ? p.bias[n + e] : 0.f;

    for (int row = rr; row < BM; row += RPP) {
        const int m = m0 + row;
        if (m >= p.M) break;
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
            const u32x4 rres = *(const u32x4*)((const T*)p.residual + (int64_t)m * p.Cout_out + n);
            float rf[8];
            unpack8<T>(rres, rf);
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

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N>
int launch_conv(const ConvArgs& a, hipStream_t s) {
    constexpr int stage = 2 * (BM + BN) * BK_BYTES;
    constexpr int epi = BM * (BN + 4) * 4;
    constexpr int smem = stage > epi ? stage : epi;
    auto kern = conv_igemm_kernel<T, BM, BN, WAVES_M, WAVES_N>;
    static bool attr_set = false;  // per instantiation
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return RSVLD_ELAUNCH;
        attr_set = true;
    }
    dim3 grid((unsigned)((a.M + BM - 1) / BM), (unsigned)((a.Cout + BN - 1) / BN));
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, a);
    return rsvld_check_launch();
}

template <typename T>
int dispatch_conv(const ConvArgs& a, hipStream_t s) {
    if (a.Cout <= 32) return launch_conv<T, 256, 32, 4, 1>(a, s);
    if (a.Cout <= 64) return launch_conv<T, 256, 64, 4, 1>(a, s);
    return launch_conv<T, 128, 128, 2, 2>(a, s);
}

}  // namespace

extern "C" int rsvld_conv2d_nhwc(const rsvld_conv_desc* d, void* stream) {
    if (d == nullptr || d->x == nullptr || d->w == nullptr || d->out == nullptr) return RSVLD_EINVAL;
    if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Ho <= 0 || d->Wo <= 0) return RSVLD_EINVAL;
    if (d->Cin <= 0 || d->Cin % 8 != 0 || d->Cout <= 0 || d->Cout % 8 != 0) return RSVLD_EINVAL;
    if (d->Cin2 < 0 || d->Cin2 % 8 != 0 || ((d->Cin2 > 0) != (d->x2 != nullptr))) return RSVLD_EINVAL;
    if (d->KH <= 0 || d->KW <= 0 || d->stride <= 0) return RSVLD_EINVAL;
    if (d->dtype != RSVLD_F16 && d->dtype != RSVLD_BF16) return RSVLD_EINVAL;
    if (d->out_f32 && (d->Cout > 32 || d->act == RSVLD_ACT_GEGLU || d->residual != nullptr)) return RSVLD_EUNSUPPORTED;
    if (d->act == RSVLD_ACT_GEGLU && (d->Cout % 16 != 0 || d->residual != nullptr)) return RSVLD_EINVAL;
    if ((int64_t)d->B * d->Ho * d->Wo >= (int64_t)1 << 31) return RSVLD_EUNSUPPORTED;
    if ((int64_t)d->B * d->H * d->W >= (int64_t)1 << 31) return RSVLD_EUNSUPPORTED;
    // every output pixel must read inside the (optionally up-sampled) padded input: shape sanity
    {
        const int Hin = d->upsample ? 2 * d->H : d->H, Win = d->upsample ? 2 * d->W : d->W;
        if ((d->Ho - 1) * d->stride - d->pad_t >= Hin || (d->Wo - 1) * d->stride - d->pad_l >= Win) return RSVLD_EINVAL;
    }
    ConvArgs a;
    a.x = d->x; a.x2 = d->x2; a.w = d->w; a.bias = d->bias; a.rowvec = d->rowvec;
    a.residual = d->residual; a.out = d->out;
    a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cin2 = d->Cin2; a.Cout = d->Cout;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad_t = d->pad_t; a.pad_l = d->pad_l;
    a.Ho = d->Ho; a.Wo = d->Wo; a.upsample = d->upsample ? 1 : 0;
    a.out_f32 = d->out_f32 ? 1 : 0; a.act = d->act; a.alpha = d->alpha; a.beta = d->beta;
    a.M = d->B * d->Ho * d->Wo;
    a.HoWo = d->Ho * d->Wo;
    a.C1_8 = d->Cin / 8;
    a.Ctot8 = (d->Cin + d->Cin2) / 8;
    a.KC = d->KH * d->KW * a.Ctot8;
    a.nk = (a.KC + 7) / 8;
    a.Cout_out = d->act == RSVLD_ACT_GEGLU ? d->Cout / 2 : d->Cout;
    a.rv_stride = d->rowvec_stride > 0 ? d->rowvec_stride : d->Cout;
    hipStream_t s = (hipStream_t)stream;
    return d->dtype == RSVLD_F16 ? dispatch_conv<f16>(a, s) : dispatch_conv<bf16>(a, s);
}
