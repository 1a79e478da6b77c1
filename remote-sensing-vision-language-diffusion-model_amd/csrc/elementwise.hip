// elementwise.hip — HBM-bound layout converters, sampler updates and tiny embedding layers.
#include "rsvld_common.h"

namespace {

// fp32 NCHW -> 16-bit NHWC (channel window [c_off, c_off+C) of a Cdst-wide row); one thread per pixel
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int C, int64_t HW, int Cdst,
                                    int c_off, int zero_pad, float scale, int64_t total_pix) {
    const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= total_pix) return;
    const int64_t b = pix / HW, r = pix - b * HW;
    T* d = dst + pix * Cdst;
    if (zero_pad) {
        for (int c = 0; c < Cdst; ++c) {
            const int cs = c - c_off;
            d[c] = (cs >= 0 && cs < C) ? (T)(src[(b * C + cs) * HW + r] * scale) : (T)0.f;
        }
    } else {
        for (int c = 0; c < C; ++c) d[c_off + c] = (T)(src[(b * C + c) * HW + r] * scale);
    }
}

template <typename T, bool SRC_F32>
__global__ void nhwc_to_nchw_kernel(const void* __restrict__ src, float* __restrict__ dst, int C, int64_t HW, int Csrc,
                                    int c_off, int64_t total_pix) {
    const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= total_pix) return;
    const int64_t b = pix / HW, r = pix - b * HW;
    for (int c = 0; c < C; ++c) {
        float v;
        if (SRC_F32) v = ((const float*)src)[pix * Csrc + c_off + c];
        else v = (float)((const T*)src)[pix * Csrc + c_off + c];
        dst[(b * C + c) * HW + r] = v;
    }
}

template <typename T>
__global__ void axpby_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o, int64_t n8, float sa,
                             float sb) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float fa[8], fb[8];
        unpack8<T>(*(const u32x4*)(a + i * 8), fa);
        unpack8<T>(*(const u32x4*)(b + i * 8), fb);
#pragma unroll
        for (int e = 0; e < 8; ++e) fa[e] = fa[e] * sa + fb[e] * sb;
        *(u32x4*)(o + i * 8) = pack8<T>(fa);
    }
}

// in [rows, 2C] = [value | gate]  ->  out [rows, C] = value * gelu(gate)
template <typename T>
__global__ void geglu_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t rows, int C8) {
    const int64_t total = rows * C8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / C8;
        const int cc = (int)(i - r * C8);
        float v[8], g[8];
        unpack8<T>(*(const u32x4*)(in + (r * 2 * C8 + cc) * 8), v);
        unpack8<T>(*(const u32x4*)(in + (r * 2 * C8 + C8 + cc) * 8), g);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= gelu_erf_f(g[e]);
        *(u32x4*)(out + (r * C8 + cc) * 8) = pack8<T>(v);
    }
}

__global__ void ddpm_step_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                 const float* __restrict__ noise, float* __restrict__ xo, int C, int64_t HW, int eps_c,
                                 float c_recip, float c_recipm1, float coef1, float coef2, float sigma, int clip,
                                 int64_t total_pix) {
    const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= total_pix) return;
    const int64_t b = pix / HW, r = pix - b * HW;
    for (int c = 0; c < C; ++c) {
        const int64_t i = (b * C + c) * HW + r;
        const float xv = x[i];
        float x0 = c_recip * xv - c_recipm1 * eps[pix * eps_c + c];
        if (clip) x0 = fminf(1.f, fmaxf(-1.f, x0));
        float o = coef1 * x0 + coef2 * xv;
        if (noise != nullptr) o += sigma * noise[i];
        xo[i] = o;
    }
}

// y[r][o] = act_out( b[o] + sum_i W[o][i] * act_in(x[r][i]) ) ; one wave per output feature
__global__ __launch_bounds__(256) void linear_small_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           int rows, int in_f, int out_f, int act_in, int act_out) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= out_f) return;
    const float* wr = w + (int64_t)o * in_f;
    for (int r = 0; r < rows; ++r) {
        const float* xr = x + (int64_t)r * in_f;
        float acc = 0.f;
        for (int i = lane; i < in_f; i += 64) {
            float xv = xr[i];
            if (act_in == 1) xv = silu_f(xv);
            acc += wr[i] * xv;
        }
        acc = wave_sum(acc);
        if (lane == 0) {
            float v = acc + (bias ? bias[o] : 0.f);
            if (act_out == 1) v = silu_f(v);
            y[(int64_t)r * out_f + o] = v;
        }
    }
}

__global__ void sinusoidal_kernel(const float* __restrict__ t, float* __restrict__ out, int rows, int dim, int kind) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = dim / 2;
    if (i >= rows * half) return;
    const int r = i / half, k = i - r * half;
    // op order follows the references: kind 0 exp(-ln(1e4) * (k/half)); kind 1 exp((-ln(1e4)*k)/half)
    const float nl = -9.210340371976184f;
    const float freq = kind == 0 ? expf(nl * ((float)k / (float)half)) : expf((nl * (float)k) / (float)half);
    const float a = t[r] * freq;
    float* o = out + (int64_t)r * dim;
    if (kind == 0) { o[k] = sinf(a); o[half + k] = cosf(a); }
    else { o[k] = cosf(a); o[half + k] = sinf(a); }
}

}  // namespace

extern "C" const char* rsvld_version(void) { return "rsvld-hip 0.1 (gfx950)"; }

extern "C" int rsvld_nchw_f32_to_nhwc(const float* src, void* dst, int B, int C, int H, int W, int Cdst, int c_off,
                                      int zero_pad, float scale, int dtype, void* stream) {
    if (!src || !dst || B <= 0 || C <= 0 || H <= 0 || W <= 0 || c_off < 0 || c_off + C > Cdst) return RSVLD_EINVAL;
    const int64_t HW = (int64_t)H * W, total = HW * B;
    const unsigned nblk = (unsigned)cdiv64(total, 256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == RSVLD_F16)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<f16>, dim3(nblk), dim3(256), 0, s, src, (f16*)dst, C, HW, Cdst, c_off, zero_pad, scale, total);
    else if (dtype == RSVLD_BF16)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16>, dim3(nblk), dim3(256), 0, s, src, (bf16*)dst, C, HW, Cdst, c_off, zero_pad, scale, total);
    else
        return RSVLD_EINVAL;
    return rsvld_check_launch();
}

extern "C" int rsvld_nhwc_to_nchw_f32(const void* src, float* dst, int B, int C, int H, int W, int Csrc, int c_off,
                                      int src_f32, int dtype, void* stream) {
    if (!src || !dst || B <= 0 || C <= 0 || H <= 0 || W <= 0 || c_off < 0 || c_off + C > Csrc) return RSVLD_EINVAL;
    const int64_t HW = (int64_t)H * W, total = HW * B;
    const unsigned nblk = (unsigned)cdiv64(total, 256);
    hipStream_t s = (hipStream_t)stream;
    if (src_f32)
        hipLaunchKernelGGL((nhwc_to_nchw_kernel<f16, true>), dim3(nblk), dim3(256), 0, s, src, dst, C, HW, Csrc, c_off, total);
    else if (dtype == RSVLD_F16)
        hipLaunchKernelGGL((nhwc_to_nchw_kernel<f16, false>), dim3(nblk), dim3(256), 0, s, src, dst, C, HW, Csrc, c_off, total);
    else if (dtype == RSVLD_BF16)
        hipLaunchKernelGGL((nhwc_to_nchw_kernel<bf16, false>), dim3(nblk), dim3(256), 0, s, src, dst, C, HW, Csrc, c_off, total);
    else
        return RSVLD_EINVAL;
    return rsvld_check_launch();
}

extern "C" int rsvld_axpby(const void* a, const void* b, void* out, int64_t n, float sa, float sb, int dtype,
                           void* stream) {
    if (!a || !b || !out || n <= 0 || (n & 7)) return RSVLD_EINVAL;
    const int64_t n8 = n / 8;
    int64_t nblk = cdiv64(n8, 256);
    if (nblk > 4096) nblk = 4096;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == RSVLD_F16)
        hipLaunchKernelGGL(axpby_kernel<f16>, dim3((unsigned)nblk), dim3(256), 0, s, (const f16*)a, (const f16*)b, (f16*)out, n8, sa, sb);
    else if (dtype == RSVLD_BF16)
        hipLaunchKernelGGL(axpby_kernel<bf16>, dim3((unsigned)nblk), dim3(256), 0, s, (const bf16*)a, (const bf16*)b, (bf16*)out, n8, sa, sb);
    else
        return RSVLD_EINVAL;
    return rsvld_check_launch();
}

extern "C" int rsvld_geglu(const void* in, void* out, int64_t rows, int C, int dtype, void* stream) {
    if (!in || !out || rows <= 0 || C <= 0 || (C & 7)) return RSVLD_EINVAL;
    const int C8 = C / 8;
    int64_t nblk = cdiv64(rows * C8, 256);
    if (nblk > 4096) nblk = 4096;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == RSVLD_F16)
        hipLaunchKernelGGL(geglu_kernel<f16>, dim3((unsigned)nblk), dim3(256), 0, s, (const f16*)in, (f16*)out, rows, C8);
    else if (dtype == RSVLD_BF16)
        hipLaunchKernelGGL(geglu_kernel<bf16>, dim3((unsigned)nblk), dim3(256), 0, s, (const bf16*)in, (bf16*)out, rows, C8);
    else
        return RSVLD_EINVAL;
    return rsvld_check_launch();
}

extern "C" int rsvld_ddpm_step(const float* x, const float* eps_nhwc, const float* noise, float* x_out, int B, int C,
                               int H, int W, int eps_c, float c_recip, float c_recipm1, float coef1, float coef2,
                               float sigma, int clip, void* stream) {
    if (!x || !eps_nhwc || !x_out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || eps_c < C) return RSVLD_EINVAL;
    const int64_t HW = (int64_t)H * W, total = HW * B;
    hipLaunchKernelGGL(ddpm_step_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       eps_nhwc, noise, x_out, C, HW, eps_c, c_recip, c_recipm1, coef1, coef2, sigma, clip, total);
    return rsvld_check_launch();
}

extern "C" int rsvld_linear_small_f32(const float* x, const float* w, const float* b, float* y, int rows, int in_f,
                                      int out_f, int act_in, int act_out, void* stream) {
    if (!x || !w || !y || rows <= 0 || rows > 4096 || in_f <= 0 || out_f <= 0) return RSVLD_EINVAL;
    hipLaunchKernelGGL(linear_small_kernel, dim3((unsigned)((out_f + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, w, b,
                       y, rows, in_f, out_f, act_in, act_out);
    return rsvld_check_launch();
}

extern "C" int rsvld_sinusoidal_embedding(const float* t, float* out, int rows, int dim, int kind, void* stream) {
    if (!t || !out || rows <= 0 || dim <= 0 || (dim & 1) || (kind != 0 && kind != 1)) return RSVLD_EINVAL;
    const int total = rows * (dim / 2);
    hipLaunchKernelGGL(sinusoidal_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, t, out,
                       rows, dim, kind);
    return rsvld_check_launch();
}
