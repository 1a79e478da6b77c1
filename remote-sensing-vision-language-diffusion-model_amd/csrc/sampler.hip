// sampler.hip — HBM-bound kernels of the Stage-2 sampler, the feature-cache test, the VAE
// posterior and the wavelet colour fix.  All sampler state is fp32 NCHW (the reference keeps
// sigma scaling, CFG, the Euler update and the cache similarity in fp32).
#include "rsvld_common.h"

namespace {

// out[b,c,y,x] = net_out_nhwc[b,y,x,c] * c_out + input[b,c,y,x] * c_skip     (denoiser.py:77-78)
__global__ void denoiser_out_kernel(const float* __restrict__ net_out, const float* __restrict__ input,
                                    float* __restrict__ out, int C, int64_t HW, int c_pad, float c_out, float c_skip,
                                    int64_t total_pix) {
    const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= total_pix) return;
    const int64_t b = pix / HW, r = pix - b * HW;
    for (int c = 0; c < C; ++c) {
        const int64_t i = (b * C + c) * HW + r;
        out[i] = net_out[pix * c_pad + c] * c_out + input[i] * c_skip;
    }
}

// out = a + w*(b - a)   (CFG combine x_u + s(x_c - x_u), sampling_utils.py:7-9)
__global__ void lerp_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o,
                                int64_t n, float w) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        o[i] = a[i] + w * (b[i] - a[i]);
}
// out = x + s*y (x may be NULL: out = s*y)   (churn noise injection, sampling.py:600-606; x *= sqrt(1+sigma0^2), :49)
__global__ void axpy_f32_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ o,
                                int64_t n, float s) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        o[i] = (x != nullptr ? x[i] : 0.f) + s * y[i];
}

// restore pull + Euler step (sampling.py:614-620)
__global__ void euler_step_kernel(const float* __restrict__ x_hat, const float* __restrict__ denoised,
                                  const float* __restrict__ x_center, float* __restrict__ x_out, int64_t n,
                                  float restore_w, float sigma_hat, float dt) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float dn = denoised[i];
        if (x_center != nullptr) dn = dn - (dn - x_center[i]) * restore_w;
        const float xv = x_hat[i];
        const float d = (xv - dn) / sigma_hat;
        x_out[i] = xv + d * dt;
    }
}

// per-row partial sums of |a-b| and |a| over 16-bit tensors: part[row][chunk] = (sum_abs_diff, sum_abs_a)
template <typename T>
__global__ __launch_bounds__(256) void absdiff_partial_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                              float* __restrict__ part, int64_t n8_per_row,
                                                              int64_t chunk8, int nchunks) {
    __shared__ float red[2][4];
    const int row = blockIdx.y, chunk = blockIdx.x;
    const int64_t lo = (int64_t)chunk * chunk8, hi = min(n8_per_row, lo + chunk8);
    const T* pa = a + (int64_t)row * n8_per_row * 8;
    const T* pb = b + (int64_t)row * n8_per_row * 8;
    float sd = 0.f, sa = 0.f;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        float fa[8], fb[8];
        unpack8<T>(*(const u32x4*)(pa + i * 8), fa);
        unpack8<T>(*(const u32x4*)(pb + i * 8), fb);
#pragma unroll
        for (int e = 0; e < 8; ++e) { sd += fabsf(fa[e] - fb[e]); sa += fabsf(fa[e]); }
    }
    sd = wave_sum(sd);
    sa = wave_sum(sa);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sd; red[1][threadIdx.x >> 6] = sa; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* o = part + ((int64_t)row * nchunks + chunk) * 2;
        o[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        o[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}
__global__ void absdiff_finalize_kernel(const float* __restrict__ part, float* __restrict__ out, int rows, int nchunks) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    double sd = 0.0, sa = 0.0;
    for (int c = 0; c < nchunks; ++c) {
        sd += (double)part[((int64_t)row * nchunks + c) * 2];
        sa += (double)part[((int64_t)row * nchunks + c) * 2 + 1];
    }
    out[2 * row] = (float)sd;
    out[2 * row + 1] = (float)sa;
}

// z = (mean + exp(0.5*clamp(logvar,-30,20)) * noise) * scale  from NHWC moments [B,H,W,m_c] (mean = channels
// 0..C-1, logvar = C..2C-1) -> fp32 NCHW [B,C,H,W]; noise NULL -> mode()   (distributions.py:24-41,71-72)
template <typename T, bool F32>
__global__ void gaussian_sample_kernel(const void* __restrict__ mom, const float* __restrict__ noise,
                                       float* __restrict__ z, int C, int64_t HW, int m_c, float scale,
                                       int64_t total_pix) {
    const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= total_pix) return;
    const int64_t b = pix / HW, r = pix - b * HW;
    for (int c = 0; c < C; ++c) {
        float mean, logvar;
        if (F32) { mean = ((const float*)mom)[pix * m_c + c]; logvar = ((const float*)mom)[pix * m_c + C + c]; }
        else { mean = (float)((const T*)mom)[pix * m_c + c]; logvar = (float)((const T*)mom)[pix * m_c + C + c]; }
        const int64_t i = (b * C + c) * HW + r;
        float v = mean;
        if (noise != nullptr) {
            logvar = fminf(20.f, fmaxf(-30.f, logvar));
            v = mean + expf(0.5f * logvar) * noise[i];
        }
        z[i] = v * scale;
    }
}

// depthwise 3x3 [1,2,1]x[1,2,1]/16 blur with dilation `radius` and replicate padding on fp32 NCHW planes
// (utils/colorfix.py:73-92).  mode 0: low = blur(img); mode 1: also high += img - low (wavelet level)
__global__ void wavelet_blur_kernel(const float* __restrict__ img, float* __restrict__ low, float* __restrict__ high,
                                    int H, int W, int radius, int64_t planes) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int64_t pl = blockIdx.z;
    if (x >= W) return;
    const float* p = img + pl * (int64_t)H * W;
    const int ym = max(y - radius, 0), yp = min(y + radius, H - 1);
    const int xm = max(x - radius, 0), xp = min(x + radius, W - 1);
    // accumulate in the order F.conv2d's reference loop would not matter here: 9 products, fp32
    float acc = 0.0625f * p[(int64_t)ym * W + xm] + 0.125f * p[(int64_t)ym * W + x] + 0.0625f * p[(int64_t)ym * W + xp] +
                0.125f * p[(int64_t)y * W + xm] + 0.25f * p[(int64_t)y * W + x] + 0.125f * p[(int64_t)y * W + xp] +
                0.0625f * p[(int64_t)yp * W + xm] + 0.125f * p[(int64_t)yp * W + x] + 0.0625f * p[(int64_t)yp * W + xp];
    const int64_t o = pl * (int64_t)H * W + (int64_t)y * W + x;
    low[o] = acc;
    if (high != nullptr) high[o] += p[(int64_t)y * W + x] - acc;
}

// out = a + b (fp32)
__global__ void add_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        o[i] = a[i] + b[i];
}

// per-(b,c) plane mean / unbiased variance over H*W for AdaIN (utils/colorfix.py:48-71): one block per plane
__global__ __launch_bounds__(256) void plane_stats_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t HW) {
    __shared__ double red[2][4];
    const float* p = x + (int64_t)blockIdx.x * HW;
    double s = 0.0, ss = 0.0;
    for (int64_t i = threadIdx.x; i < HW; i += 256) { const double v = p[i]; s += v; ss += v * v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        ss = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        const double mean = s / (double)HW;
        double var = (ss - s * mean) / (double)(HW - 1);  // unbiased, as torch.var default
        out[2 * blockIdx.x] = (float)mean;
        out[2 * blockIdx.x + 1] = (float)var;
    }
}
// out = (content - cm)/cs * ss + sm per plane; stats = [planes][2] (mean, var), eps 1e-5 added to var
__global__ void adain_apply_kernel(const float* __restrict__ content, const float* __restrict__ cstat,
                                   const float* __restrict__ sstat, float* __restrict__ out, int64_t HW, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pl = i / HW;
        const float cm = cstat[2 * pl], cs = sqrtf(cstat[2 * pl + 1] + 1e-5f);
        const float sm = sstat[2 * pl], ss = sqrtf(sstat[2 * pl + 1] + 1e-5f);
        out[i] = (content[i] - cm) / cs * ss + sm;
    }
}

// NHWC concat along C of two 16-bit tensors
template <typename T>
__global__ void concat_c_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o, int64_t rows,
                                int C1_8, int C2_8) {
    const int C8 = C1_8 + C2_8;
    const int64_t total = rows * C8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / C8;
        const int cc = (int)(i - r * C8);
        u32x4 v;
        if (cc < C1_8) v = *(const u32x4*)(a + (r * C1_8 + cc) * 8);
        else v = *(const u32x4*)(b + (r * C2_8 + (cc - C1_8)) * 8);
        *(u32x4*)(o + i * 8) = v;
    }
}

inline unsigned grid1d(int64_t n, int per = 256, int64_t cap = 4096) {
    int64_t g = cdiv64(n, per);
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

extern "C" int rsvld_denoiser_out(const float* net_out_nhwc, const float* input, float* out, int B, int C, int H, int W,
                                  int c_pad, float c_out, float c_skip, void* stream) {
    if (!net_out_nhwc || !input || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || c_pad < C) return RSVLD_EINVAL;
    const int64_t HW = (int64_t)H * W, total = HW * B;
    hipLaunchKernelGGL(denoiser_out_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       net_out_nhwc, input, out, C, HW, c_pad, c_out, c_skip, total);
    return rsvld_check_launch();
}

extern "C" int rsvld_lerp_f32(const float* a, const float* b, float* out, int64_t n, float w, void* stream) {
    if (!a || !b || !out || n <= 0) return RSVLD_EINVAL;
    hipLaunchKernelGGL(lerp_f32_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n, w);
    return rsvld_check_launch();
}

extern "C" int rsvld_axpy_f32(const float* x, const float* y, float* out, int64_t n, float s, void* stream) {
    if (!y || !out || n <= 0) return RSVLD_EINVAL;
    hipLaunchKernelGGL(axpy_f32_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, x, y, out, n, s);
    return rsvld_check_launch();
}

extern "C" int rsvld_euler_step(const float* x_hat, const float* denoised, const float* x_center, float* x_out,
                                int64_t n, float restore_w, float sigma_hat, float dt, void* stream) {
    if (!x_hat || !denoised || !x_out || n <= 0 || sigma_hat == 0.f) return RSVLD_EINVAL;
    hipLaunchKernelGGL(euler_step_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, x_hat, denoised, x_center,
                       x_out, n, restore_w, sigma_hat, dt);
    return rsvld_check_launch();
}

// ---- latent-tile blending of TiledRestoreEDMSampler (sampling.py:733-736): acc[tile window] += tile * w, cnt += w; out = acc / cnt
__global__ __launch_bounds__(256) void tile_blend_accumulate_kernel(float* __restrict__ acc, float* __restrict__ cnt,
                                                                    const float* __restrict__ tile, const float* __restrict__ w,
                                                                    int planes, int H, int W, int y0, int x0, int th, int tw) {
    const int64_t n = (int64_t)planes * th * tw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % tw);
        const int64_t r = i / tw;
        const int y = (int)(r % th);
        const int64_t pl = r / th;
        const float wv = w[y * tw + x];
        const int64_t o = (pl * H + y0 + y) * W + x0 + x;
        acc[o] = fmaf(tile[i], wv, acc[o]);
        cnt[o] += wv;
    }
}

__global__ __launch_bounds__(256) void tile_blend_finish_kernel(const float* __restrict__ acc, const float* __restrict__ cnt,
                                                                float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = acc[i] / cnt[i];
}

extern "C" int rsvld_tile_blend_accumulate(float* acc, float* cnt, const float* tile, const float* weights, int B, int C, int H,
                                           int W, int y0, int x0, int th, int tw, void* stream) {
    if (!acc || !cnt || !tile || !weights || B <= 0 || C <= 0 || th <= 0 || tw <= 0) return RSVLD_EINVAL;
    if (y0 < 0 || x0 < 0 || y0 + th > H || x0 + tw > W) return RSVLD_EINVAL;   // the window must lie inside the latent
    const int64_t n = (int64_t)B * C * th * tw;
    hipLaunchKernelGGL(tile_blend_accumulate_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, acc, cnt, tile, weights,
                       B * C, H, W, y0, x0, th, tw);
    return rsvld_check_launch();
}

extern "C" int rsvld_tile_blend_finish(const float* acc, const float* cnt, float* out, int64_t n, void* stream) {
    if (!acc || !cnt || !out || n <= 0) return RSVLD_EINVAL;
    hipLaunchKernelGGL(tile_blend_finish_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, acc, cnt, out, n);
    return rsvld_check_launch();
}

extern "C" int64_t rsvld_absdiff_ws_bytes(int rows, int64_t n_per_row) {
    if (rows <= 0 || n_per_row <= 0) return 0;
    return (int64_t)rows * 256 * 2 * sizeof(float);
}

extern "C" int rsvld_absdiff_sums(const void* a, const void* b, float* out, int rows, int64_t n_per_row, int dtype,
                                  void* ws, void* stream) {
    if (!a || !b || !out || !ws || rows <= 0 || rows > 65535 || n_per_row <= 0 || (n_per_row & 7)) return RSVLD_EINVAL;
    const int64_t n8 = n_per_row / 8;
    int nchunks = (int)cdiv64(n8, 2048);
    if (nchunks > 256) nchunks = 256;
    const int64_t chunk8 = cdiv64(n8, nchunks);
    nchunks = (int)cdiv64(n8, chunk8);
    hipStream_t s = (hipStream_t)stream;
    float* part = (float*)ws;
    if (dtype == RSVLD_F16)
        hipLaunchKernelGGL(absdiff_partial_kernel<f16>, dim3(nchunks, rows), dim3(256), 0, s, (const f16*)a, (const f16*)b, part, n8, chunk8, nchunks);
    else if (dtype == RSVLD_BF16)
        hipLaunchKernelGGL(absdiff_partial_kernel<bf16>, dim3(nchunks, rows), dim3(256), 0, s, (const bf16*)a, (const bf16*)b, part, n8, chunk8, nchunks);
    else
        return RSVLD_EINVAL;
    hipLaunchKernelGGL(absdiff_finalize_kernel, dim3((rows + 63) / 64), dim3(64), 0, s, part, out, rows, nchunks);
    return rsvld_check_launch();
}

extern "C" int rsvld_gaussian_sample(const void* moments_nhwc, const float* noise, float* z, int B, int C, int H, int W,
                                     int m_c, float scale, int src_f32, int dtype, void* stream) {
    if (!moments_nhwc || !z || B <= 0 || C <= 0 || H <= 0 || W <= 0 || m_c < 2 * C) return RSVLD_EINVAL;
    const int64_t HW = (int64_t)H * W, total = HW * B;
    const unsigned nb = (unsigned)cdiv64(total, 256);
    hipStream_t s = (hipStream_t)stream;
    if (src_f32)
        hipLaunchKernelGGL((gaussian_sample_kernel<f16, true>), dim3(nb), dim3(256), 0, s, moments_nhwc, noise, z, C, HW, m_c, scale, total);
    else if (dtype == RSVLD_F16)
        hipLaunchKernelGGL((gaussian_sample_kernel<f16, false>), dim3(nb), dim3(256), 0, s, moments_nhwc, noise, z, C, HW, m_c, scale, total);
    else if (dtype == RSVLD_BF16)
        hipLaunchKernelGGL((gaussian_sample_kernel<bf16, false>), dim3(nb), dim3(256), 0, s, moments_nhwc, noise, z, C, HW, m_c, scale, total);
    else
        return RSVLD_EINVAL;
    return rsvld_check_launch();
}

extern "C" int rsvld_wavelet_blur(const float* img, float* low, float* high_accum, int planes, int H, int W, int radius,
                                  void* stream) {
    if (!img || !low || planes <= 0 || planes > 65535 || H <= 0 || H > 65535 || W <= 0 || radius <= 0 || img == low) return RSVLD_EINVAL;
    hipLaunchKernelGGL(wavelet_blur_kernel, dim3((W + 255) / 256, H, planes), dim3(256), 0, (hipStream_t)stream, img, low,
                       high_accum, H, W, radius, (int64_t)planes);
    return rsvld_check_launch();
}

extern "C" int rsvld_add_f32(const float* a, const float* b, float* out, int64_t n, void* stream) {
    if (!a || !b || !out || n <= 0) return RSVLD_EINVAL;
    hipLaunchKernelGGL(add_f32_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
    return rsvld_check_launch();
}

extern "C" int rsvld_adain(const float* content, const float* style, float* out, float* ws_stats, int planes, int64_t HW,
                           void* stream) {
    if (!content || !style || !out || !ws_stats || planes <= 0 || HW <= 1) return RSVLD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(plane_stats_kernel, dim3(planes), dim3(256), 0, s, content, ws_stats, HW);
    hipLaunchKernelGGL(plane_stats_kernel, dim3(planes), dim3(256), 0, s, style, ws_stats + 2 * planes, HW);
    const int64_t n = (int64_t)planes * HW;
    hipLaunchKernelGGL(adain_apply_kernel, dim3(grid1d(n)), dim3(256), 0, s, content, ws_stats, ws_stats + 2 * planes, out, HW, n);
    return rsvld_check_launch();
}

extern "C" int rsvld_concat_c(const void* a, const void* b, void* out, int64_t rows, int C1, int C2, int dtype,
                              void* stream) {
    if (!a || !b || !out || rows <= 0 || C1 <= 0 || C2 <= 0 || (C1 & 7) || (C2 & 7)) return RSVLD_EINVAL;
    const int64_t total = rows * ((C1 + C2) / 8);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == RSVLD_F16)
        hipLaunchKernelGGL(concat_c_kernel<f16>, dim3(grid1d(total)), dim3(256), 0, s, (const f16*)a, (const f16*)b, (f16*)out, rows, C1 / 8, C2 / 8);
    else if (dtype == RSVLD_BF16)
        hipLaunchKernelGGL(concat_c_kernel<bf16>, dim3(grid1d(total)), dim3(256), 0, s, (const bf16*)a, (const bf16*)b, (bf16*)out, rows, C1 / 8, C2 / 8);
    else
        return RSVLD_EINVAL;
    return rsvld_check_launch();
}
