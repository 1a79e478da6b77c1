/*
 * rsvld_hip.h — C ABI of librsvld_hip.so, the MI355X (gfx950) kernel library behind the
 * two-stage diffusion super-resolution sampler.
 *
 * The reference (Bluear7878/Remote-Sensing-Vision-Language-Diffusion-Model) has no FFI of
 * its own: every op below replaces a call the reference makes into a third-party wheel
 * (cuDNN / cuBLAS / xformers through torch).  Each entry names the reference call sites it
 * stands in for (file:line relative to the reference tree).
 *
 * Conventions
 *   - every function returns 0 on success, a negative RSVLD_E* code otherwise; nothing throws;
 *   - all pointers are DEVICE pointers owned by the caller; nothing is allocated inside;
 *     workspaces are passed in; `stream` is a hipStream_t passed as void*;
 *   - activations are NHWC ("channels last", token-major) with C % 8 == 0;
 *     `dtype` selects the 16-bit storage/MFMA operand type: 0 = fp16, 1 = bf16.
 *     Accumulation, normalisation statistics and softmax are always fp32;
 *   - stateless and re-entrant: one stream per call; no environment variables are read and no
 *     mutable global state is kept (the only process-wide objects are one-time, thread-safe
 *     registrations of kernel attributes, C++11 function-local statics);
 *   - launch plans are BATCH-INVARIANT on request: the kernel family, tile shape and split plan of a
 *     call are derived from (batch / plan_div), i.e. from ONE of `plan_div` independent work units
 *     stacked along the batch, so a unit's result is bit-identical however many units share the
 *     launch (data-parallel sharding and per-image feature-cache sub-batches rely on it).
 */
#ifndef RSVLD_HIP_H
#define RSVLD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSVLD_OK 0
#define RSVLD_EINVAL (-1)      /* bad shape / null pointer / unsupported combination */
#define RSVLD_EUNSUPPORTED (-2)
#define RSVLD_ELAUNCH (-3)     /* hipGetLastError() after launch was not hipSuccess   */

#define RSVLD_F16 0
#define RSVLD_BF16 1
#define RSVLD_F32 2   /* accepted ONLY by the *_f32 entry points at the end of this header */
#define RSVLD_SPLIT 3 /* split-operand precision on the 16-bit tilings (rsvld_conv_desc.dtype; the *_split entry points below) */
#define RSVLD_F16W2 4 /* fp16 activations x fp16 weight PAIRS [W_lo | W_hi] (rsvld_conv_desc.dtype; see below): two MFMAs per product */
#define RSVLD_F16Q8 6 /* fp16 hi x hi + the two cross terms of the split product in e4m3 on the block-scaled matrix instruction (see below) */
#define RSVLD_F16W1 5 /* fp16 activations x the SAME weights rounded to fp16, fp32 out (+ fp32 residual): one MFMA per product (see below) */

/* epilogue activations for rsvld_conv2d_nhwc */
#define RSVLD_ACT_NONE 0
#define RSVLD_ACT_SILU 1
#define RSVLD_ACT_GEGLU 2   /* weights packed value/gate interleaved; output has Cout/2 channels */

const char* rsvld_version(void);

/* ---------------------------------------------------------------------------------------
 * Implicit-GEMM convolution / linear on MFMA (v_mfma_f32_32x32x16_{f16,bf16}).
 *
 *   out[b,oy,ox,co] = alpha * ( sum_{ky,kx,ci} X[b, iy, ix, ci] * W[co, ky, kx, ci]
 *                               + bias[co] + rowvec[b,co] ) + beta * residual[b,oy,ox,co]
 *   iy = oy*stride + ky - pad_t, ix = ox*stride + kx - pad_l (zero outside the image);
 *   with `upsample` = 1 the input is read through a nearest x2 up-sampling (iy>>1, ix>>1)
 *   so the up-sampled tensor is never materialised;
 *   with x2 != NULL the input is the channel concatenation [x (Cin) | x2 (Cin2)].
 *   A Linear layer is the same call with B=1, H=1, W=rows, KH=KW=1.
 *
 * Replaces: nn.Conv2d / nn.Linear calls in models/sr3_model/sr3_modules/unet.py:59-75,81-143;
 *   sgm/modules/diffusionmodules/openaimodel.py:102-350; sgm/modules/attention.py:84-110,196-285;
 *   sgm/modules/diffusionmodules/model.py:55-148; models/modules/SR_modules.py:59-149.
 * ------------------------------------------------------------------------------------- */
typedef struct rsvld_conv_desc {
    const void* x;         /* [B,H,W,Cin]   16-bit                                  */
    const void* x2;        /* [B,H,W,Cin2]  16-bit or NULL                          */
    const void* w;         /* [Cout][KH*KW*(Cin+Cin2)] 16-bit, K-major (tap, channel) */
    const float* bias;     /* [Cout] or NULL                                        */
    const float* rowvec;   /* [B,Cout] or NULL (time / noise-level embedding)       */
    const void* residual;  /* [B,Ho,Wo,Cout_out] 16-bit or NULL                     */
    void* out;             /* [B,Ho,Wo,Cout_out] 16-bit, or fp32 when out_f32       */
    int32_t B, H, W, Cin, Cin2, Cout;
    int32_t KH, KW, stride, pad_t, pad_l, Ho, Wo;
    int32_t upsample;      /* 0 / 1                                                 */
    int32_t dtype;         /* RSVLD_F16 / RSVLD_BF16                                */
    int32_t out_f32;       /* 1: out is fp32 (16-bit dtypes: Cout <= 32 only); 2: RSVLD_SPLIT only, out is fp16 */
    int32_t act;           /* RSVLD_ACT_*                                           */
    float alpha, beta;
    int32_t rowvec_stride; /* elements between rows of rowvec; 0 = Cout                     */
    int32_t plan_div;      /* independent work units stacked along B (or along the rows of a Linear): the launch
                              plan is made for B*Ho*Wo / plan_div rows; 0 or 1 = plan for the whole call      */
    int32_t tune;          /* RSVLD_TUNE_* developer A/B overrides; 0 = the library's own choice             */
} rsvld_conv_desc;

/* rsvld_conv_desc.tune (benchmarking only; every combination computes the same function) */
#define RSVLD_TUNE_TILE_MASK 7          /* 1: 256x64, 2: 128x64, 3: 128x128, 4: 64x128 implicit-GEMM tile */
#define RSVLD_TUNE_STAGES_SHIFT 3       /* bits 3..5: LDS ring depth 2..4 (0 = per-tile default)          */
#define RSVLD_TUNE_NO_KSPLIT (1 << 6)   /* no intra-workgroup split-K variants                            */
#define RSVLD_TUNE_REG_STAGING (1 << 7) /* register-staged operands instead of LDS-DMA                    */
#define RSVLD_TUNE_HALO_NW4 (1 << 8)    /* halo conv: force the 4-wave 8x32 tile                          */
#define RSVLD_TUNE_HALO_NW8 (1 << 9)    /* halo conv: force the 8-wave 16x32 tile                         */
#define RSVLD_TUNE_NO_GEMM256 (1 << 10) /* keep large 1x1 / Linear layers on the implicit-GEMM kernel     */
#define RSVLD_TUNE_GEMM_ONE_TILE (1 << 12) /* gemm256: one tile per workgroup (rounds 2-3) instead of the persistent form          */
#define RSVLD_TUNE_F32_SPLIT (1 << 11)  /* rsvld_conv2d_nhwc_f32 only -- a precision MODE, not an A/B switch: every fp32 operand is
                                         * split into hi + lo bf16 and each product runs as three 16-bit MFMAs into the fp32
                                         * accumulator (~1e-5 relative; the "split" precision of SR_backbone.set_precision)        */

/* dtype = RSVLD_SPLIT (accepted by rsvld_conv2d_nhwc and rsvld_conv3x3_halo_nhwc): the "split" precision of
 * SR_backbone.set_precision / compute_dtype "split" as a PRODUCT path.  An fp32 activation v travels as two bf16 PLANES per row,
 *     x, x2 : [B,H,W, lo(C) | hi(C)]   hi = bf16(v), lo = bf16(v - hi)        (Cin / Cin2 stay the LOGICAL channel counts)
 * and a weight as the per-tap TRIPLE  w : [Cout][KH*KW][ W_hi(Cin+Cin2) | W_lo(Cin+Cin2) | W_hi(Cin+Cin2) ]  (rsvld_split_pack_weights),
 * so that v W = v_lo W_hi + v_hi W_lo + v_hi W_hi (the dropped lo*lo term is 2^-16 of the product) is ONE bf16 contraction over
 * 3 (Cin+Cin2) logical channels per tap whose third segment re-reads the hi planes: the 16-bit kernels run unchanged at three MFMAs
 * per fp32 product, fp32 accumulation.  residual: fp32 [.., Cout_out].  out: fp32 [.., Cout_out] when out_f32 = 1 (the residual
 * stream), bf16 planes [.., lo(Cout_out) | hi(Cout_out)] when out_f32 = 0 (a tensor that only feeds another matrix product; no
 * residual then).  The fused GroupNorm prologue of the halo kernel is not available (rsvld_groupnorm_apply_split writes planes). */
/* dtype = RSVLD_F16W2 (round 5; accepted by rsvld_conv2d_nhwc and rsvld_conv3x3_halo_nhwc): the layers of the tolerance-compliant
 * mode whose INPUT may be rounded to fp16 (measured: DESIGN.md section 4) but whose WEIGHTS may not.  x, x2 : fp16 [B,H,W,C];
 *     w : fp16 [Cout][KH*KW][ W_lo(Cin+Cin2) | W_hi(Cin+Cin2) ]   W_hi = fp16(W), W_lo = fp16(W - W_hi)   (rsvld_pack_weight_pairs)
 * so that x W = x W_lo + x W_hi is ONE fp16 contraction over 2 (Cin+Cin2) logical channels per tap whose second segment re-reads
 * the activation: two MFMAs per product instead of RSVLD_SPLIT's three, half the activation bytes, and the fused GroupNorm prologue
 * of the halo kernel stays available.  residual: of the OUTPUT's type (fp32 with out_f32 = 1: a split-precision network's residual
 * stream; fp16 with out_f32 = 0: SR3's compute dtype "w2", fp16 tensors throughout).  out: fp32 when out_f32 = 1, fp16 when out_f32 = 0.
 * RSVLD_SPLIT with out_f32 = 2 writes fp16 as well (q | k | v on their way to the 16-bit attention kernels). */
/* dtype = RSVLD_F16W1 (round 5; accepted by rsvld_conv2d_nhwc, 1x1 only): a Linear layer of a split-precision transformer block whose policy
 * rounds its WEIGHTS to fp16 as well (SplitPolicy.f16_weights "attn_out" / "ff_out": to_out of sgm/modules/attention.py:288-373, ff.net.2
 * of :250-285).  x : fp16; w : plain fp16 [Cout][Cin] (what RSVLD_F16 takes): ONE MFMA per product -- with the OUTPUT side of the
 * multi-segment family: out fp32 (out_f32 must be 1), residual fp32, out = alpha * (x W + b) + beta * residual in fp32.  The same layer with
 * a 16-bit output is plain RSVLD_F16. */
/* dtype = RSVLD_F16Q8 (round 6; accepted by rsvld_conv3x3_halo_nhwc: Cout > 64, ONE source (Cin2 = 0), Cin % 64 == 0, no fused norm, no up-sampling): the 3x3 convolutions
 * of the split-precision UNets with FEWER MATRIX CYCLES PER PRODUCT.  x w = x_hi w_hi + x_lo w_hi + x_hi w_lo; the two cross terms are 2^-11 of
 * the product when the hi parts are fp16, so their operands run as e4m3 on v_mfma_scale_f32_32x32x64_f8f6f4 (2.26 x the 16-bit FLOP rate
 * measured, profiles/r06_mfma_f8f6f4.txt): per 32 channels and tap 2 fp16 MFMAs + 1 scaled MFMA instead of 6 bf16 MFMAs.
 *     x     : [B,H,W, 4 C bytes]  row = [ fp16(x) (C) | C / 32 blocks of 64 B { P0[0:16] | P1[0:16] | P0[16:32] | P1[16:32] } ],
 *             P0 = e4m3((x - fp16 x) 2^14), P1 = e4m3(x 2^2) of the block's 32 channels  (rsvld_groupnorm_apply_split out_f32 = 3, rsvld_split_hq8)
 *     w     : [Cout][9][ fp16(w) (Ctot) | Ctot / 32 blocks likewise ], P0 = e4m3(w 2^6), P1 = e4m3((w - fp16 w) 2^18)   (rsvld_pack_weight_hq8)
 * (the 16-byte interleave lets the kernel read a block with the fragment addresses of its fp16 k-steps: lane half h gets all of P_h)
 * out fp32 (out_f32 must be 1), residual fp32, epilogue statistics as for RSVLD_SPLIT.  Measured at full network depth over 50 steps the
 * distance from the fp32 family is that of RSVLD_SPLIT (+3.5 % in the mean, profiles/r06_conv_lo8_emulation.txt). */
#define RSVLD_HQ8_SX_LO 14
#define RSVLD_HQ8_SX_HI 2
#define RSVLD_HQ8_SW_HI 6
#define RSVLD_HQ8_SW_LO 18
int rsvld_conv2d_nhwc(const rsvld_conv_desc* d, void* stream);

/* 3x3 / stride-1 / pad-1 convolution with an LDS-resident input halo patch (each activation byte crosses
 * L2->LDS once per 64-channel chunk instead of once per tap) and an OPTIONAL fused GroupNorm prologue:
 * when norm_scale_shift != NULL (fp32 [B][Cin+Cin2][2] from rsvld_groupnorm_scale_shift) the input is
 * y = act(scale*x + shift) (act = SiLU if norm_silu) applied while the patch is staged, so the
 * normalised tensor is never written to HBM.  Same descriptor / epilogue as rsvld_conv2d_nhwc.
 * rsvld_conv3x3_halo_supported tells whether a descriptor is eligible (1) or must use the gather kernel (0).
 * Replaces GN+SiLU+Conv3x3 of unet.py:81-92, openaimodel.py:263-301, model.py:128-141.
 * out_stats_partials (optional, fp32 [B][ceil(Ho/8)*ceil(Wo/32)][Cout][2]): per-tile per-channel (sum, sum of squares)
 * of the stored output, produced in the epilogue, so that the NEXT GroupNorm needs no pass over the tensor
 * (consumed by rsvld_groupnorm_scale_shift_from_partials).  Deterministic: plain stores, merged later in fp64. */
int rsvld_conv3x3_halo_supported(const rsvld_conv_desc* d);
int rsvld_conv3x3_halo_nhwc(const rsvld_conv_desc* d, const float* norm_scale_shift, int norm_silu,
                            float* out_stats_partials, void* stream);

/* ---------------------------------------------------------------------------------------
 * GroupNorm (+ optional SiLU / Swish) over NHWC, statistics in fp32, two launches
 * (deterministic partial sums -> finalize+apply).  `ws` must hold
 * rsvld_groupnorm_ws_bytes(B, HW, C, groups) bytes.
 *   y = act( (x - mean[b,g]) * rstd[b,g] * gamma[c] + beta[c] )
 * With x2 != NULL statistics and output cover the concatenation [x | x2] along C.
 * If scale1p/shift (16-bit, rows of `mod_stride` elements, 0 = C; both may point into one stacked
 * [B,HW,2C] conv output) are given: y = y*(1+scale1p) + shift  (ZeroSFT, SR_modules.py:100-106).
 *
 * Replaces: nn.GroupNorm + Swish/SiLU at sr3_modules/unet.py:81-92,114-125;
 *   GroupNorm32 (sgm/modules/diffusionmodules/util.py:273-276) in openaimodel.py:207-350;
 *   attention.Normalize (sgm/modules/attention.py:152-155); model.py:49-52;
 *   SR_modules.py:59-110 (ZeroSFT param_free_norm), :113-149 (ZeroCrossAttn norm1/2).
 * ------------------------------------------------------------------------------------- */
int64_t rsvld_groupnorm_ws_bytes(int B, int HW, int C, int groups);
int rsvld_groupnorm_nhwc(const void* x, const void* x2, void* y,
                         const float* gamma, const float* beta,
                         const void* mod_scale1p, const void* mod_shift, int mod_stride,
                         int B, int HW, int C1, int C2, int groups, float eps,
                         int silu, int dtype, void* ws, void* stream);
/* statistics only: writes mean_var[B*groups*2] fp32 = (mean, biased variance) per (image, group);
 * used on its own by the tiled VAE's cross-tile GroupNorm (utils/tilevae.py:511-521,599-674). */
int rsvld_groupnorm_stats(const void* x, const void* x2, float* mean_var,
                          int B, int HW, int C1, int C2, int groups,
                          int dtype, void* ws, void* stream);
/* apply with caller-provided statistics (mean_var[B*groups*2]) */
int rsvld_groupnorm_apply(const void* x, const void* x2, void* y, const float* mean_var,
                          const float* gamma, const float* beta,
                          const void* mod_scale1p, const void* mod_shift, int mod_stride,
                          int B, int HW, int C1, int C2, int groups, float eps,
                          int silu, int dtype, void* stream);

/* statistics + per-(image, channel) affine of a GroupNorm, for the fused conv prologue:
 * scale_shift[b][c] = (gamma[c]*rstd[b,g], beta[c] - mean[b,g]*gamma[c]*rstd[b,g]), fp32 [B][C1+C2][2] */
int rsvld_groupnorm_scale_shift(const void* x, const void* x2, const float* gamma, const float* beta,
                                float* scale_shift, int B, int HW, int C1, int C2, int groups, float eps,
                                int dtype, void* ws, void* stream);

/* same affine from per-channel partial sums written by conv epilogues (one or two producers: [x | x2]) */
int rsvld_groupnorm_scale_shift_from_partials(const float* part1, int ntiles1, int C1,
                                              const float* part2, int ntiles2, int C2,
                                              const float* gamma, const float* beta, float* scale_shift,
                                              int B, int HW, int groups, float eps, void* stream);

/* LayerNorm over the last dim of [rows, C] (eps 1e-5, affine).  attention.py:376-486 */
int rsvld_layernorm(const void* x, void* y, const float* gamma, const float* beta,
                    int64_t rows, int C, float eps, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------
 * Flash-style attention  out = softmax(scale * Q K^T) V, online softmax in fp32,
 * scores never materialised.  q/k/v/out are token-major: element (b, n, h, d) lives at
 * base + b*batch_stride + n*tok_stride + h*D + d (strides in ELEMENTS), which lets q, k, v
 * alias one fused qkv projection output.
 *   D = 512 : single-head (SR3 SelfAttention, VAE mid attention)
 *   D = 64  : multi-head (sgm CrossAttention / MemoryEfficientCrossAttention, ZeroCrossAttn)
 *
 * Replaces: the two einsums + softmax with the materialised [B,1,h,w,h,w] tensor at
 *   sr3_modules/unet.py:133-141; xformers.ops.memory_efficient_attention at
 *   sgm/modules/attention.py:357-359 and F.scaled_dot_product_attention at :273-277;
 *   sgm/modules/diffusionmodules/model.py:187-189,246-248; utils/tilevae.py:335-336.
 * ------------------------------------------------------------------------------------- */
int rsvld_attention(const void* q, const void* k, const void* v, void* out,
                    int B, int heads, int Nq, int Nk, int D,
                    int64_t q_batch_stride, int64_t q_tok_stride,
                    int64_t k_batch_stride, int64_t k_tok_stride,
                    int64_t v_batch_stride, int64_t v_tok_stride,
                    int64_t o_batch_stride, int64_t o_tok_stride,
                    float scale, int dtype, int plan_div, void* ws, void* stream);
/* The same call with a developer A/B override (tests, tools): which of the three d = 64 kernels runs.  They agree bit for
 * bit on every shape (tests/test_gpu_kernels.py), so the library's own choice (tune = 0: by grid size) is a speed decision only:
 *   attn_d64b  four waves per SIMD, 128 / 256 query rows per workgroup (short sequences, small grids);
 *   attn_d64c  "ping-pong": 8 waves x 64 query rows, matrix and vector segments of the two waves of a SIMD in anti-phase
 *              (grids of >= 1024 workgroups: the Stage-2 self-attention of the headline, +2-5 %);
 *   attn_d64p  "pipelined": every wave mixes its MFMAs with chunks of the next tile's softmax (experiment, 12 % slower). */
#define RSVLD_ATTN_D64_FOUR_WAVE 1
#define RSVLD_ATTN_D64_PINGPONG 2
#define RSVLD_ATTN_D64_PIPELINED 3
int rsvld_attention_tuned(const void* q, const void* k, const void* v, void* out,
                          int B, int heads, int Nq, int Nk, int D,
                          int64_t q_batch_stride, int64_t q_tok_stride,
                          int64_t k_batch_stride, int64_t k_tok_stride,
                          int64_t v_batch_stride, int64_t v_tok_stride,
                          int64_t o_batch_stride, int64_t o_tok_stride,
                          float scale, int dtype, int plan_div, void* ws, void* stream, int tune);
/* bytes of `ws` the call above needs (0 = none, ws may be NULL): the D = 512 kernel splits the keys over
 * workgroups when the query tiles of B / plan_div batch rows alone cannot fill the chip and merges the partial
 * results from ws (plan_div: independent units stacked along B, see the conventions; 0 or 1 = plan on B). */
int64_t rsvld_attention_ws_bytes(int B, int heads, int Nq, int Nk, int D, int plan_div);

/* ---------------------------------------------------------------------------------------
 * Small dense layers on embeddings (rows <= 64): y = act_out( W * act_in(x) + b ), fp32.
 *   act: 0 none, 1 SiLU.   W is [out_f, in_f] fp32 row-major (torch nn.Linear layout).
 * Replaces: noise_level_mlp / FeatureWiseAffine (sr3_modules/unet.py:35-51,180-185);
 *   time_embed / label_emb / emb_layers (openaimodel.py:246-255,657-664,683-691).
 * ------------------------------------------------------------------------------------- */
int rsvld_linear_small_f32(const float* x, const float* w, const float* b, float* y,
                           int rows, int in_f, int out_f, int act_in, int act_out, void* stream);

/* sinusoidal embeddings.
 *   kind 0: SR3 PositionalEncoding  [sin(l*f_k) | cos(l*f_k)], f_k = exp(-ln(1e4)*k/(dim/2))
 *           (sr3_modules/unet.py:19-32)
 *   kind 1: sgm timestep_embedding  [cos(t*f_k) | sin(t*f_k)], f_k = exp(-ln(1e4)*k/(dim/2))
 *           (sgm/modules/diffusionmodules/util.py:206-230)                                  */
int rsvld_sinusoidal_embedding(const float* t, float* out, int rows, int dim, int kind, void* stream);

/* ---------------------------------------------------------------------------------------
 * Layout / element-wise kernels (HBM-bound).
 * ------------------------------------------------------------------------------------- */
/* fp32 NCHW [B,C,H,W] * scale -> 16-bit NHWC [B,H,W,Cdst] at channel offset c_off (pad lanes untouched
 * unless zero_pad=1, which zeroes channels outside [c_off, c_off+C)).  `scale` carries the denoiser's
 * input scaling c_in (denoiser.py:72-76). */
int rsvld_nchw_f32_to_nhwc(const float* src, void* dst, int B, int C, int H, int W,
                           int Cdst, int c_off, int zero_pad, float scale, int dtype, void* stream);
/* 16-bit (or fp32 when src_f32) NHWC [B,H,W,Csrc] channels [c_off,c_off+C) -> fp32 NCHW */
int rsvld_nhwc_to_nchw_f32(const void* src, float* dst, int B, int C, int H, int W,
                           int Csrc, int c_off, int src_f32, int dtype, void* stream);
/* out = a + b  /  out = a*sa + b*sb  on 16-bit tensors of n elements (n % 8 == 0) */
int rsvld_axpby(const void* a, const void* b, void* out, int64_t n, float sa, float sb,
                int dtype, void* stream);
/* x * gelu(gate), exact erf GELU: in [rows, 2*C] -> out [rows, C]  (attention.py:84-96) */
int rsvld_geglu(const void* in, void* out, int64_t rows, int C, int dtype, void* stream);

/* SR3 ancestral DDPM step on fp32 NCHW tensors, eps read from the UNet's fp32 NHWC output
 * (C padded to eps_c):  x0 = clamp(c_recip*x - c_recipm1*eps, -1, 1);
 * x_out = coef1*x0 + coef2*x + sigma*noise   (noise may be NULL when sigma == 0).
 * Replaces sr3_modules/diffusion.py:142-175. */
int rsvld_ddpm_step(const float* x, const float* eps_nhwc, const float* noise, float* x_out,
                    int B, int C, int H, int W, int eps_c,
                    float c_recip, float c_recipm1, float coef1, float coef2, float sigma,
                    int clip, void* stream);

/* ---------------------------------------------------------------------------------------
 * Stage-2 sampler (RestoreEDMSampler + DiscreteDenoiserWithControl + LinearCFG), fp32 NCHW state.
 * ------------------------------------------------------------------------------------- */
/* out (fp32 NCHW) = net_out_nhwc (fp32 NHWC [B,H,W,c_pad]) * c_out + input (fp32 NCHW) * c_skip.
 * Replaces denoiser.py:77-78. */
int rsvld_denoiser_out(const float* net_out_nhwc, const float* input, float* out,
                       int B, int C, int H, int W, int c_pad, float c_out, float c_skip, void* stream);
/* out = a + w*(b-a): CFG combine x_u + s*(x_c - x_u) (guiders.py:58-63, sampling_utils.py:7-9) */
int rsvld_lerp_f32(const float* a, const float* b, float* out, int64_t n, float w, void* stream);
/* out = x + s*y (x == NULL: out = s*y): churn noise x + eps*s_noise*sqrt(sigma_hat^2 - sigma^2)
 * (sampling.py:600-606) and the initial x *= sqrt(1 + sigma_0^2) (:49) */
int rsvld_axpy_f32(const float* x, const float* y, float* out, int64_t n, float s, void* stream);
/* dn = denoised - (denoised - x_center)*restore_w (x_center may be NULL); x_out = x_hat +
 * (x_hat - dn)/sigma_hat * dt.   Replaces sampling.py:614-620. */
int rsvld_euler_step(const float* x_hat, const float* denoised, const float* x_center, float* x_out,
                     int64_t n, float restore_w, float sigma_hat, float dt, void* stream);
/* Latent-tile blending of TiledRestoreEDMSampler: fp32 NCHW acc / cnt [B,C,H,W]; tile [B,C,th,tw] is the sampler step's
 * result for the window (y0, x0); weights [th,tw] the Gaussian tile mask.  acc[window] += tile * weights,
 * cnt[window] += weights (sampling.py:733-734); finish: out = acc / cnt (:735). */
int rsvld_tile_blend_accumulate(float* acc, float* cnt, const float* tile, const float* weights,
                                int B, int C, int H, int W, int y0, int x0, int th, int tw, void* stream);
int rsvld_tile_blend_finish(const float* acc, const float* cnt, float* out, int64_t n, void* stream);
/* First-block-cache similarity (models/modules/DFBCache.py:98-112): out[row] = (sum|a-b|, sum|a|) per
 * row of two 16-bit [rows, n_per_row] tensors; fp32 partials, fp64 merge, deterministic. */
int64_t rsvld_absdiff_ws_bytes(int rows, int64_t n_per_row);
int rsvld_absdiff_sums(const void* a, const void* b, float* out, int rows, int64_t n_per_row,
                       int dtype, void* ws, void* stream);

/* DiagonalGaussianDistribution (sgm/modules/distributions/distributions.py:24-41,71-72) on NHWC moments
 * [B,H,W,m_c] (mean | logvar): z = (mean + exp(0.5*clamp(logvar,-30,20))*noise)*scale, or mode()*scale
 * when noise == NULL; z is fp32 NCHW [B,C,H,W]. */
int rsvld_gaussian_sample(const void* moments_nhwc, const float* noise, float* z,
                          int B, int C, int H, int W, int m_c, float scale, int src_f32,
                          int dtype, void* stream);

/* Colour fix (utils/colorfix.py).  wavelet_blur: 3x3 [1,2,1]x[1,2,1]/16 depthwise blur, dilation =
 * radius, replicate padding, on `planes` fp32 H x W planes; if high_accum != NULL it also accumulates
 * high_accum += img - low (one level of wavelet_decomposition, :94-106). */
int rsvld_wavelet_blur(const float* img, float* low, float* high_accum, int planes, int H, int W,
                       int radius, void* stream);
int rsvld_add_f32(const float* a, const float* b, float* out, int64_t n, void* stream);
/* adaptive_instance_normalization (:59-71); ws_stats holds 4*planes floats */
int rsvld_adain(const float* content, const float* style, float* out, float* ws_stats,
                int planes, int64_t HW, void* stream);

/* channel concatenation of two 16-bit NHWC tensors [rows,C1] | [rows,C2] -> [rows,C1+C2] */
int rsvld_concat_c(const void* a, const void* b, void* out, int64_t rows, int C1, int C2,
                   int dtype, void* stream);

/* ---------------------------------------------------------------------------------------
 * fp32-operand family: the VAE under ``ae_dtype: fp32`` and the UNet / ControlNet under ``diffusion_dtype: fp32``
 * (models/SR_model.py:28-33 runs them without autocast; sgm/modules/diffusionmodules/model.py:55-262,482-743,
 * openaimodel.py:102-350, sgm/modules/attention.py:84-486, models/modules/SR_modules.py:59-149).  fp32 NHWC activations (C % 8 == 0), fp32 K-major weights,
 * fp32 MFMA (v_mfma_f32_32x32x2_f32).  Accuracy mode: simple LDS tiling, 1/16 of the 16-bit matrix rate.
 * ------------------------------------------------------------------------------------- */
/* rsvld_conv2d_nhwc with dtype = RSVLD_F32: x, x2, w, residual, out are fp32 (same formula, same epilogues incl. GEGLU);
 * out_f32, plan_div and tune are ignored (the plan never depends on the batch). */
int rsvld_conv2d_nhwc_f32(const rsvld_conv_desc* d, void* stream);
/* GroupNorm on fp32 NHWC [B,HW,C]: statistics (mean, biased variance) per (image, group) through fp64 partial sums
 * merged in a fixed order; apply = (x - mean) * rsqrt(var + eps) * gamma + beta (+ SiLU). */
int64_t rsvld_groupnorm_f32_ws_bytes(int B, int HW, int C, int groups);
int rsvld_groupnorm_stats_f32(const float* x, float* mean_var, int B, int HW, int C, int groups, void* ws, void* stream);
int rsvld_groupnorm_apply_f32(const float* x, float* y, const float* mean_var, const float* gamma, const float* beta,
                              const float* mod_scale1p, const float* mod_shift, int mod_stride,
                              int B, int HW, int C, int groups, float eps, int silu, void* stream);
/* rsvld_attention on fp32 tensors (same addressing; D % 32 == 0, D <= 512; scores never materialised) */
int rsvld_attention_f32(const float* q, const float* k, const float* v, float* out,
                        int B, int heads, int Nq, int Nk, int D,
                        int64_t q_batch_stride, int64_t q_tok_stride,
                        int64_t k_batch_stride, int64_t k_tok_stride,
                        int64_t v_batch_stride, int64_t v_tok_stride,
                        int64_t o_batch_stride, int64_t o_tok_stride,
                        float scale, void* stream);
/* rsvld_attention_f32 with split operands (hi + lo bf16, three 16-bit MFMAs per product; see RSVLD_TUNE_F32_SPLIT) */
int rsvld_attention_f32_split(const float* q, const float* k, const float* v, float* out,
                              int B, int heads, int Nq, int Nk, int D,
                              int64_t q_batch_stride, int64_t q_tok_stride,
                              int64_t k_batch_stride, int64_t k_tok_stride,
                              int64_t v_batch_stride, int64_t v_tok_stride,
                              int64_t o_batch_stride, int64_t o_tok_stride,
                              float scale, void* stream);
/* fp32 forms of rsvld_layernorm, rsvld_concat_c, rsvld_axpby and rsvld_absdiff_sums (the Stage-2 networks under
 * ``diffusion_dtype: fp32``; sgm/modules/attention.py:376-486, models/modules/SR_modules.py:59-149, DFBCache.py:98-112) */
int rsvld_layernorm_f32(const float* x, float* y, const float* gamma, const float* beta, int64_t rows, int C, float eps,
                        void* stream);
int rsvld_concat_c_f32(const float* a, const float* b, float* out, int64_t rows, int C1, int C2, void* stream);
int rsvld_axpby_f32(const float* a, const float* b, float* out, int64_t n, float sa, float sb, void* stream);
int rsvld_absdiff_sums_f32(const float* a, const float* b, float* out, int rows, int64_t n_per_row, void* stream);
/* rsvld_nchw_f32_to_nhwc with an fp32 destination */
int rsvld_nchw_f32_to_nhwc_f32(const float* src, float* dst, int B, int C, int H, int W,
                               int Cdst, int c_off, int zero_pad, float scale, void* stream);

/* Weight-streaming matrix-vector product of the caption pass's token loop: y[n] = sum_k w[n][k] x[k] (+ bias[n]); w [N][K]
 * row-major, x [K], bias [N] or NULL, y [N], all 16-bit (dtype), fp32 accumulation; K % 8 == 0, K <= 32768.
 * Replaces torch.nn.functional.linear with ONE activation row in the Llama decode step behind models/util.py:17-66
 * (llava/model/language_model/llava_llama.py:118-137): q|k|v, o, gate|up, down projections and lm_head. */
int rsvld_gemv(const void* w, const void* x, const void* bias, void* y, int N, int K, int dtype, void* stream);
/* rsvld_gemv with the element-wise neighbours of a Llama decoder layer's decode step folded in (the same reference call sites; the unfused
 * sequence is ~26 launches per layer, the token loop is launch-bound): norm_w != NULL: x <- RMSNorm(x) * norm_w (LlamaRMSNorm in front of
 * q|k|v, gate|up and lm_head; norm_eps); glu: x holds 2 K elements [gate | up] and the product runs on silu(gate) * up (SwiGLU in front of
 * down_proj); residual != NULL [N]: y <- residual + (w x + bias) (the layer's two residual additions).  Roundings to the 16-bit type where
 * the unfused sequence has them. */
int rsvld_gemv_fused(const void* w, const void* x, const void* bias, const void* norm_w, float norm_eps, const void* residual, int glu,
                     void* y, int N, int K, int dtype, void* stream);
/* One decode step of grouped-query attention over a static cache (LlamaAttention.forward with ONE new token, llava_llama.py:118-137 ->
 * transformers' Llama decode): qkv [ (n_q + 2 n_kv) x head_dim ] = the new token's q | k | v rows; cos, sin [head_dim] (half-split rotary
 * embedding, 16-bit); *pos (DEVICE int64: the step is replayed from a hipGraph) = the token's position; kcache / vcache
 * [n_kv][max_len][head_dim] are UPDATED at *pos; out [n_q x head_dim] = softmax(q K^T scale) V over positions <= *pos.  head_dim = 128,
 * n_q / n_kv <= 8.  ws: rsvld_llama_decode_attention_ws_bytes(n_q, n_kv, max_len) bytes of scratch (no initial state).  *pos outside
 * [0, max_len) is clamped into it by the kernel (the host cannot see a device scalar): nothing is written past the cache. */
size_t rsvld_llama_decode_attention_ws_bytes(int n_q, int n_kv, int max_len);
int rsvld_llama_decode_attention(const void* qkv, const void* cos, const void* sin, const int64_t* pos, void* kcache, void* vcache, void* out,
                                 float* ws, int n_q, int n_kv, int head_dim, int max_len, float scale, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------
 * Split-operand product path (round 4): the producers / consumers of bf16 PLANES around the RSVLD_SPLIT convolutions.
 * "planes [rows][2C]" = per row lo(C) | hi(C) of an fp32 [rows][C] tensor (see RSVLD_SPLIT above); C % 8 == 0.
 * Replaces the same reference calls as the fp32-operand family (models/SR_model.py:28-33,57-85 without autocast;
 * sgm/modules/diffusionmodules/wrappers.py:84-110; models/sr3_model/sr3_modules/unet.py:81-143).
 * ------------------------------------------------------------------------------------- */
/* fp32 [rows][C] -> planes [rows][2C], and back (x = float(hi) + float(lo)) */
int rsvld_split_planes(const float* x, void* planes, int64_t rows, int C, void* stream);
int rsvld_merge_planes(const void* planes, float* x, int64_t rows, int C, void* stream);
/* The "split, attention in fp16" composition (rsvld_amd.ops.SPLIT_ATTN = "f16"): the operands of an attention leave the planes as fp16
 * (x = fp16(float(hi) + float(lo)); `ld` = row stride and `plane_stride` = distance lo -> hi of the source, in elements: a channel
 * slice of a fused qkv planes tensor is read in place) for rsvld_attention, and its fp16 output returns as planes (exact).
 * Replaces xformers.ops.memory_efficient_attention at sgm/modules/attention.py:357-359 and the einsum + softmax of
 * models/sr3_model/sr3_modules/unet.py:133-141 when every other product of the network runs in the split precision. */
int rsvld_planes_to_f16(const void* planes, int64_t ld, int64_t plane_stride, void* out_f16, int64_t out_ld, int64_t rows, int C, void* stream);
int rsvld_f16_to_planes(const void* x_f16, int64_t ld, void* planes, int64_t rows, int C, void* stream);
/* fp32 K-major weights [Cout][taps][Ctot] -> bf16 triples [Cout][taps][W_hi(Ctot) | W_lo(Ctot) | W_hi(Ctot)] */
int rsvld_split_pack_weights(const float* w, void* w3, int64_t Cout, int taps, int Ctot, void* stream);
/* fp32 K-major weights [Cout][taps][Ctot] -> fp16 pairs [Cout][taps][W_lo(Ctot) | W_hi(Ctot)], W_hi = fp16(W), W_lo = fp16(W - W_hi): the
 * weights of dtype RSVLD_F16W2 (same nn.Conv2d / nn.Linear call sites as rsvld_conv2d_nhwc) */
int rsvld_pack_weight_pairs(const float* w, void* w2, int64_t Cout, int taps, int Ctot, void* stream);
/* fp32 K-major weights [Cout][taps][Ctot] (Ctot % 32 == 0) -> the rows of dtype RSVLD_F16Q8: [Cout][taps][ fp16(w) (Ctot) | Ctot / 32 blocks
 * of P0 = e4m3(w 2^RSVLD_HQ8_SW_HI), P1 = e4m3((w - fp16 w) 2^RSVLD_HQ8_SW_LO) in 16-byte pieces ]: 4 Ctot bytes per tap (nn.Conv2d 3x3 call sites of
 * sgm/modules/diffusionmodules/openaimodel.py:207-350) */
int rsvld_pack_weight_hq8(const float* w, void* whq, int64_t Cout, int taps, int Ctot, void* stream);
/* fp32 rows [rows][C] (C % 32 == 0) -> RSVLD_F16Q8 activation rows (4 C bytes each; saturating at the fp16 / e4m3 ranges, NaN kept in the fp16 part) */
int rsvld_split_hq8(const float* x, void* out, int64_t rows, int C, void* stream);
/* planes [rows][lo(C) | hi(C)] (row stride ld elements, ld >= 2C) -> the TRANSPOSED triple [C][V_hi^T(rows_p) | V_lo^T(rows_p) | V_hi^T(rows_p)]
 * (rows_p = rows padded with zeros to a multiple of 8): the "weights" of the P V product of an attention run as two split GEMMs */
int rsvld_planes_transpose_triple(const void* planes, void* w3, int64_t rows, int64_t rows_p, int C, int64_t ld, void* stream);
/* planes [rows][lo(C) | hi(C)] -> row-wise triple [rows_p][hi(C) | lo(C) | hi(C)] (rows >= rows zero): the "weights" of Q K^T */
int rsvld_planes_to_triple(const void* planes, void* w3, int64_t rows, int64_t rows_p, int C, int64_t ld, void* stream);
/* row softmax of fp32 scores s [rows][ld] over the first `cols` columns, p = softmax(scale * s) -> planes [rows][lo(cols_p) | hi(cols_p)]
 * (cols_p = cols padded to a multiple of 8; pad columns are written as zeros).  One wave per row, two passes. */
int rsvld_softmax_rows_split(const float* s, void* p_planes, int64_t rows, int cols, int cols_p, int64_t ld, float scale, void* stream);
/* GroupNorm statistics of fp32 NHWC [x | x2] -> the per-(image, channel) affine (scale, shift) fp32 [B][C1+C2][2]; ws holds
 * rsvld_groupnorm_ws_bytes(B, HW, C1+C2, groups) bytes.  fp32 partial sums per row chunk, fp64 merge in a fixed order. */
int rsvld_groupnorm_scale_shift_f32(const float* x, const float* x2, const float* gamma, const float* beta, float* scale_shift,
                                    int B, int HW, int C1, int C2, int groups, float eps, void* ws, void* stream);
/* statistics only, (mean, biased variance) fp32 [B][groups][2], through the same coalesced pass (the tiled VAE merges them across
 * tiles, utils/tilevae.py:629-648), and the affine of SUPPLIED statistics */
int rsvld_groupnorm_stats_f32_fast(const float* x, const float* x2, float* mean_var, int B, int HW, int C1, int C2, int groups,
                                   void* ws, void* stream);
int rsvld_groupnorm_scale_shift_from_stats(const float* mean_var, const float* gamma, const float* beta, float* scale_shift,
                                           int B, int C, int groups, float eps, void* stream);
/* y = act(scale * [x | x2] + shift) [* (1 + mod_scale1p) + mod_shift] from fp32 NHWC inputs; out: planes [B,HW, 2(C1+C2)]
 * (out_f32 = 0), fp32 [B,HW,C1+C2] (out_f32 = 1), fp16 [B,HW,C1+C2] (out_f32 = 2: the input of an RSVLD_F16W2 layer) or RSVLD_F16Q8
 * rows [B,HW, 4 (C1+C2) bytes] (out_f32 = 3; (C1+C2) % 32 == 0).
 * mod_* fp32 with row stride mod_stride (ZeroSFT, SR_modules.py:100-106). */
int rsvld_groupnorm_apply_split(const float* x, const float* x2, void* out, const float* scale_shift,
                                const float* mod_scale1p, const float* mod_shift, int mod_stride,
                                int B, int HW, int C1, int C2, int silu, int out_f32, void* stream);
/* LayerNorm of fp32 rows -> planes [rows][2C] (out_f32 = 0), fp32 (out_f32 = 1) or fp16 (out_f32 = 2); C % 8 == 0, C <= 4096 */
int rsvld_layernorm_split(const float* x, void* out, const float* gamma, const float* beta, int64_t rows, int C, float eps,
                          int out_f32, void* stream);
/* Flash attention on planes, D = 64 (sgm CrossAttention at sgm/modules/attention.py:357-359 under diffusion_dtype "split"):
 * q / k / v point at the LO plane of element (b, n, h, d) = base + b*batch_stride + n*tok_stride + h*64 + d, the HI plane sits
 * *_plane elements further; three bf16 MFMAs per product in both contractions, fp32 softmax.  out: planes (o_plane = distance of
 * its hi plane, out_f32 = 0) or fp32 (out_f32 = 1).  out_f32 bit 2 (value 4, developer override like rsvld_attention_tuned): run the
 * ping-pong form of the kernel (an experiment: bit-identical, 10-13 % slower). */
int rsvld_attention_split_d64(const void* q, const void* k, const void* v, void* out, int B, int heads, int Nq, int Nk,
                              int64_t q_batch_stride, int64_t q_tok_stride, int64_t q_plane,
                              int64_t k_batch_stride, int64_t k_tok_stride, int64_t k_plane,
                              int64_t v_batch_stride, int64_t v_tok_stride, int64_t v_plane,
                              int64_t o_batch_stride, int64_t o_tok_stride, int64_t o_plane,
                              float scale, int out_f32, void* stream);
/* The same for ONE head of D = 512 whose keys and values are the SAME planes tensor x (SR3 SelfAttention after the pack-time
 * re-association, sr3_modules/unet.py:114-143): two waves share 32 query rows, each owning half of the head dimension. */
int rsvld_attention_split_d512_shared(const void* q, const void* x, void* out, int B, int Nq, int Nk,
                                      int64_t q_batch_stride, int64_t q_tok_stride, int64_t q_plane,
                                      int64_t x_batch_stride, int64_t x_tok_stride, int64_t x_plane,
                                      int64_t o_batch_stride, int64_t o_tok_stride, int64_t o_plane,
                                      float scale, int out_f32, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RSVLD_HIP_H */
