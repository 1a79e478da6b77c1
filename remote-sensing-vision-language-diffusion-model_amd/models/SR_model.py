"""Stage-2 driver: ``SR_backbone`` (reference: models/SR_model.py:17-298).

Same constructor (yaml ``model.params``), same public methods — ``just_sampling``,
``encode_first_stage``, ``encode_first_stage_with_denoise``, ``decode_first_stage``,
``batchify_denoise``, ``init_tile_vae``, ``prepare_condition`` — and the same order of random draws
in ``just_sampling``: posterior sample of ``z_stage1`` (CPU generator, distributions.py:37-41), then
``randn_like(_z)`` (device generator, :265), then one draw per sampler step.  ``noise_source="cpu"``
draws the device-side noises with the CPU generator instead, which makes a run comparable with the
reference's CPU path for a fixed ``torch.manual_seed``.
"""
import copy

import torch

from ..sgm.models.diffusion import DiffusionEngine
from ..sgm.util import AttrDict, instantiate_from_config
from ..utils.colorfix import adaptive_instance_normalization, wavelet_reconstruction
from .modules.DFBCache import MyCacheContext, cache_context


class SR_backbone(DiffusionEngine):
    def __init__(self, control_stage_config, ae_dtype="fp32", diffusion_dtype="fp32", p_p="", n_p="", *args, **kwargs):
        kwargs = {k: (AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v) for k, v in kwargs.items()}
        super().__init__(*args, **kwargs)
        control_model = instantiate_from_config(control_stage_config)
        self.model.load_control_model(control_model)
        self.first_stage_model.denoise_encoder = copy.deepcopy(self.first_stage_model.encoder)
        self.sampler_config = kwargs["sampler_config"]

        self.set_precision(ae_dtype, diffusion_dtype)
        self.p_p, self.n_p = p_p, n_p
        self.noise_source = "device"
        self.upscale, self.min_size = 1, 256
        self._measure = None   # (stamp, max_steps) while bench.py measures (rsvld_amd.measure.hooks), else None

    def set_precision(self, ae_dtype, diffusion_dtype, policy=None, ae_policy=None):
        """The reference fixes both in the constructor (SR_model.py:28-33); here they can also be switched on a loaded model.
        ``policy`` / ``ae_policy`` (``ops.SplitPolicy``; "split" only): which layer inputs of the UNet + ControlNet / of the VAE are
        handed over in fp16 inside the split precision; default ``ops.UNET_POLICY`` (attention operands, to_out and FeedForward
        inputs) / ``ops.VAE_POLICY`` (none).  An explicit argument, never the environment; ``precision_key()`` names the composition.
        "fp32" (either one) runs that network on the fp32-operand kernel family (csrc/f32.hip) -- the reference without
        autocast, i.e. what its CPU path computes; "bf16" / "fp16" run the fast 16-bit kernels (fp32 accumulation);
        "split" (an addition) keeps fp32 tensors and fp32 arithmetic everywhere except the matrix products, whose operands are
        split into hi + lo bf16 with three 16-bit MFMAs per product: ~1e-5 relative per product, between the two."""
        assert (ae_dtype in ["fp32", "split", "fp16", "bf16"]) and (diffusion_dtype in ["fp32", "split", "fp16", "bf16"])
        if ae_dtype == "fp16":
            raise RuntimeError("fp16 cause NaN in AE")
        self.ae_dtype = {"fp32": torch.float32, "split": torch.float32, "bf16": torch.bfloat16}[ae_dtype]
        self.model.dtype = {"fp32": torch.float32, "split": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[diffusion_dtype]
        from .. import ops
        for pol in (policy, ae_policy):
            if pol is not None and not isinstance(pol, ops.SplitPolicy):
                raise TypeError("set_precision: policy must be an rsvld_amd.ops.SplitPolicy")
        self.model.split = (policy or ops.UNET_POLICY) if diffusion_dtype == "split" else None
        self.first_stage_model.split = (ae_policy or ops.VAE_POLICY) if ae_dtype == "split" else None
        self.first_stage_model.set_compute_dtype(self.ae_dtype)
        self._precision_names = (ae_dtype, diffusion_dtype)

    def precision_key(self):
        """(ae_dtype, diffusion_dtype, VAE policy, UNet policy) -- what a captured graph or a bench line has to name."""
        ae, df = self._precision_names
        fs, md = self.first_stage_model.split, self.model.split
        return (ae, df, None if fs is None else fs.key(), None if md is None else md.key())

    # ---- first stage ------------------------------------------------------------------------
    @torch.no_grad()
    def encode_first_stage(self, x):
        """image fp32 NCHW -> sampled latent * scale_factor, fp32 NCHW (:57-62)."""
        from ..sgm.modules.distributions.distributions import DiagonalGaussianDistribution
        fs = self.first_stage_model
        post = DiagonalGaussianDistribution(fs.moments(x), channels=fs.embed_dim)
        B, H, W, _ = post.parameters.shape
        # the reference draws this one with the CPU generator whatever the device (distributions.py:37-41)
        return post.sample(self.scale_factor, noise=self._posterior_noise((B, fs.embed_dim, H, W)))

    @torch.no_grad()
    def encode_first_stage_with_denoise(self, x, use_sample=True, is_stage1=False):
        """denoise_encoder -> quant_conv -> posterior sample()/mode() * scale_factor (:64-78)."""
        fs = self.first_stage_model
        if is_stage1:
            raise NotImplementedError("denoise_encoder_s1 does not exist in the reference model either (:69)")
        from ..sgm.modules.distributions.distributions import DiagonalGaussianDistribution
        post = DiagonalGaussianDistribution(fs.moments(x, encoder=fs.denoise_encoder), channels=fs.embed_dim)
        return post.sample(self.scale_factor) if use_sample else post.mode(self.scale_factor)

    @torch.no_grad()
    def decode_first_stage(self, z):
        from .. import ops
        return self.first_stage_model.decode(ops.axpy_f32(None, z.float().contiguous(), 1.0 / self.scale_factor)).float()

    @torch.no_grad()
    def batchify_denoise(self, x, is_stage1=False):
        return self.decode_first_stage(self.encode_first_stage_with_denoise(x, use_sample=False, is_stage1=is_stage1))

    def _encode_center(self, x_stage1, needed):
        """``z_stage1 = encode_first_stage(x_stage1)`` (:246) is read by the sampler only while the restoration pull is on
        (``restore_cfg > 0``: sampling.py:614-616); with it off (infer.py's default ``s_stage1 = -1``) the encoder pass is dead work.
        Skipped then -- but its posterior noise is still drawn, so that every later random draw is the one the reference makes."""
        if needed:
            return self.encode_first_stage(x_stage1)
        B, _, H, W = x_stage1.shape
        self._posterior_noise((B, self.first_stage_model.embed_dim, H // 8, W // 8))
        return None

    @torch.no_grad()
    def vae_front(self, x, num_samples=1, restoration_scale=4.0):
        """The three VAE passes that open ``just_sampling`` (:236-246: denoise-encode, decode, re-encode) -> ``(_z, x_stage1, z_stage1)``
        for ``just_sampling(..., vae_front=...)``.  They depend on the image only, not on the caption: a caller may issue them on a
        second HIP stream while the caption pass (a weight-streaming token loop that leaves the matrix pipes idle) runs on the first.
        Same kernels, same order of random draws (the posterior sample is the only one; the caption samples inside a forked generator)."""
        from .. import ops
        x = x.float().contiguous()
        if num_samples > 1:
            x = x.repeat(num_samples, 1, 1, 1)
        with ops.plan_units(len(x)):
            _z = self.encode_first_stage_with_denoise(x, use_sample=False)
            x_stage1 = self.decode_first_stage(_z)
            z_stage1 = self._encode_center(x_stage1, restoration_scale > 0)
        return _z, x_stage1, z_stage1

    def init_tile_vae(self, encoder_tile_size=512, decoder_tile_size=64):
        from ..utils.tilevae import VAEHook
        fs = self.first_stage_model
        for net, size, dec in ((fs.denoise_encoder, encoder_tile_size, False), (fs.encoder, encoder_tile_size, False),
                               (fs.decoder, decoder_tile_size, True)):
            net.original_forward = net.forward
            net.forward = VAEHook(net, size, is_decoder=dec, fast_decoder=False, fast_encoder=False, color_fix=False,
                                  to_gpu=True)

    # ---- conditioning -----------------------------------------------------------------------
    def prepare_condition(self, _z, p, p_p, n_p, N):
        batch = {
            "original_size_as_tuple": torch.tensor([1024, 1024]).repeat(N, 1).to(_z.device),
            "crop_coords_top_left": torch.tensor([0, 0]).repeat(N, 1).to(_z.device),
            "target_size_as_tuple": torch.tensor([1024, 1024]).repeat(N, 1).to(_z.device),
            "aesthetic_score": torch.tensor([9.0]).repeat(N, 1).to(_z.device),
            "control": _z,
        }
        batch_uc = copy.copy(batch)
        batch_uc["txt"] = [n_p for _ in p]
        if isinstance(p[0], list):
            raise NotImplementedError("per-tile prompts (:145-156) belong to the tiled sampler, not selected by the yaml")
        batch["txt"] = [" ".join([_p, p_p]) for _p in p]
        return self.conditioner.get_unconditional_conditioning(batch, batch_uc)

    def _posterior_noise(self, shape):
        return torch.randn(shape)

    def _randn_like(self, t):
        if self.noise_source == "cpu":
            return torch.randn(t.shape).to(t.device)
        return torch.randn_like(t)

    # ---- the sampling loop (:200-298) ---------------------------------------------------------
    @torch.no_grad()
    def just_sampling(self, x, p, p_p="default", n_p="default", img_threshold=0.1, dec_img=1.0, num_steps=100,
                      restoration_scale=4.0, s_churn=0, s_noise=1.003, cfg_scale=4.0, seed=-1, num_samples=1,
                      control_scale=1, color_fix_type="None", use_linear_CFG=False, use_linear_control_scale=False,
                      cfg_scale_start=1.0, control_scale_start=0.0, **kwargs):
        """Reference signature (:200-223).  Images of a batch are independent units: every kernel launch inside is
        planned for ONE image (``ops.plan_units``), so image b of a batch is bit-identical to its batch-of-1 run --
        with the feature cache on as well (per-image decisions, sub-batched second UNet half)."""
        from .. import ops
        with ops.plan_units(num_samples if num_samples > 1 else len(x)):
            return self._just_sampling(x, p, p_p, n_p, img_threshold, dec_img, num_steps, restoration_scale, s_churn, s_noise,
                                       cfg_scale, seed, num_samples, control_scale, color_fix_type, use_linear_CFG,
                                       use_linear_control_scale, cfg_scale_start, control_scale_start, **kwargs)

    def _just_sampling(self, x, p, p_p="default", n_p="default", img_threshold=0.1, dec_img=1.0, num_steps=100,
                       restoration_scale=4.0, s_churn=0, s_noise=1.003, cfg_scale=4.0, seed=-1, num_samples=1,
                       control_scale=1, color_fix_type="None", use_linear_CFG=False, use_linear_control_scale=False,
                       cfg_scale_start=1.0, control_scale_start=0.0, **kwargs):
        assert len(x) == len(p)
        assert color_fix_type in ["Wavelet", "AdaIn", "None"]
        N = len(x)
        if num_samples > 1:
            assert N == 1
            N = num_samples
            x = x.repeat(N, 1, 1, 1)
            p = p * N
        p_p = self.p_p if p_p == "default" else p_p
        n_p = self.n_p if n_p == "default" else n_p

        sp = self.sampler_config.params
        sp.num_steps = num_steps
        sp.guider_config.params.scale_min = cfg_scale
        sp.guider_config.params.scale = cfg_scale_start if use_linear_CFG else cfg_scale
        sp.restore_cfg, sp.s_churn, sp.s_noise = restoration_scale, s_churn, s_noise
        self.sampler = instantiate_from_config(self.sampler_config)

        # measurement (bench.py, through rsvld_amd.measure.hooks -- not reachable from the call signature): ``stamp(name)``
        # is called at every phase border, ``max_steps`` truncates the sampler loop (the per-image fixed part runs in full)
        _stamp, _max_steps = self._measure if self._measure is not None else (None, None)

        def stamp(name):
            if _stamp is not None:
                _stamp(name)

        x = x.float().contiguous()
        front = kwargs.pop("vae_front", None)
        if front is None:
            _z = self.encode_first_stage_with_denoise(x, use_sample=False)
            stamp("vae_denoise_encode")
            x_stage1 = self.decode_first_stage(_z)
            stamp("vae_decode_stage1")
            z_stage1 = self._encode_center(x_stage1, restoration_scale > 0)
            stamp("vae_encode_stage1")
        else:       # computed ahead of the call (``vae_front``), e.g. on a second stream beside the caption pass
            _z, x_stage1, z_stage1 = front
        c_img, uc_img = self.prepare_condition(_z, p, p_p, n_p, N)
        stamp("conditioner")

        def denoiser(inp, sigma, c, *a, **kw):
            return self.denoiser(self.model, inp, sigma, c, *a, **kw)

        noised_z = self._randn_like(_z)
        sampler = self.sampler
        sampler.noise_fn = self._randn_like   # per-step churn draws follow the same generator choice
        z, s_in, sigmas, num_sigmas, c_img, uc_img = sampler.init_loop(noised_z, c_img, uc=uc_img, num_steps=num_steps)
        x_center_cur = z_stage1
        if len(x) > 1 and num_samples == 1 and img_threshold > 0:
            # a batch of independent images: the cache decides PER IMAGE, one threshold per image, and
            # sampler.step returns a list.  ``num_samples > 1`` (ONE image repeated, :231-235) keeps the reference's
            # behaviour instead: one decision over the whole stacked [2 * num_samples, ...] tensor (DFBCache.py:98-112)
            img_threshold = [float(img_threshold)] * N
        stamp("sampler_init")
        self.cache_trace = []
        n_iter = num_sigmas - 1 if _max_steps is None else min(_max_steps, num_sigmas - 1)
        with cache_context(MyCacheContext()) as ctx:
            ctx.trace = self.cache_trace
            for i in range(n_iter):
                z, img_threshold = sampler.step(z, i, s_in, sigmas, denoiser, c_img, uc_img, x_center=x_center_cur,
                                                control_scale=control_scale,
                                                use_linear_control_scale=use_linear_control_scale,
                                                control_scale_start=control_scale_start, threshold=img_threshold)
                x_center_cur = z
                img_threshold = [t * dec_img for t in img_threshold] if isinstance(img_threshold, list) \
                    else img_threshold * dec_img
        stamp("edm_sampler_loop")

        samples = self.decode_first_stage(z)
        stamp("vae_decode_final")
        if color_fix_type == "Wavelet":
            samples = wavelet_reconstruction(samples, x_stage1)
        elif color_fix_type == "AdaIn":
            samples = adaptive_instance_normalization(samples, x_stage1)
        stamp("colour_fix")
        return samples
