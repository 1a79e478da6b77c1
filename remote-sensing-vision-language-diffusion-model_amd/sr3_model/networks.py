"""Network factory for Stage 1 (reference: models/sr3_model/networks.py:84-136, 'sr3' branch)."""
import logging

from .sr3_modules import diffusion, unet

logger = logging.getLogger("base")


def define_G(opt):
    model_opt = opt["model"]
    if model_opt["which_model_G"] != "sr3":
        raise NotImplementedError(
            f"which_model_G={model_opt['which_model_G']!r}: only the 'sr3' generator is on the inference path")
    u = model_opt["unet"]
    if u.get("norm_groups") is None:
        u["norm_groups"] = 32
    model = unet.UNet(
        in_channel=u["in_channel"], out_channel=u["out_channel"], norm_groups=u["norm_groups"],
        inner_channel=u["inner_channel"], channel_mults=u["channel_multiplier"], attn_res=u["attn_res"],
        res_blocks=u["res_blocks"], dropout=u["dropout"], image_size=model_opt["diffusion"]["image_size"])
    # own extension: the reference runs Stage 1 in plain fp32 (no autocast); this build defaults to fp16 storage + fp32
    # accumulation (the fast 16-bit kernels) and offers the fp32-operand kernel family as "fp32" (csrc/f32.hip)
    model.set_compute_dtype(u.get("compute_dtype", "fp16"))
    netG = diffusion.GaussianDiffusion(
        model, image_size=model_opt["diffusion"]["image_size"], channels=model_opt["diffusion"]["channels"],
        loss_type="l1", conditional=model_opt["diffusion"]["conditional"],
        schedule_opt=model_opt["beta_schedule"]["train"])
    if opt.get("gpu_ids") and opt.get("distributed"):
        # the reference wraps in nn.DataParallel (networks.py:133-135); here multi-GPU is one process
        # per GPU (bench.py / rsvld_amd.parallel), so a single replica is returned.
        logger.info("distributed=True: use one process per GPU; returning a single replica")
    return netG
