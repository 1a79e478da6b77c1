"""Shared by the Stage-2 golden generator and the tests: the reduced configuration and the input recipes.

Channel widths cannot shrink (LightGLVUNet hard-codes 320/640/1280, SR_modules.py:544-548); transformer
depth, context width and the ADM vector are reduced so the CPU reference runs in seconds."""
import copy
import os

import torch
import yaml

WEIGHT_SEED = 4321
SMALL = dict(transformer_depth=[1, 1, 2], context_dim=64, adm_in_channels=32)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
YAML = os.path.join(ROOT, "remote-sensing-vision-language-diffusion-model_amd", "model_configs", "juggernautXL.yaml")


def cond_dicts():
    g2, g3 = torch.Generator().manual_seed(2), torch.Generator().manual_seed(3)
    c = {"crossattn": torch.randn(1, 77, SMALL["context_dim"], generator=g2), "vector": torch.randn(1, SMALL["adm_in_channels"], generator=g2)}
    uc = {"crossattn": torch.randn(1, 77, SMALL["context_dim"], generator=g3), "vector": torch.randn(1, SMALL["adm_in_channels"], generator=g3)}
    return c, uc


def product_params():
    """model.params of the product yaml with the reduced sizes and in-memory cached embeddings."""
    cfg = yaml.safe_load(open(YAML))["model"]["params"]
    for k in ("control_stage_config", "network_config"):
        cfg[k]["params"].update(copy.deepcopy(SMALL))
    c, uc = cond_dicts()
    cfg["conditioner_config"] = {"target": "rsvld_amd.sgm.modules.PreparedConditioner",   # the reference's cached-embedding class
                                 "params": {"cond_pth": c, "un_cond_pth": uc}}
    return cfg


def rnd(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


PIPE_OPT = dict(num_steps=6, s_churn=5, s_noise=1.003, cfg_scale=7.5, cfg_scale_start=4.0, use_linear_CFG=True,
                restoration_scale=-1, control_scale=1.0, img_threshold=0.3, dec_img=1.0, color_fix_type="Wavelet")

# variants of PIPE_OPT pinned by tests/golden/s2_branches.npz (gen_s2_branches_golden.py): branches the default call does not take
BRANCHES = {
    "restore": dict(restoration_scale=4.0),                                        # sampling.py:614-616
    "lincs": dict(use_linear_control_scale=True, control_scale_start=0.3),         # sampling.py:608-609
    "adain": dict(color_fix_type="AdaIn"),                                         # utils/colorfix.py:44-71
    "ns2": dict(num_samples=2, img_threshold=0.0),                                 # SR_model.py:231-235
}
STEP_OPT = dict(num_steps=50, s_churn=5, s_noise=1.003, restore_cfg=4.0, cfg_scale=7.5, cfg_scale_start=4.0)
