#!/usr/bin/env python3
"""Build the cached text embeddings PreparedConditioner reads (model_configs/juggernautXL_cached.yaml).

    python tools/build_cached_cond.py --out_dir CKPT_PTH [--caption "..."] [--a_prompt "..."] [--n_prompt "..."] \
        [--yaml remote-sensing-vision-language-diffusion-model_amd/model_configs/juggernautXL.yaml] [--device cuda:0]

Instantiates ONLY the conditioner of the given yaml (GeneralConditionerWithControl: CLIP-L + OpenCLIP bigG + the three
size embedders; needs their checkpoints, i.e. network access or a populated HF / open_clip cache), runs it exactly as
SR_backbone.prepare_condition does (models/SR_model.py:127-156: sizes 1024^2, crop 0,0, positive text = caption + a_prompt,
negative text = n_prompt) and writes cond.pth / uncond.pth = {"crossattn": [1,77,2048], "vector": [1,2816]}."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch


def main():
    from rsvld_amd.infer import PipelineConfig
    from rsvld_amd.models.util import load_config
    from rsvld_amd.sgm.util import instantiate_from_config
    dflt = PipelineConfig.__dataclass_fields__
    ap = argparse.ArgumentParser()
    ap.add_argument("--yaml", default=os.path.join(ROOT, "remote-sensing-vision-language-diffusion-model_amd", "model_configs",
                                                   "juggernautXL.yaml"))
    ap.add_argument("--out_dir", default="./CKPT_PTH")
    ap.add_argument("--caption", default="")
    ap.add_argument("--a_prompt", default=dflt["a_prompt"].default)
    ap.add_argument("--n_prompt", default=dflt["n_prompt"].default)
    ap.add_argument("--device", default="cuda:0" if torch.cuda.is_available() else "cpu")
    a = ap.parse_args()
    cond = instantiate_from_config(load_config(a.yaml).model.params.conditioner_config).to(a.device).eval()
    sizes = {k: torch.tensor([v]).to(a.device) for k, v in (("original_size_as_tuple", [1024, 1024]),
                                                             ("crop_coords_top_left", [0, 0]),
                                                             ("target_size_as_tuple", [1024, 1024]))}
    control = torch.zeros(1, 4, 8, 8, device=a.device)                     # passed through untouched
    batch = dict(sizes, txt=[" ".join([a.caption, a.a_prompt])], control=control)
    batch_uc = dict(sizes, txt=[a.n_prompt], control=control)
    with torch.no_grad():
        c, uc = cond.get_unconditional_conditioning(batch, batch_uc)
    os.makedirs(a.out_dir, exist_ok=True)
    for name, d in (("cond.pth", c), ("uncond.pth", uc)):
        torch.save({k: d[k].float().cpu() for k in ("crossattn", "vector")}, os.path.join(a.out_dir, name))
        print("wrote", os.path.join(a.out_dir, name), {k: tuple(d[k].shape) for k in ("crossattn", "vector")})


if __name__ == "__main__":
    main()
