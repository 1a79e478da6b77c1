"""Generate the conditioner goldens by running the REFERENCE's embedder classes (sgm/modules/encoders/modules.py)
over the small seeded towers of cond_common.py.  Authoring container only:
    python tests/golden/gen_cond_golden.py        -> tests/golden/s2_conditioner.npz
The reference constructors download / load real checkpoints (CLIP-L, OpenCLIP bigG: absent offline), so the objects
are created with object.__new__ and given exactly the attributes the constructors would set; every forward /
encode_with_transformer / pool / get_unconditional_conditioning that produces the vectors is the reference's code."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_shims

ref_shims.install()
sys.path.insert(0, "/root/reference")

import numpy as np
import torch
from torch import nn

import cond_common as CC

import sgm.modules.encoders.modules as RM


def bare(cls, **attrs):
    e = object.__new__(cls)
    RM.AbstractEmbModel.__init__(e)
    for k, v in attrs.items():
        setattr(e, k, v)
    return e


@torch.no_grad()
def main():
    out = {}
    # ---- ConcatTimestepEmbedderND: constructed normally
    cte = RM.ConcatTimestepEmbedderND(256)
    x = torch.tensor([[1024.0, 1024.0], [0.0, 0.0], [512.0, 768.0], [3.5, 4096.0]])
    out["cte_in"] = x.numpy()
    out["cte_out"] = cte(x).numpy()
    out["cte_1d_out"] = cte(x[:, 0]).numpy()

    # ---- FrozenCLIPEmbedder (yaml :72-77: layer hidden, layer_idx 11 of 12 -> here 2 of 3)
    hf, hf_tok = CC.make_hf_clip()
    clip = bare(RM.FrozenCLIPEmbedder, tokenizer=hf_tok, transformer=hf, device="cpu", max_length=77, layer="hidden",
                layer_idx=2, return_pooled=False)
    prompts = list(CC.PROMPTS)
    out["clip_hidden2"] = clip(prompts).numpy()
    clip.layer = "last"
    clip.return_pooled = True
    z, pooled = clip(prompts)
    out["clip_last"], out["clip_pooled"] = z.numpy(), pooled.numpy()
    clip.layer, clip.return_pooled = "hidden", False

    # ---- FrozenOpenCLIPEmbedder2 (yaml :79-88: penultimate, always_return_pooled, legacy False)
    oc, oc_tok = CC.make_open_clip()
    RM.open_clip.tokenize = oc_tok
    big = bare(RM.FrozenOpenCLIPEmbedder2, model=oc, device="cpu", max_length=77, return_pooled=True, layer="penultimate",
               layer_idx=1, legacy=False)
    z, pooled = big(prompts)
    out["oc_penultimate"], out["oc_pooled"] = z.numpy(), pooled.numpy()
    big.layer, big.layer_idx = "last", 0
    z, _ = big(prompts)
    out["oc_last"] = z.numpy()
    big.legacy, big.return_pooled = True, False
    out["oc_legacy_last"] = big(prompts).numpy()
    big.layer, big.layer_idx, big.legacy, big.return_pooled = "penultimate", 1, False, True

    # ---- the whole conditioner of the yaml (five embedders), c and uc
    cond = object.__new__(RM.GeneralConditionerWithControl)
    nn.Module.__init__(cond)
    embs = [clip, big, RM.ConcatTimestepEmbedderND(256), RM.ConcatTimestepEmbedderND(256), RM.ConcatTimestepEmbedderND(256)]
    for e, key in zip(embs, ("txt", "txt", "original_size_as_tuple", "crop_coords_top_left", "target_size_as_tuple")):
        e.is_trainable, e.ucg_rate, e.input_key, e.legacy_ucg_val = False, 0.0, key, None
    cond.embedders = nn.ModuleList(embs)
    batch, batch_uc = CC.batches()
    c, uc = cond.get_unconditional_conditioning(batch, batch_uc)
    for name, d in (("c", c), ("uc", uc)):
        assert sorted(d) == ["control", "crossattn", "vector"], sorted(d)
        for k, v in d.items():
            out[f"cond_{name}_{k}"] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "s2_conditioner.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
