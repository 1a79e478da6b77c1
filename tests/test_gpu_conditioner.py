"""The live conditioner ON THE DEVICE: the same cases as tests/test_conditioner.py -- vectors the
REFERENCE's classes produced over small seeded text towers -- with the towers, the size embedders and the conditioner
running on cuda:0 (stock PyTorch-ROCm modules, fp32), and the conditioner's output consumed by the HIP denoiser.
Tolerance 2e-4: fp32 on the GPU (different GEMM reduction order) against the CPU-generated goldens."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import cond_common as CC  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = dict(atol=2e-4, rtol=1e-4)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "s2_conditioner.npz"))


def _close(a, b):
    assert a.is_cuda and a.shape == tuple(b.shape)
    a = a.detach().float().cpu().numpy()
    assert np.allclose(a, b, **TOL), float(np.abs(a - b).max())


def test_embedders_on_device(cuda, gold):
    from rsvld_amd.sgm.modules.encoders.modules import ConcatTimestepEmbedderND, FrozenCLIPEmbedder, FrozenOpenCLIPEmbedder2
    prompts = list(CC.PROMPTS)
    e = ConcatTimestepEmbedderND(256)
    _close(e(torch.from_numpy(gold["cte_in"]).to(cuda)), gold["cte_out"])
    hf, tok = CC.make_hf_clip()
    e = FrozenCLIPEmbedder(device="cuda", layer="hidden", layer_idx=2, tokenizer=tok, transformer=hf.to(cuda))
    _close(e(prompts), gold["clip_hidden2"])
    e = FrozenCLIPEmbedder(device="cuda", layer="last", always_return_pooled=True, tokenizer=tok, transformer=hf)
    z, pooled = e.encode(prompts)
    _close(z, gold["clip_last"])
    _close(pooled, gold["clip_pooled"])
    oc, otok = CC.make_open_clip()
    e = FrozenOpenCLIPEmbedder2(device="cuda", layer="penultimate", always_return_pooled=True, legacy=False, model=oc.to(cuda), tokenize=otok)
    z, pooled = e(prompts)
    _close(z, gold["oc_penultimate"])
    _close(pooled, gold["oc_pooled"])


def test_general_conditioner_with_control_on_device(cuda, gold):
    from rsvld_amd.sgm.modules.encoders.modules import GeneralConditionerWithControl
    hf, hf_tok = CC.make_hf_clip()
    oc, oc_tok = CC.make_open_clip()
    mod = "rsvld_amd.sgm.modules.encoders.modules."
    cfg = [
        {"is_trainable": False, "input_key": "txt", "target": mod + "FrozenCLIPEmbedder",
         "params": {"layer": "hidden", "layer_idx": 2, "device": "cuda", "tokenizer": hf_tok, "transformer": hf.to(cuda)}},
        {"is_trainable": False, "input_key": "txt", "target": mod + "FrozenOpenCLIPEmbedder2",
         "params": {"layer": "penultimate", "always_return_pooled": True, "legacy": False, "device": "cuda", "model": oc.to(cuda),
                    "tokenize": oc_tok}},
        {"is_trainable": False, "input_key": "original_size_as_tuple", "target": mod + "ConcatTimestepEmbedderND", "params": {"outdim": 256}},
        {"is_trainable": False, "input_key": "crop_coords_top_left", "target": mod + "ConcatTimestepEmbedderND", "params": {"outdim": 256}},
        {"is_trainable": False, "input_key": "target_size_as_tuple", "target": mod + "ConcatTimestepEmbedderND", "params": {"outdim": 256}},
    ]
    cond = GeneralConditionerWithControl(cfg).to(cuda)
    batch, batch_uc = CC.batches()
    dev = lambda d: {k: (v.to(cuda) if torch.is_tensor(v) else v) for k, v in d.items()}
    c, uc = cond.get_unconditional_conditioning(dev(batch), dev(batch_uc))
    for name, d in (("c", c), ("uc", uc)):
        assert sorted(d) == ["control", "crossattn", "vector"]
        for k, v in d.items():
            _close(v, gold[f"cond_{name}_{k}"])
