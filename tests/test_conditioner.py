"""Live conditioner: ConcatTimestepEmbedderND, FrozenCLIPEmbedder, FrozenOpenCLIPEmbedder2 and
GeneralConditionerWithControl against vectors the REFERENCE's classes produced over the same small seeded towers
(tests/golden/gen_cond_golden.py).  PyTorch modules on the CPU: the text towers stay stock PyTorch by design."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import cond_common as CC  # noqa: E402

TOL = dict(atol=2e-5, rtol=1e-5)   # same torch ops as the generator; slack for another host's BLAS


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "s2_conditioner.npz"))


def _close(a, b):
    assert a.shape == tuple(b.shape)
    assert np.allclose(a.detach().numpy(), b, **TOL), float(np.abs(a.detach().numpy() - b).max())


def test_concat_timestep_embedder(gold):
    from rsvld_amd.sgm.modules.encoders.modules import ConcatTimestepEmbedderND
    e = ConcatTimestepEmbedderND(256)
    x = torch.from_numpy(gold["cte_in"])
    _close(e(x), gold["cte_out"])
    _close(e(x[:, 0]), gold["cte_1d_out"])
    with pytest.raises(ValueError):
        e(torch.zeros(2, 2, 2))


def test_frozen_clip_embedder_layers(gold):
    from rsvld_amd.sgm.modules.encoders.modules import FrozenCLIPEmbedder
    hf, tok = CC.make_hf_clip()
    prompts = list(CC.PROMPTS)
    e = FrozenCLIPEmbedder(device="cpu", layer="hidden", layer_idx=2, tokenizer=tok, transformer=hf)
    _close(e(prompts), gold["clip_hidden2"])
    e = FrozenCLIPEmbedder(device="cpu", layer="last", always_return_pooled=True, tokenizer=tok, transformer=hf)
    z, pooled = e.encode(prompts)
    _close(z, gold["clip_last"])
    _close(pooled, gold["clip_pooled"])
    assert not any(p.requires_grad for p in e.parameters())
    with pytest.raises(ValueError):
        FrozenCLIPEmbedder(device="cpu", layer="hidden", tokenizer=tok, transformer=hf)   # layer_idx missing
    with pytest.raises(ValueError):
        FrozenCLIPEmbedder(device="cpu", layer="penultimate", tokenizer=tok, transformer=hf)


def test_frozen_openclip_embedder2_layers_and_pooling(gold):
    from rsvld_amd.sgm.modules.encoders.modules import FrozenOpenCLIPEmbedder2
    oc, tok = CC.make_open_clip()
    prompts = list(CC.PROMPTS)
    e = FrozenOpenCLIPEmbedder2(device="cpu", layer="penultimate", always_return_pooled=True, legacy=False, model=oc, tokenize=tok)
    z, pooled = e(prompts)
    _close(z, gold["oc_penultimate"])      # hidden states BEFORE the last block, no ln_final
    _close(pooled, gold["oc_pooled"])      # ln_final(last)[eot] @ text_projection
    e = FrozenOpenCLIPEmbedder2(device="cpu", layer="last", always_return_pooled=True, legacy=False, model=oc, tokenize=tok)
    _close(e(prompts)[0], gold["oc_last"])
    e = FrozenOpenCLIPEmbedder2(device="cpu", layer="last", legacy=True, model=oc, tokenize=tok)
    _close(e(prompts), gold["oc_legacy_last"])
    with pytest.raises(ValueError):
        FrozenOpenCLIPEmbedder2(device="cpu", layer="last", always_return_pooled=True, legacy=True, model=oc, tokenize=tok)(prompts)


def test_general_conditioner_with_control_matches_reference(gold):
    """The yaml's five embedders (model_configs/juggernautXL.yaml:67-105) through instantiate_from_config: crossattn =
    [CLIP hidden | OpenCLIP penultimate] on the channel axis, vector = [pooled | size | crop | target], control passed through;
    c and uc differ only through txt."""
    from rsvld_amd.sgm.modules.encoders.modules import GeneralConditionerWithControl
    hf, hf_tok = CC.make_hf_clip()
    oc, oc_tok = CC.make_open_clip()
    mod = "rsvld_amd.sgm.modules.encoders.modules."
    cfg = [
        {"is_trainable": False, "input_key": "txt", "target": mod + "FrozenCLIPEmbedder",
         "params": {"layer": "hidden", "layer_idx": 2, "device": "cpu", "tokenizer": hf_tok, "transformer": hf}},
        {"is_trainable": False, "input_key": "txt", "target": mod + "FrozenOpenCLIPEmbedder2",
         "params": {"layer": "penultimate", "always_return_pooled": True, "legacy": False, "device": "cpu", "model": oc, "tokenize": oc_tok}},
        {"is_trainable": False, "input_key": "original_size_as_tuple", "target": mod + "ConcatTimestepEmbedderND", "params": {"outdim": 256}},
        {"is_trainable": False, "input_key": "crop_coords_top_left", "target": mod + "ConcatTimestepEmbedderND", "params": {"outdim": 256}},
        {"is_trainable": False, "input_key": "target_size_as_tuple", "target": mod + "ConcatTimestepEmbedderND", "params": {"outdim": 256}},
    ]
    cond = GeneralConditionerWithControl(cfg)
    batch, batch_uc = CC.batches()
    c, uc = cond.get_unconditional_conditioning(batch, batch_uc)
    for name, d in (("c", c), ("uc", uc)):
        assert sorted(d) == ["control", "crossattn", "vector"]
        for k, v in d.items():
            _close(v, gold[f"cond_{name}_{k}"])
    assert c["crossattn"].shape == (2, 77, 32 + 48) and c["vector"].shape == (2, 40 + 6 * 256)
    assert torch.equal(c["vector"][:, 40:], uc["vector"][:, 40:])          # size embeddings do not depend on the prompt
    assert not torch.equal(c["crossattn"], uc["crossattn"])


def test_openclip_embedder_without_package_fails_loudly():
    from rsvld_amd.sgm.modules.encoders.modules import FrozenOpenCLIPEmbedder2
    try:
        import open_clip  # noqa: F401
        pytest.skip("open_clip is installed here")
    except ImportError:
        pass
    with pytest.raises(ImportError, match="open_clip"):
        FrozenOpenCLIPEmbedder2(arch="ViT-bigG-14", device="cpu")
