"""Host-side pre/post functions either side of the hot path against outputs of the
REFERENCE's own functions (tests/golden/gen_host_golden.py): PIL2Tensor (x64 rounding, min_size / fix_resize, bicubic),
Tensor2PIL (bicubic back to the hand-off size, 8-bit), tensor2img (clamp, 8-bit rounding).  Bit-exact: 8-bit images."""
import os
import sys

import numpy as np
import pytest
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [((37, 53), 1, 128, None), ((50, 40), 2, 64, None), ((96, 64), 1, 64, None), ((45, 31), 3, 64, 100), ((130, 70), 1, 64, None)]


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "host_prepost.npz"))


@pytest.mark.parametrize("i", range(len(CASES)))
def test_pil2tensor_tensor2pil_tensor2img(gold, i):
    from rsvld_amd.models.util import PIL2Tensor, Tensor2PIL
    from rsvld_amd.utils.tensor2img import tensor2img
    (w, h), up, ms, fr = CASES[i]
    img = Image.fromarray(gold[f"in{i}"])
    assert img.size == (w, h)
    x, h0, w0 = PIL2Tensor(img, upscale=up, min_size=ms, fix_resize=fr)
    want = torch.tensor(gold[f"p2t{i}_u8"] / 255 * 2 - 1, dtype=torch.float32)
    assert x.dtype == torch.float32 and x.shape[1] % 64 == 0 and x.shape[2] % 64 == 0
    assert torch.equal(x, want)
    assert [h0, w0] == list(gold[f"p2t{i}_hw"])
    y = (x + torch.from_numpy(gold[f"noise{i}"]).float()).clamp(-1.2, 1.2)
    out = Tensor2PIL(y, h0, w0)
    assert out.size == (w0, h0)
    assert np.array_equal(np.asarray(out), gold[f"t2p{i}"])
    u8 = tensor2img(y.unsqueeze(0).clone())
    assert u8.dtype == np.uint8 and np.array_equal(u8, gold[f"t2i{i}"])


def test_tensor2img_other_ranks(gold):
    from rsvld_amd.utils.tensor2img import tensor2img
    assert np.array_equal(tensor2img(torch.linspace(-1.5, 1.5, 48).reshape(6, 8)), gold["t2i_2d"])
    f = tensor2img(torch.linspace(-1, 1, 3 * 4 * 5).reshape(3, 4, 5), out_type=np.float32)
    assert f.dtype == np.float32 and np.array_equal(f, gold["t2i_f32"])
    with pytest.raises(TypeError):
        tensor2img(torch.zeros(4, 3, 8, 8))   # the reference's 4-D branch calls a make_grid it never imports


def test_stage1_loader_contract(tmp_path):
    """data/dataset.py:16-42 (torchvision semantics restated with PIL; the reference loader itself needs torchvision and is
    not runnable here): shorter side -> int(max(w,h)*scale) keeping the aspect, centre crop to a square, [-1,1] NCHW."""
    from rsvld_amd.data.dataset import dataloader, resize_and_convert
    rng = np.random.default_rng(3)
    img = Image.fromarray(rng.integers(0, 256, (20, 32, 3), dtype=np.uint8))
    p = str(tmp_path / "lr.png")
    img.save(p)
    batch = next(iter(dataloader(p, scale=4)))
    x = batch["SR"]
    assert x.shape == (1, 3, 128, 128) and x.dtype == torch.float32      # max(32, 20) * 4
    assert float(x.min()) >= -1.0 and float(x.max()) <= 1.0
    sq = resize_and_convert(img, 4)
    assert sq.size == (128, 128)
    # the crop is centred: the same image flipped left-right yields the flipped crop
    flipped = resize_and_convert(img.transpose(Image.FLIP_LEFT_RIGHT), 4).transpose(Image.FLIP_LEFT_RIGHT)
    assert np.abs(np.asarray(flipped).astype(int) - np.asarray(sq).astype(int)).max() <= 1
    # a square input is only resized
    sq2 = resize_and_convert(Image.fromarray(rng.integers(0, 256, (16, 16, 3), dtype=np.uint8)), 2)
    assert sq2.size == (32, 32)



def test_stage1_loader_vs_fixture(tmp_path):
    """Product loader == tests/golden/stage1_loader.npz bit for bit, and so is the oracle restatement that produced it
    (torchvision formulas, oracle/loader_oracle.py; the reference loader needs torchvision and cannot run in the
    authoring container: parity unpinned by a reference run)."""
    import os
    from oracle import loader_oracle as LO
    from rsvld_amd.data.dataset import load_sr_input
    sys_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_loader_golden", os.path.join(sys_path, "gen_loader_golden.py"))
    G = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(G)
    z = np.load(os.path.join(sys_path, "stage1_loader.npz"))
    for i, ((w, h), s) in enumerate(G.CASES):
        img = Image.fromarray(z[f"in{i}"])
        assert img.size == (w, h)
        want = ((z[f"out{i}_u8"].astype(np.float32) / np.float32(255)) - np.float32(0.5)) / np.float32(0.5)
        assert np.array_equal(LO.load(img, s), want)
        p = str(tmp_path / f"lr{i}.png")
        img.save(p)
        got = load_sr_input(p, s)["SR"].numpy()
        side = int(max(w, h) * s)
        assert got.shape == (1, 3, side, side) and np.array_equal(got, want), (i, w, h, s)


def test_stage1_option_parser_refuses_a_missing_checkpoint(tmp_path):
    """A configured-but-missing ``resume_state`` must raise (same rule as ``create_SR_model``), not fall back to random weights
    (reference: sr3_model/model.py:149-170 loads the file unconditionally); ``allow_random_init`` is the explicit opt-out."""
    import json
    from rsvld_amd.configs import sr3 as SR3
    from rsvld_amd.utils import logger as Logger
    base = json.loads("".join(l.split("//")[0] for l in open(SR3.SR3_Config.config)))
    base["path"]["resume_state"] = str(tmp_path / "missing_ckpt")
    cfg = tmp_path / "opt.json"
    cfg.write_text(json.dumps(base))
    args = SR3.SR3_Config()
    args.config = str(cfg)
    with pytest.raises(FileNotFoundError):
        Logger.parse(args)
    assert Logger.parse(args, allow_random_init=True)["path"]["resume_state"] is None
    (tmp_path / "missing_ckpt_gen.pth").write_bytes(b"x")
    assert Logger.parse(args)["path"]["resume_state"] == str(tmp_path / "missing_ckpt")
