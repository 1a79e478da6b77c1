"""Window lists of the REFERENCE's `_sliding_windows` (sgm/modules/diffusionmodules/sampling.py:850-863) for a set of
latent sizes.  Authoring container only:  python tests/golden/gen_tiled_golden.py -> tests/golden/tiled_sampler_windows.json
+ the blend mask of the REFERENCE's `gaussian_weights` (:830-847) -> tests/golden/tiled_sampler_mask.npz.  That function is pure numpy up to
its last line, `torch.tile(torch.tensor(weights, device='cuda'), ...)`: it is CALLED here as it stands, with `torch.tensor` wrapped for the
duration of the call so that the `device` keyword is dropped (no GPU in the authoring container); nothing of its arithmetic is touched.
(The sampler's tile LOOP cannot run as shipped -- `sampler_step` returns a tuple that :739-753 multiplies by a tensor, see oracle/s2_oracle.py
-- and stays "parity unpinned".)"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import ref_shims

ref_shims.install()
sys.path.insert(0, "/root/reference")
import numpy as np
import torch
from sgm.modules.diffusionmodules.sampling import _sliding_windows, gaussian_weights

CASES = [(128, 128, 128, 64), (512, 512, 128, 64), (256, 384, 128, 64), (200, 136, 128, 64), (24, 40, 16, 8), (30, 17, 16, 8),
         (129, 128, 128, 64), (512, 512, 128, 96), (64, 64, 16, 16)]
json.dump([{"args": list(c), "windows": [list(t) for t in _sliding_windows(*c)]} for c in CASES],
          open(os.path.join(HERE, "tiled_sampler_windows.json"), "w"))


def reference_mask(tw, th, nb):
    real = torch.tensor

    def on_cpu(*a, **k):
        k.pop("device", None)
        return real(*a, **k)

    torch.tensor = on_cpu
    try:
        return gaussian_weights(tw, th, nb)
    finally:
        torch.tensor = real


masks = {}
for tw, th in ((128, 128), (16, 16), (9, 16)):
    m = reference_mask(tw, th, 2)
    assert m.dtype == torch.float64 and tuple(m.shape) == (2, 4, th, tw) and bool((m[0, 0] == m[1, 3]).all())
    masks[f"mask_{tw}x{th}"] = m[0, 0].numpy()
np.savez_compressed(os.path.join(HERE, "tiled_sampler_mask.npz"), **masks)
print("ok", {k: v.shape for k, v in masks.items()})
