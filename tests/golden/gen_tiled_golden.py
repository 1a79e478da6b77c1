"""Window lists of the REFERENCE's `_sliding_windows` (sgm/modules/diffusionmodules/sampling.py:850-863) for a set of
latent sizes.  Authoring container only:  python tests/golden/gen_tiled_golden.py -> tests/golden/tiled_sampler_windows.json
(`gaussian_weights` next to it hard-codes device='cuda' and the sampler's tile loop cannot run as shipped, see
oracle/s2_oracle.py: neither can be captured here.)"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import ref_shims

ref_shims.install()
sys.path.insert(0, "/root/reference")
from sgm.modules.diffusionmodules.sampling import _sliding_windows

CASES = [(128, 128, 128, 64), (512, 512, 128, 64), (256, 384, 128, 64), (200, 136, 128, 64), (24, 40, 16, 8), (30, 17, 16, 8),
         (129, 128, 128, 64), (512, 512, 128, 96), (64, 64, 16, 16)]
json.dump([{"args": list(c), "windows": [list(t) for t in _sliding_windows(*c)]} for c in CASES],
          open(os.path.join(HERE, "tiled_sampler_windows.json"), "w"))
print("ok")
