"""bench.py seeds the Stage-2 networks on the device and re-draws the ``zero_module`` tensors (an all-zero projection would make the
network's output 0): WHICH tensors are tagged as zero-initialised must not depend on what the process imported or built before
(round 5: the first build of a process left the lazily imported modules' ``zero_module`` unwrapped, every later build wrapped it, so
the "seeded" weights depended on test order).  Runs in a fresh interpreter: first build == second build."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch, bench
import s2_common as S
from rsvld_amd.sgm.util import instantiate_from_config
def tagged():
    import rsvld_amd.models.SR_model
    with bench._host_init_skipped():
        m = instantiate_from_config({"target": "rsvld_amd.models.SR_model.SR_backbone", "params": S.product_params()})
    return sorted(n for n, p in m.named_parameters() if getattr(p, "_zero_init", False))
a, b = tagged(), tagged()
assert a == b, (len(a), len(b), sorted(set(a) ^ set(b))[:5])
assert any("out_layers.3" in n for n in a) and any("proj_out" in n for n in a) and any("zero_conv" in n or "zero_" in n for n in a), a[:10]
print("TAGGED", len(a))
"""


def test_zero_module_tagging_is_the_same_on_the_first_and_on_later_builds():
    out = subprocess.run([sys.executable, "-c", SCRIPT % (ROOT, os.path.join(ROOT, "tests", "golden"))], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "TAGGED" in out.stdout
