"""Build-time audits of the hand-scheduled kernels (CPU: hipcc cross-compiles gfx950 assembly here)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


import pytest


@pytest.mark.parametrize("unit", ["gemm.hip", "attention.hip"])
def test_compiler_never_touches_m0_beside_the_asm_lds_dma(unit):
    """The LDS-DMA pieces of gemm256_kernel and attn_d512b_kernel are inline asm that writes M0 (the LDS destination) and declares
    it clobbered instead of saving / restoring it; hipcc reserves M0 and only warns.  The build is sound as long as the compiler
    itself never reads or writes M0 in that translation unit: every M0 use in the assembly must sit inside an
    ;;#ASMSTART .. ;;#ASMEND block."""
    import audit_m0
    bad, dma = audit_m0.audit(unit)
    assert dma >= 16, dma            # the pieces are really there
    assert not bad, bad[:5]
