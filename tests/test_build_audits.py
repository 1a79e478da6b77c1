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


@pytest.mark.parametrize("unit", ["conv_igemm.hip", "conv_halo.hip", "gemm.hip", "split.hip"])
def test_lds_dma_kernels_use_no_scratch(unit):
    """A spill or a compiler-built stack table inside a kernel that streams its operands by LDS-DMA is reloaded with scratch_load,
    which shares vmcnt with the DMA ring: the wait in front of the reload drains the prefetch (round 4: conv_igemm 586 -> 437 TFLOP/s
    from a runtime flag in its source selection).  No MFMA kernel of these units may contain a scratch instruction."""
    import audit_scratch
    res = audit_scratch.audit(unit)
    assert len(res) >= 3, res
    bad = {k: v for k, v in res.items() if v > 0}
    assert not bad, bad
