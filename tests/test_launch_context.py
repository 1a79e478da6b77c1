"""The launch context of rsvld_amd.ops (plan divisor, precision policy, developer overrides, profiler): one immutable object in a
ContextVar instead of module globals -- nested ``with`` blocks compose and restore, threads do not see each other's settings."""
import threading

import pytest


def test_context_nesting_restores_and_is_immutable():
    from rsvld_amd import ops
    base = ops.context()
    assert (base.plan_div, base.policy, base.tune, base.use_halo, base.halo_min_wgs, base.profiler) == (1, None, 0, True, 256, None)
    with ops.plan_units(3):
        with ops.f32_split(True):
            with ops.tuning(use_halo=False, halo_min_wgs=0):
                c = ops.context()
                assert (c.plan_div, c.use_halo, c.halo_min_wgs) == (3, False, 0) and c.policy is ops.UNET_POLICY
                assert ops.f16_group("attn") and not ops.f16_group("conv1") and ops.precision_token() == ops.UNET_POLICY.key()
            assert ops.context().use_halo and ops.context().plan_div == 3
        assert ops.context().policy is None and ops.precision_token() is None and not ops.f16_group("attn")
    assert ops.context() is base
    with pytest.raises(AttributeError):
        ops.context().tune = 1
    with pytest.raises(TypeError):
        ops.tuning(no_such_field=1).__enter__()
    with pytest.raises(RuntimeError):
        with ops.f32_split(ops.ALL_SPLIT):
            raise RuntimeError("x")
    assert ops.context() is base                      # restored on the exception path too


def test_two_threads_hold_two_precisions():
    """What module globals could not do: a second thread (a second HIP stream's driver) runs another precision at the same time."""
    from rsvld_amd import ops
    seen, gate_a, gate_b = {}, threading.Event(), threading.Event()

    def worker():
        with ops.f32_split(ops.ALL_SPLIT), ops.plan_units(2):
            gate_a.set()
            gate_b.wait(10)
            seen["worker"] = (ops.precision_token(), ops.context().plan_div)

    th = threading.Thread(target=worker)
    with ops.f32_split(ops.UNET_POLICY), ops.plan_units(7):
        th.start()
        gate_a.wait(10)
        seen["main"] = (ops.precision_token(), ops.context().plan_div)
        gate_b.set()
        th.join()
    assert seen["main"] == (ops.UNET_POLICY.key(), 7) and seen["worker"] == (ops.ALL_SPLIT.key(), 2)
