"""Developer A/B switches of the tools layer (benchmark scripts under tools/, ``bench.py --dev-env``).

The product reads no environment variable for anything that changes what is computed or how it is planned: precision
compositions are ``ops.SplitPolicy`` arguments of the owning network, launch-plan constants are fields of the immutable
``rsvld_amd.ops.LaunchContext`` (``ops.tuning(...)`` around calls; ``ops.set_defaults(...)`` once at process start).  A tool that wants the old ``RSVLD_*`` switches calls ``apply_env()`` explicitly; nothing calls it on import.

    RSVLD_CONV_TILE / RSVLD_CONV_STAGES / RSVLD_CONV_KSPLIT / RSVLD_CONV_STAGING / RSVLD_HALO_NW /
    RSVLD_GEMM256_OFF / RSVLD_GEMM256_ONE_TILE   -> LaunchContext.tune   (rsvld_conv_desc.tune: every combination computes the same function)
    RSVLD_D64_KERNEL=b|c|p                        -> LaunchContext.d64_kernel (the three bit-identical forms of the d = 64 attention)
    RSVLD_PROFILE_DETAIL=1                        -> LaunchContext.profile_detail (layer shapes in the profiler group names)
    RSVLD_HALO_MIN_WGS=n                          -> LaunchContext.halo_min_wgs
"""
import os

from . import _lib as L
from . import ops


def tune_from_env(e=None):
    e = os.environ if e is None else e
    t = L.TUNE_TILE.get(e.get("RSVLD_CONV_TILE", ""), 0)
    if e.get("RSVLD_CONV_STAGES"):
        t |= (int(e["RSVLD_CONV_STAGES"]) & 7) << L.TUNE_STAGES_SHIFT
    if e.get("RSVLD_CONV_KSPLIT", "1")[:1] == "0":
        t |= L.TUNE_NO_KSPLIT
    if e.get("RSVLD_CONV_STAGING", "")[:1] == "r":
        t |= L.TUNE_REG_STAGING
    if e.get("RSVLD_HALO_NW"):
        t |= L.TUNE_HALO_NW8 if e["RSVLD_HALO_NW"][:1] == "8" else L.TUNE_HALO_NW4
    if e.get("RSVLD_GEMM256_OFF") is not None:
        t |= L.TUNE_NO_GEMM256
    if e.get("RSVLD_GEMM256_ONE_TILE") is not None:
        t |= L.TUNE_GEMM_ONE_TILE
    return t


def d64_kernel(name):
    """"b" (four-wave), "c" (ping-pong), "p" (pipelined), "" = the library's choice -> this thread's base LaunchContext.d64_kernel."""
    ops.set_defaults(d64_kernel={"b": 1, "c": 2, "p": 3}.get(name or "", 0))


def apply_env(e=None):
    e = os.environ if e is None else e
    ops.set_defaults(tune=tune_from_env(e), profile_detail=bool(e.get("RSVLD_PROFILE_DETAIL")))
    d64_kernel(e.get("RSVLD_D64_KERNEL", ""))
    if e.get("RSVLD_HALO_MIN_WGS"):
        ops.set_defaults(halo_min_wgs=int(e["RSVLD_HALO_MIN_WGS"]))
