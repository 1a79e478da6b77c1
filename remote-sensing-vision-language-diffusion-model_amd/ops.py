"""Tensor-level wrappers over the C ABI (include/rsvld_hip.h).

torch is used here only for device memory (allocation through its caching allocator) and for the
current HIP stream; every computation is a kernel of librsvld_hip.so.  Activations are NHWC
16-bit tensors ``[B, H, W, C]`` with ``C % 8 == 0``; tokens ``[B, N, C]`` are the same thing with
``H = 1``.  Every function raises if the tensor is not on a GPU: there is no CPU path.

fp32 activations select the fp32-operand family (csrc/f32.hip): ``conv2d`` /
``linear``, ``group_norm*``, ``layer_norm``, ``attention``, ``concat_c``, ``axpby``, ``absdiff_sums`` and ``nchw_to_nhwc``
dispatch on the tensor's dtype (the VAE under ``ae_dtype: fp32``, the UNet / ControlNet under ``diffusion_dtype: fp32``).
"""
import contextlib
import ctypes as C
import contextvars
import math

import torch

from . import _lib as L

_DT = {torch.float16: L.F16, torch.bfloat16: L.BF16}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.RsvldError("rsvld_amd ops run on the GPU only (got a CPU tensor); there is no CPU fallback")


def _dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise L.RsvldError(f"16-bit activation expected, got {t.dtype}")


def pad8(c):
    return (c + 7) // 8 * 8


# ----------------------------------------------------------------------------- launch context
# Everything a launch reads BESIDES its arguments lives in ONE immutable ``LaunchContext`` held in a ``contextvars.ContextVar``:
# the launch-plan divisor, the precision policy of the network that is running, the developer A/B overrides and the profiler.
# The context managers below (``plan_units``, ``f32_split``, ``tuning``) install a modified COPY for the duration of a ``with``
# block and restore the previous object on exit; nothing in this module is a mutable global.  A ContextVar is per thread (and per
# asyncio task): two threads driving two HIP streams can run two precisions at once; nested blocks on one thread compose.
class LaunchContext:
    """``plan_div``  batch-invariant plans (include/rsvld_hip.h, conventions): every launch is planned -- kernel family, tile shape,
                  split-K / split-KV -- for ONE of ``plan_div`` independent work units (images) stacked along its batch, so an
                  image's result is bit-identical however many images share the launch.  1 = plan on the whole call.
    ``policy``    the ``SplitPolicy`` of the split-precision network whose forward is running (None: not in one).
    ``tune``      developer A/B overrides -> rsvld_conv_desc.tune (every combination computes the same function); 0 = the library's choice.
    ``use_halo`` / ``halo_min_wgs`` / ``split_halo_min_wgs``   routing of eligible 3x3 convolutions through conv_halo.hip (below).
    ``split_d512_fused_min`` / ``split_attn_s_bytes``          form of the split-precision d != 64 attention (below).
    ``d64_kernel``                                             developer A/B of the d = 64 attention's three bit-identical forms.
    ``profiler`` / ``profile_detail``   per-launch HIP-event bracketing (``LaunchProfiler``) and layer shapes in its group names."""

    __slots__ = ("plan_div", "policy", "tune", "use_halo", "halo_min_wgs", "split_halo_min_wgs", "split_d512_fused_min",
                 "split_attn_s_bytes", "profiler", "profile_detail", "d64_kernel")

    def __init__(self, plan_div=1, policy=None, tune=0, use_halo=True, halo_min_wgs=256, split_halo_min_wgs=64,
                 split_d512_fused_min=2048, split_attn_s_bytes=32 << 30, profiler=None, profile_detail=False, d64_kernel=0):
        # d64_kernel: developer A/B (devtools.d64_kernel): 1 / 2 / 3 = force attn_d64b / attn_d64c / attn_d64p (bit-identical forms)
        # use_halo: route eligible 3x3 convs through conv_halo.hip (False: A/B against the gather kernel)
        # halo_min_wgs 256: below one workgroup per CU the 8x32-pixel halo tile under-fills the chip (measured 130 vs 334 TFLOP/s on
        #   32x32 maps): such layers use the 64x128 gather kernel with the 3-stage ring
        # split_halo_min_wgs 64: the split precision's gather kernel runs at 190-200 effective TFLOP/s where the halo kernel does
        #   385-400, and the small maps of this precision are the tiles of the tiled VAE, stacked 7-49 per launch (86 x 86 x 512:
        #   132 workgroups per planning unit, 72 x 72: 108, 86 x 64: 88): the halo tile from a quarter of a chip per unit
        # split_d512_fused_min 2048: query rows from which the fused split d = 512 kernel runs (64 per workgroup)
        # split_attn_s_bytes 32 GiB: fp32 score block of the GEMM form (P planes: as much again)
        object.__setattr__(self, "plan_div", max(1, int(plan_div)))
        for k, v in (("policy", policy), ("tune", int(tune)), ("use_halo", bool(use_halo)), ("halo_min_wgs", int(halo_min_wgs)),
                     ("split_halo_min_wgs", int(split_halo_min_wgs)), ("split_d512_fused_min", int(split_d512_fused_min)),
                     ("split_attn_s_bytes", int(split_attn_s_bytes)), ("profiler", profiler), ("profile_detail", bool(profile_detail)),
                     ("d64_kernel", int(d64_kernel))):
            object.__setattr__(self, k, v)

    def __setattr__(self, k, v):
        raise AttributeError("LaunchContext is immutable: use ops.tuning(...) / ops.plan_units(n) / ops.f32_split(policy) around the calls")

    def replace(self, **changes):
        kw = {k: getattr(self, k) for k in self.__slots__}
        bad = [k for k in changes if k not in kw]
        if bad:
            raise TypeError(f"LaunchContext has no field {bad}")
        kw.update(changes)
        return LaunchContext(**kw)


_CTX = contextvars.ContextVar("rsvld_launch_context", default=LaunchContext())


def context():
    """The LaunchContext the next launch of this thread will read."""
    return _CTX.get()


@contextlib.contextmanager
def tuning(**changes):
    """``with ops.tuning(use_halo=False, halo_min_wgs=0, tune=..., profiler=...):`` -- a modified copy of the current context for the block."""
    token = _CTX.set(_CTX.get().replace(**changes))
    try:
        yield _CTX.get()
    finally:
        _CTX.reset(token)


def set_defaults(**changes):
    """Process start-up only (rsvld_amd.devtools.apply_env, bench.py's flags): change this thread's BASE context for good."""
    _CTX.set(_CTX.get().replace(**changes))


def plan_units(n):
    """Inside ``with plan_units(n)`` every launch is planned for ONE of the ``n`` independent work units stacked along its batch."""
    return tuning(plan_div=max(1, int(n)))


# ----------------------------------------------------------------------------- split-operand precision mode
class SplitPolicy:
    """What runs at which width inside a network in the "split" precision (fp32 tensors; a matrix product = three bf16 MFMAs on
    hi + lo operands).  A policy is an ARGUMENT of the owning network (``SR_backbone.set_precision(..., policy=...)``, SR3's
    ``set_compute_dtype("split", policy=...)``), never the environment; it is part of the hipGraph key and printed by bench.py.

    ``impl``        "planes": the product path (bf16 planes + weight triples through gemm256 / conv_halo / conv_igemm);
                    "f32": round 3's on-the-fly split inside the fp32 family's simple kernels (an independent implementation of the
                    same arithmetic, kept for tests and A/B runs).
    ``f16_inputs``  the layer groups whose INPUT may be rounded to fp16 (11 significant bits) -- each measured against the reference's
                    CPU runs after 50 + 50 steps and at full depth on the device.  Their weights keep ~22 bits:
                    the layer runs as fp16 activation x fp16 weight pair [W_lo | W_hi], two MFMAs per product (dtype RSVLD_F16W2).
                      "attn"      the attention operands q, k, v, P: the 16-bit attention kernels (attn_d64c / attn_d512b)
                      "attn_out"  the input of ``to_out``: with "attn" it IS the fp16 attention output (no further rounding)
                      "ff"        the two inputs of a FeedForward: LayerNorm3's output and the GEGLU product
                      "qkv"       the inputs of to_q / to_k / to_v: LayerNorm1 / LayerNorm2's outputs (ZeroCrossAttn: its two GroupNorms)
                      "proj"      the inputs of proj_in / proj_out -- measured (+1.3e-4 after 50 steps for two GEMMs per transformer): not
                                  in the default.
                      "conv1", "conv2"  the input of a ResBlock's first / second 3x3 convolution (GroupNorm + SiLU outputs; NOT conv_in,
                                  the Down / Upsample and ZeroSFT convolutions or the output convolution).  Every convolution input in
                                  fp16 is 3.4e-3 on the CPU restatement; experiments only: on the reduced-depth goldens the first
                                  convolution's input costs +1.5e-4 after 50 steps and the second's +3.2e-4, but at full network depth the
                                  first alone ends at 1.2-1.7e-3 -- the reason the full-depth test exists.
                    Measured after 50 Stage-2 steps against the reference's CPU run (tools/tolerance_check.py, round 5, max / mean):
                    ("attn",) 2.5e-4 / 3.8e-5; + attn_out + ff 3.5e-4 / 4.6e-5; + qkv 3.5e-4 / 5.3e-5 (the default); + conv1 5.0e-4 / 6.7e-5
                    on the reduced-depth goldens but 1.2e-3 / 1.7e-3 (cache off / 0.3) at FULL depth against the fp32 family
                    (tests/test_gpu_fulldepth.py): outside the bar, rejected; qkv + conv2 6.6e-4 / 9.7e-5; qkv + conv1 + conv2 6.9e-4 / 9.3e-5;
                    qkv + proj 4.8e-4 / 6.5e-5.
    ``f16_weights`` the GEMM groups whose WEIGHTS are rounded to fp16 as well -- fp16 activation x fp16 weight, ONE MFMA per product, the plain
                    gemm256 / conv_igemm kernels -- where the input already is fp16 (default: all four wherever ``f16_inputs`` allows):
                      "qkv"       to_q / to_k / to_v (fp16 out: the attention operands)
                      "geglu"     the GEGLU projection of a FeedForward (fp16 out)
                      "attn_out", "ff_out"   to_out / ff.net.2 (fp32 out + fp32 residual: dtype RSVLD_F16W1)
                    A rounded weight is the SAME perturbation at every step, so this was expected to cost more than a rounded activation; measured
                    it does not: after 50 steps on the reduced-depth goldens 3.5e-4 / 5.3e-5 (every weight a pair) -> 5.3e-4 / 6.3e-5 (qkv + geglu) ->
                    4.5e-4 / 7.0e-5 (all four: the default; single groups scatter between 4.5e-4 and 7.8e-4 in the maximum, 6.3-7.5e-5 in the
                    mean); at FULL depth against the fp32 family 2.9e-4 -> 5.4e-4 (cache off), 4.0e-4 -> 4.9e-4 (0.3), means 3.9e-5 / 4.6e-5, every
                    cache decision equal (tools/tolerance_check.py, tools/fulldepth_check.py; the residual stream, the norms, proj_in / proj_out and
                    every convolution stay fp32 / three-MFMA).  16.7 -> 10.5 s of GEMM time per 4096^2 image.
                    The convolutions are different: their weights rounded to fp16 (emulated through the triples of fp16(W), round 5) end at
                    1.8e-3 on the goldens and 2.6e-3 / 2.3e-3 at full depth -- no two-MFMA form of the 3x3 layers on fp16 planes x fp16 weights.
    ``q8_convs``    (round 6) the ResBlock convolutions ("conv1", "conv2": the two 3x3 layers behind a GroupNorm + SiLU) that keep BOTH operands
                    at ~22 bits but spend fewer matrix cycles on them: x w = f16(x) f16(w) + x_lo w_hi + x_hi w_lo with the two cross terms
                    (2^-11 of the product) contracted in e4m3 on the block-scaled matrix instruction (dtype RSVLD_F16Q8: per 32 channels and tap
                    two fp16 MFMAs + one scaled MFMA instead of six bf16 MFMAs).  Emulated at full depth over 50 steps before the kernel was
                    written: mean distance from the fp32 family +3.5 %, maxima 5.8e-4 / 5.1e-4, every cache decision unchanged
                    (profiles/r06_conv_lo8_emulation.txt).  Default: both groups in the shipped composition, none elsewhere (the VAE: real
                    SDXL-VAE activations leave the fp16 range)."""

    __slots__ = ("impl", "f16_inputs", "f16_weights", "q8_convs")
    GROUPS = ("attn", "attn_out", "ff", "qkv", "proj", "conv1", "conv2")
    WEIGHT_GROUPS = ("qkv", "geglu", "attn_out", "ff_out")
    Q8_GROUPS = ("conv1", "conv2")

    def __init__(self, impl="planes", f16_inputs=("attn", "attn_out", "ff", "qkv"), f16_weights=None, q8_convs=None):
        if q8_convs is None:        # the default: both ResBlock convolutions, in the shipped composition of the UNets only
            q8_convs = self.Q8_GROUPS if (impl == "planes" and {"attn", "attn_out", "ff", "qkv"} <= set(f16_inputs)) else ()
        bad = [g for g in q8_convs if g not in self.Q8_GROUPS]
        if bad:
            raise ValueError(f"SplitPolicy.q8_convs {bad}: one of {self.Q8_GROUPS}")
        if set(q8_convs) & set(f16_inputs):
            raise ValueError("SplitPolicy: a convolution's input is either rounded to fp16 (f16_inputs) or kept with e4m3 cross terms (q8_convs)")
        if q8_convs and impl != "planes":
            raise ValueError("SplitPolicy.q8_convs exists in the product path (impl \"planes\") only")
        self.q8_convs = frozenset(q8_convs)
        if f16_weights is None:     # the default: every transformer GEMM whose input is fp16
            f16_weights = tuple(g for g, i in (("qkv", "qkv"), ("geglu", "ff"), ("attn_out", "attn_out"), ("ff_out", "ff")) if i in f16_inputs and impl == "planes")
        bad = [g for g in f16_weights if g not in self.WEIGHT_GROUPS]
        if bad:
            raise ValueError(f"SplitPolicy.f16_weights {bad}: one of {self.WEIGHT_GROUPS}")
        self.f16_weights = frozenset(f16_weights)
        if impl not in ("planes", "f32"):
            raise ValueError(f"SplitPolicy.impl {impl!r}: planes or f32")
        bad = [g for g in f16_inputs if g not in self.GROUPS]
        if bad:
            raise ValueError(f"SplitPolicy.f16_inputs {bad}: one of {self.GROUPS}")
        self.impl, self.f16_inputs = impl, frozenset(f16_inputs)
        if self.f16_inputs - {"attn"} and "attn" not in self.f16_inputs:
            raise ValueError("SplitPolicy: fp16 layer inputs go with fp16 attention operands (\"attn\")")
        if self.f16_inputs and impl != "planes":
            raise ValueError("SplitPolicy: fp16 layer inputs exist in the product path (impl \"planes\") only")
        need = {"qkv": "qkv", "geglu": "ff", "attn_out": "attn_out", "ff_out": "ff"}
        bad = [g for g in self.f16_weights if need[g] not in self.f16_inputs]
        if bad:
            raise ValueError(f"SplitPolicy.f16_weights {bad}: a layer's weights are rounded to fp16 only where its input is (f16_inputs)")

    def key(self):
        return (self.impl, tuple(sorted(self.f16_inputs)), tuple(sorted(self.f16_weights)), tuple(sorted(self.q8_convs)))

    def describe(self):
        return {"impl": self.impl, "f16_inputs": sorted(self.f16_inputs), "f16_weights": sorted(self.f16_weights), "q8_convs": sorted(self.q8_convs)}

    def __eq__(self, o):
        return isinstance(o, SplitPolicy) and self.key() == o.key()

    def __hash__(self):
        return hash(self.key())

    def __repr__(self):
        return (f"SplitPolicy(impl={self.impl!r}, f16_inputs={sorted(self.f16_inputs)}, f16_weights={sorted(self.f16_weights)}, "
                f"q8_convs={sorted(self.q8_convs)})")


UNET_POLICY = SplitPolicy()                                   # the UNets / ControlNet of Stage 2
ALL_SPLIT = SplitPolicy(f16_inputs=(), f16_weights=())                        # every product in three MFMAs, attention in the split kernels
VAE_POLICY = ALL_SPLIT                                        # the VAE (its single-head attentions are a third of the Stage-2 distance in fp16)

# Inside ``with f32_split(policy)`` the matrix products of fp32 tensors run in the split precision under that policy
# (``True`` = UNET_POLICY, ``False`` / ``None`` = off).  Entered by the owning network around its forward.
def f32_split(on):
    return tuning(policy=UNET_POLICY if on is True else (on if isinstance(on, SplitPolicy) else None))


def _policy():
    return _CTX.get().policy


def _split_fast():
    pol = _CTX.get().policy
    return pol is not None and pol.impl == "planes"


def precision_token():
    """Hashable name of the precision the current call runs in (None outside a split-precision network): cached intermediate tensors
    (the text context's K | V) are keyed on it."""
    pol = _CTX.get().policy
    return None if pol is None else pol.key()


def f16_group(group):
    """Does the current policy hand the inputs of layer group ``group`` over as fp16?"""
    if group is None:
        return False
    pol = _CTX.get().policy
    return pol is not None and pol.impl == "planes" and group in pol.f16_inputs


def q8_group(group):
    """Does the current policy run the 3x3 convolution behind norm group ``group`` with e4m3 cross terms (RSVLD_F16Q8)?"""
    if group is None:
        return False
    pol = _CTX.get().policy
    return pol is not None and pol.impl == "planes" and group in pol.q8_convs


class Q8Rows:
    """An activation in the RSVLD_F16Q8 row format: ``t`` is an fp16-typed tensor ``[B, H, W, 2, C]`` whose first plane is fp16(x) and whose
    second plane holds, per 32 channels, 64 bytes of e4m3 cross-term operands (include/rsvld_hip.h); only the halo convolution reads it."""
    __slots__ = ("t",)

    def __init__(self, t):
        self.t = t

    @property
    def shape(self):
        s = self.t.shape
        return tuple(s[:-2]) + (s[-1],)


class Planes:
    """The split form of an fp32 tensor ``[..., C]``: ``t`` is a bf16 tensor ``[..., 2, C]`` with plane 0 = lo, plane 1 = hi
    (hi = bf16(v), lo = bf16(v - hi)); contiguous planes are rows ``lo(C) | hi(C)``, the layout RSVLD_SPLIT kernels read and write.
    A tensor that only feeds matrix products travels like this (same bytes as fp32, nothing lost that the product would keep).
    Duck-types the few tensor members the network code touches between ops: shape, reshape, last-dim slices."""

    __slots__ = ("t", "_gn_part", "_nhwc")
    dtype = torch.float32      # the LOGICAL dtype: code that asks ``x.dtype == torch.float32`` means "the fp32 / split families"

    def __init__(self, t):
        self.t = t

    @property
    def shape(self):
        return tuple(self.t.shape[:-2]) + (self.t.shape[-1],)

    @property
    def device(self):
        return self.t.device

    @property
    def is_cuda(self):
        return self.t.is_cuda

    def dim(self):
        return self.t.dim() - 1

    def numel(self):
        return self.t.numel() // 2

    def element_size(self):
        return 4

    def is_contiguous(self):
        return self.t.is_contiguous()

    def contiguous(self):
        return self if self.t.is_contiguous() else Planes(self.t.contiguous())

    def reshape(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        return Planes(self.t.reshape(*shape[:-1], 2, self.t.shape[-1] if shape[-1] == -1 else shape[-1]))

    def __getitem__(self, idx):
        if not isinstance(idx, tuple):
            idx = (idx,)
        if Ellipsis in idx:
            i = idx.index(Ellipsis)
            idx = idx[:i] + (slice(None),) * (self.dim() - (len(idx) - 1)) + idx[i + 1:]
        idx = idx + (slice(None),) * (self.dim() - len(idx))
        return Planes(self.t[idx[:-1] + (slice(None), idx[-1])])

    def index_select(self, dim, index):
        return Planes(self.t.index_select(dim, index))

    def float(self):
        return self.f32()

    def f32(self):
        """-> the fp32 tensor hi + lo (rsvld_merge_planes)."""
        t = self.t.contiguous()
        Cc = t.shape[-1]
        out = torch.empty(tuple(t.shape[:-2]) + (Cc,), device=t.device, dtype=torch.float32)
        L.check(L.load().rsvld_merge_planes(_ptr(t), _ptr(out), t.numel() // (2 * Cc), Cc, _stream()), "rsvld_merge_planes")
        if out.dim() == 4:
            out._nhwc = True
        return out


def to_planes(x):
    """fp32 ``[..., C]`` (C % 8 == 0, contiguous) -> Planes (rsvld_split_planes); a Planes passes through."""
    if isinstance(x, Planes):
        return x
    _need_gpu(x)
    if x.dtype != torch.float32 or not x.is_contiguous() or x.shape[-1] % 8:
        raise L.RsvldError("to_planes: a contiguous fp32 tensor with C % 8 == 0 expected")
    Cc = x.shape[-1]
    t = torch.empty(tuple(x.shape[:-1]) + (2, Cc), device=x.device, dtype=torch.bfloat16)
    _launch("split_planes", 0.0, 8.0 * x.numel(), lambda: L.check(L.load().rsvld_split_planes(_ptr(x), _ptr(t), x.numel() // Cc, Cc, _stream()),
                                                                  "rsvld_split_planes"))
    return Planes(t)


def as_f32(x):
    return x.f32() if isinstance(x, Planes) else x


def maybe_planes(x):
    """In the split precision: the planes of an fp32 tensor that SEVERAL matrix products are about to read (split it once instead
    of once per product); any other mode / tensor passes through."""
    if _split_fast() and not isinstance(x, Planes) and x.dtype == torch.float32 and x.is_contiguous() and x.shape[-1] % 8 == 0:
        return to_planes(x)
    return x


# ----------------------------------------------------------------------------- launch profiling
class LaunchProfiler:
    """Brackets every C-ABI launch with HIP events on the launch stream (torch's current stream is
    the stream the kernels are enqueued on) and accumulates time / algorithmic FLOPs per kernel."""

    def __init__(self):
        self.records = []

    def run(self, name, flops, nbytes, fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        self.records.append((name, flops, nbytes, e0, e1))
        return out

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for name, flops, nbytes, e0, e1 in self.records:
            r = agg.setdefault(name, {"name": name, "ms": 0.0, "n": 0, "flops": 0.0, "bytes": 0.0})
            r["ms"] += e0.elapsed_time(e1)
            r["n"] += 1
            r["flops"] += flops
            r["bytes"] += nbytes
        return agg


def set_profiler(p):
    """Bracket every launch of this thread from now on (``None``: stop).  Scoped form: ``with ops.tuning(profiler=p):``."""
    set_defaults(profiler=p)


def _launch(name, flops, nbytes, fn):
    prof = _CTX.get().profiler
    if prof is None:
        return fn()
    return prof.run(name, flops, nbytes, fn)


# ----------------------------------------------------------------------------- weights
class PackedConv:
    """K-major 16-bit weights of a Conv2d / Linear: ``w[Cout_p, KH*KW*Cin_p]``, bias fp32."""

    __slots__ = ("w", "bias", "cin", "cout", "cin_p", "cout_p", "kh", "kw", "geglu", "w3", "w2", "w1", "wq8")

    def __init__(self, w, bias, cin, cout, cin_p, cout_p, kh, kw, geglu=False):
        self.w, self.bias = w, bias
        self.w3 = None     # split product path: bf16 triples [Cout_p][KH*KW][W_hi | W_lo | W_hi], packed on first use from the fp32 ``w``
        self.w1 = None     # the fp32 ``w`` rounded to fp16 (SplitPolicy.f16_weights: fp16 activation x fp16 weight, one MFMA), likewise
        self.w2 = None     # fp16 pairs [Cout_p][KH*KW][W_lo | W_hi] (dtype RSVLD_F16W2: fp16 activations), likewise
        self.wq8 = None    # RSVLD_F16Q8 rows [Cout_p][KH*KW][fp16(w) | e4m3 cross-term blocks], likewise
        self.cin, self.cout, self.cin_p, self.cout_p = cin, cout, cin_p, cout_p
        self.kh, self.kw, self.geglu = kh, kw, geglu


def pack_conv(weight, bias, dtype, device, cin_split=None, geglu=False):
    """fp32 ``[Cout, Cin, KH, KW]`` (or Linear ``[Cout, Cin]``) -> PackedConv on ``device``.

    ``cin_split=(C1, C2)``: the layer consumes the concatenation of two NHWC tensors whose channel
    counts are padded separately to multiples of 8.  ``geglu``: rows are re-ordered so that value j
    and gate j (rows j and j + Cout/2 of the reference weight, sgm/modules/attention.py:84-96)
    become adjacent output channels 2j, 2j+1.
    """
    # (the re-layout runs where the master weights live: on the device for a loaded network -- no 15 GB round trip over PCIe and no
    #  host-side permute of 3.9 B values when a precision is switched -- on the host for weights that are still there)
    w = weight.detach().to(torch.float32)
    if w.dim() == 2:
        w = w[:, :, None, None]
    cout, cin, kh, kw = w.shape
    b = None if bias is None else bias.detach().to(device=w.device, dtype=torch.float32)
    if geglu:
        half = cout // 2
        idx = torch.stack([torch.arange(half), torch.arange(half) + half], 1).reshape(-1).to(w.device)
        w = w[idx]
        b = None if b is None else b[idx]
    parts = [cin] if cin_split is None else list(cin_split)
    assert sum(parts) == cin
    w = w.permute(0, 2, 3, 1)  # [Cout, KH, KW, Cin]
    chunks, off = [], 0
    for c in parts:
        blk = w[..., off:off + c]
        if pad8(c) != c:
            blk = torch.nn.functional.pad(blk, (0, pad8(c) - c))
        chunks.append(blk)
        off += c
    w = torch.cat(chunks, dim=-1)
    cin_p = w.shape[-1]
    cout_p = pad8(cout) if not geglu else (cout + 15) // 16 * 16
    if cout_p != cout:
        w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 0, 0, cout_p - cout))
        if b is not None:
            b = torch.nn.functional.pad(b, (0, cout_p - cout))
    w = w.reshape(cout_p, kh * kw * cin_p).contiguous().to(device=device, dtype=dtype)
    b = None if b is None else b.contiguous().to(device)
    return PackedConv(w, b, cin, cout, cin_p, cout_p, kh, kw, geglu)


# ----------------------------------------------------------------------------- conv / linear
def conv2d(x, pc, *, x2=None, stride=1, pad=None, upsample=False, rowvec=None, residual=None,
           out_f32=False, act=L.ACT_NONE, alpha=1.0, beta=1.0, norm=None, stats=False, out_planes=False, out_group=None, norm_group=None,
           group=None):
    """NHWC convolution.  ``pad`` = int or (top, left, bottom, right).

    ``norm=(gamma, beta, groups, eps, silu)``: a GroupNorm(+SiLU) over the input ([x | x2]) precedes the
    conv.  For 3x3/stride-1 convs it is FUSED into the conv's input staging (rsvld_conv3x3_halo_nhwc): one
    statistics pass over x, no normalised tensor in HBM.  Otherwise it runs as its own kernels first.

    ``stats=True``: the caller will feed the output to a GroupNorm; when the halo kernel runs it also writes
    per-tile per-channel (sum, sumsq) of the output from its epilogue and attaches them to the returned tensor
    (``_gn_part``), so the consumer's ``norm=`` needs no statistics pass at all.

    ``out_planes=True`` (honoured in the split precision only, ignored otherwise): the output only feeds another matrix
    product (q|k|v, GEGLU, ...) and is returned as ``Planes`` -- or as fp16 when the policy hands the inputs of its consumer's
    layer group ``out_group`` over in fp16 (``SplitPolicy.f16_inputs``).

    fp16 ``x`` with weights packed in fp32 (a network in the split precision whose policy rounds this layer's input to fp16, or SR3's
    compute dtype "w2"): dtype RSVLD_F16W2, fp16 activation x fp16 weight pair [W_lo | W_hi], two MFMAs per product; in a
    split-precision network the output is fp32 (+ fp32 residual) unless ``out_planes`` asks for the fp16 hand-over."""
    _need_gpu(x, x2, pc.w, rowvec, residual)
    ctx = _CTX.get()
    if isinstance(x, Planes) or (x.dtype == torch.float32 and _split_fast()):
        return _conv2d_split(x, pc, x2=x2, stride=stride, pad=pad, upsample=upsample, rowvec=rowvec, residual=residual,
                             act=act, alpha=alpha, beta=beta, norm=norm, stats=stats, out_planes=out_planes, out_group=out_group,
                             norm_group=norm_group, group=group)
    if x.dtype == torch.float32:
        return _conv2d_f32(x, pc, x2=x2, stride=stride, pad=pad, upsample=upsample, rowvec=rowvec, residual=residual,
                           act=act, alpha=alpha, beta=beta, norm=norm)
    w2 = pc.w.dtype == torch.float32         # fp32 masters under a 16-bit activation: the weight-pair form
    w1res = False
    if w2:
        if x.dtype != torch.float16:
            raise L.RsvldError("conv2d: fp32-packed weights take fp16 activations (RSVLD_F16W2), fp32 tensors or planes")
        if _split_fast():
            out_f32 = not out_planes
            # the layer's weight group (SplitPolicy.f16_weights): named by the caller, or implied by the consumer of an fp16 output
            wg = group if group is not None else ({"attn": "qkv", "ff": "geglu"}.get(out_group) if not out_f32 else None)
            if wg is not None and ctx.policy is not None and wg in ctx.policy.f16_weights:
                if out_f32:
                    if pc.kh != 1 or pc.kw != 1 or x2 is not None:
                        raise L.RsvldError(f"conv2d: SplitPolicy.f16_weights {wg!r} with an fp32 output is a Linear layer's form (RSVLD_F16W1)")
                    w1res = True  # fp16 x fp16, ONE MFMA per product, fp32 out + fp32 residual (dtype RSVLD_F16W1)
                else:
                    w2 = False    # the plain fp16 kernels
    wt = _w1(pc) if w1res else _w2(pc) if w2 else (_w1(pc) if pc.w.dtype == torch.float32 else pc.w)
    B, H, W, Cin = x.shape
    Cin2 = 0 if x2 is None else x2.shape[-1]
    if Cin + Cin2 != pc.cin_p:
        raise L.RsvldError(f"conv2d: input channels {Cin}+{Cin2} != packed {pc.cin_p}")
    if pad is None:
        pad = pc.kh // 2
    if isinstance(pad, int):
        pt = pl = pb = pr = pad
    else:
        pt, pl, pb, pr = pad
    Hin, Win = (2 * H, 2 * W) if upsample else (H, W)
    Ho = (Hin + pt + pb - pc.kh) // stride + 1
    Wo = (Win + pl + pr - pc.kw) // stride + 1
    geglu = act == L.ACT_GEGLU
    c_out = pc.cout_p // 2 if geglu else pc.cout_p
    out = torch.empty((B, Ho, Wo, c_out), device=x.device, dtype=torch.float32 if out_f32 else x.dtype)
    if not x.is_contiguous() or (x2 is not None and not x2.is_contiguous()):
        raise L.RsvldError("conv2d: inputs must be contiguous NHWC")
    if residual is not None and (tuple(residual.shape) != tuple(out.shape) or not residual.is_contiguous()
                                 or (w2 and residual.dtype != out.dtype)):
        raise L.RsvldError("conv2d: residual must match the output shape (weight-pair form: and the output's type)")
    rv_stride = 0
    if rowvec is not None:
        if tuple(rowvec.shape) != (B, pc.cout_p) or rowvec.dtype != torch.float32 or rowvec.stride(1) != 1:
            raise L.RsvldError("conv2d: rowvec must be fp32 [B, Cout] with unit inner stride")
        rv_stride = rowvec.stride(0) if B > 1 else pc.cout_p
    d = L.ConvDesc(
        x=x.data_ptr(), x2=None if x2 is None else x2.data_ptr(), w=wt.data_ptr(),
        bias=None if pc.bias is None else pc.bias.data_ptr(),
        rowvec=None if rowvec is None else rowvec.data_ptr(),
        residual=None if residual is None else residual.data_ptr(), out=out.data_ptr(),
        B=B, H=H, W=W, Cin=Cin, Cin2=Cin2, Cout=pc.cout_p, KH=pc.kh, KW=pc.kw, stride=stride,
        pad_t=pt, pad_l=pl, Ho=Ho, Wo=Wo, upsample=int(upsample), dtype=L.F16W1 if w1res else L.F16W2 if w2 else _dt(x), out_f32=int(out_f32),
        act=act, alpha=alpha, beta=beta, rowvec_stride=rv_stride, plan_div=ctx.plan_div, tune=ctx.tune)
    sfx = "_w1" if w1res else "_w2" if w2 else ""
    lib = L.load()
    halo = ctx.use_halo and bool(lib.rsvld_conv3x3_halo_supported(C.byref(d)))
    Bp = -(-B // ctx.plan_div)            # batch rows of one planning unit: every plan decision below uses Bp / Mp
    if halo:
        bn = 64 if pc.cout_p <= 64 else 128
        halo = Bp * ((Ho + 7) // 8) * ((Wo + 31) // 32) * ((pc.cout_p + bn - 1) // bn) >= ctx.halo_min_wgs and not (upsample and norm is not None)
    if norm is not None and not halo:   # unfused: normalise into a (single) tensor, then convolve it
        gamma, nbeta, groups, eps, silu = norm
        xn = group_norm(x, gamma, nbeta, groups, eps, x2=x2, silu=silu)
        return conv2d(xn, pc, stride=stride, pad=pad, upsample=upsample, rowvec=rowvec, residual=residual,
                      out_f32=out_f32, act=act, alpha=alpha, beta=beta, stats=stats, out_planes=out_planes, out_group=out_group, group=group)
    esz = x.element_size()
    flops = 2.0 * B * Ho * Wo * pc.cout * pc.cin * pc.kh * pc.kw
    nbytes = (x.numel() + (0 if x2 is None else x2.numel()) + wt.numel()) * esz + out.numel() * out.element_size() \
        + (0 if residual is None else residual.numel() * residual.element_size())
    if halo:
        ab, silu = None, 0
        part1 = getattr(x, "_gn_part", None)
        part2 = None if x2 is None else getattr(x2, "_gn_part", None)
        if norm is not None and part1 is not None and (x2 is None or part2 is not None):
            gamma, nbeta, groups, eps, silu = norm     # statistics came with the producers' epilogues
            ab = torch.empty((B, Cin + Cin2, 2), device=x.device, dtype=torch.float32)
            _launch("groupnorm_ab_from_partials", 0.0, 0.0, lambda: L.check(lib.rsvld_groupnorm_scale_shift_from_partials(
                _ptr(part1[0]), part1[1], Cin, None if part2 is None else _ptr(part2[0]), 0 if part2 is None else part2[1],
                Cin2, _ptr(gamma), _ptr(nbeta), _ptr(ab), B, H * W, groups, eps, _stream()),
                "rsvld_groupnorm_scale_shift_from_partials"))
        elif norm is not None:
            gamma, nbeta, groups, eps, silu = norm
            ws = torch.empty(lib.rsvld_groupnorm_ws_bytes(B, H * W, Cin + Cin2, groups), device=x.device, dtype=torch.uint8)
            ab = torch.empty((B, Cin + Cin2, 2), device=x.device, dtype=torch.float32)
            _launch("groupnorm_stats(3 kernels)", 0.0, x.numel() * esz + (0 if x2 is None else x2.numel() * esz),
                    lambda: L.check(lib.rsvld_groupnorm_scale_shift(_ptr(x), _ptr(x2), _ptr(gamma), _ptr(nbeta), _ptr(ab), B,
                                                                    H * W, Cin, Cin2, groups, eps, _dt(x), _ptr(ws), _stream()),
                                    "rsvld_groupnorm_scale_shift"))
        name = ("conv_halo_64" if pc.cout_p <= 64 else "conv_halo_128") + sfx
        part_out = None
        if stats and (w2 or not out_f32):
            ntiles = ((Ho + 7) // 8) * ((Wo + 31) // 32)
            part_out = torch.empty((B, ntiles, pc.cout_p, 2), device=x.device, dtype=torch.float32)
        _launch(name + _detail(B, Ho, Wo, Cin, Cin2, pc, stride, upsample), flops, nbytes, lambda: L.check(lib.rsvld_conv3x3_halo_nhwc(C.byref(d), _ptr(ab), int(silu), _ptr(part_out),
                                                                                 _stream()), "rsvld_conv3x3_halo_nhwc"))
        if part_out is not None:
            out._gn_part = (part_out, ntiles)
        return out
    M = B * Ho * Wo
    Mp = -(-M // ctx.plan_div)
    if (pc.kh == 1 and pc.kw == 1 and stride == 1 and (pt, pl) == (0, 0) and not upsample and x2 is None and rowvec is None
            and (w2 or not out_f32) and Cin % 32 == 0 and pc.cout_p >= 256 and Mp >= 4096
            and ((Mp + 255) // 256) * ((pc.cout_p + 255) // 256) >= 128 and 256 * Cin * (4 if w2 else 2) < 2 ** 32
            and not (ctx.tune & L.TUNE_NO_GEMM256)):   # mirrors rsvld_gemm256_try in csrc/gemm.hip (profiler label only)
        variant = "gemm_256x256"
    elif pc.cout_p <= 32:
        variant = "conv_igemm_256x32"
    elif pc.cout_p <= 64:
        variant = "conv_igemm_128x64"
    else:   # mirrors dispatch_conv2 in csrc/conv_igemm.hip
        wg128 = ((Mp + 127) // 128) * ((pc.cout_p + 127) // 128)
        wg64 = ((Mp + 63) // 64) * ((pc.cout_p + 127) // 128)
        variant = "conv_igemm_64x64" if wg64 < 256 else ("conv_igemm_64x128" if wg128 < 256 else "conv_igemm_128x128")
    _launch(variant + sfx + _detail(B, Ho, Wo, Cin, Cin2, pc, stride, upsample), flops, nbytes, lambda: L.check(lib.rsvld_conv2d_nhwc(C.byref(d), _stream()), "rsvld_conv2d_nhwc"))
    if out_f32 and out.dim() == 4:
        out._nhwc = True     # an fp32 4-d tensor is otherwise taken for NCHW by the VAE's input adapter
    return out


def _conv2d_f32(x, pc, *, x2, stride, pad, upsample, rowvec, residual, act, alpha, beta, norm):
    """fp32 NHWC convolution with fp32 packed weights (rsvld_conv2d_nhwc_f32); a ``norm=`` GroupNorm runs first, unfused."""
    if pc.w.dtype != torch.float32:
        raise L.RsvldError("conv2d (fp32): weights must be packed in fp32 (the owning network's compute_dtype)")
    if norm is not None:
        gamma, nbeta, groups, eps, silu = norm
        x, x2 = group_norm(x, gamma, nbeta, groups, eps, x2=x2, silu=silu), None
    B, H, W, Cin = x.shape
    Cin2 = 0 if x2 is None else x2.shape[-1]
    if Cin + Cin2 != pc.cin_p:
        raise L.RsvldError(f"conv2d: input channels {Cin}+{Cin2} != packed {pc.cin_p}")
    if pad is None:
        pad = pc.kh // 2
    pt, pl, pb, pr = (pad,) * 4 if isinstance(pad, int) else pad
    Hin, Win = (2 * H, 2 * W) if upsample else (H, W)
    Ho = (Hin + pt + pb - pc.kh) // stride + 1
    Wo = (Win + pl + pr - pc.kw) // stride + 1
    c_out = pc.cout_p // 2 if act == L.ACT_GEGLU else pc.cout_p
    out = torch.empty((B, Ho, Wo, c_out), device=x.device, dtype=torch.float32)
    if not x.is_contiguous() or (x2 is not None and (not x2.is_contiguous() or x2.dtype != torch.float32)):
        raise L.RsvldError("conv2d (fp32): inputs must be contiguous fp32 NHWC")
    if residual is not None and (tuple(residual.shape) != tuple(out.shape) or not residual.is_contiguous()
                                 or residual.dtype != torch.float32):
        raise L.RsvldError("conv2d (fp32): residual must be fp32 and match the output shape")
    rv_stride = 0
    if rowvec is not None:
        if tuple(rowvec.shape) != (B, pc.cout_p) or rowvec.dtype != torch.float32 or rowvec.stride(1) != 1:
            raise L.RsvldError("conv2d: rowvec must be fp32 [B, Cout] with unit inner stride")
        rv_stride = rowvec.stride(0) if B > 1 else pc.cout_p
    d = L.ConvDesc(
        x=x.data_ptr(), x2=None if x2 is None else x2.data_ptr(), w=pc.w.data_ptr(),
        bias=None if pc.bias is None else pc.bias.data_ptr(), rowvec=None if rowvec is None else rowvec.data_ptr(),
        residual=None if residual is None else residual.data_ptr(), out=out.data_ptr(),
        B=B, H=H, W=W, Cin=Cin, Cin2=Cin2, Cout=pc.cout_p, KH=pc.kh, KW=pc.kw, stride=stride, pad_t=pt, pad_l=pl, Ho=Ho, Wo=Wo,
        upsample=int(upsample), dtype=L.F32, out_f32=1, act=act, alpha=alpha, beta=beta, rowvec_stride=rv_stride, plan_div=1,
        tune=L.TUNE_F32_SPLIT if _policy() is not None else 0)
    flops = 2.0 * B * Ho * Wo * pc.cout * pc.cin * pc.kh * pc.kw
    nbytes = 4.0 * (x.numel() + pc.w.numel() + out.numel() + (0 if residual is None else residual.numel()))
    _launch("conv_f32_split" if _policy() is not None else "conv_f32", flops, nbytes, lambda: L.check(L.load().rsvld_conv2d_nhwc_f32(C.byref(d), _stream()),
                                                       "rsvld_conv2d_nhwc_f32"))
    out._nhwc = True     # an fp32 4-d tensor is otherwise taken for NCHW by the VAE's input adapter
    return out



def _w3(pc):
    """bf16 weight triples of a PackedConv (fp32 K-major ``w``), packed once on the device."""
    if pc.w3 is None:
        if pc.w.dtype != torch.float32:
            raise L.RsvldError("split precision: weights must be packed in fp32 (the owning network's compute_dtype)")
        taps = pc.kh * pc.kw
        w3 = torch.empty((pc.cout_p, taps * 3 * pc.cin_p), device=pc.w.device, dtype=torch.bfloat16)
        L.check(L.load().rsvld_split_pack_weights(_ptr(pc.w), _ptr(w3), pc.cout_p, taps, pc.cin_p, _stream()), "rsvld_split_pack_weights")
        pc.w3 = w3
    return pc.w3


def _w1(pc):
    """The fp32 K-major ``w`` of a PackedConv rounded to fp16 (SplitPolicy.f16_weights), once."""
    if pc.w1 is None:
        pc.w1 = pc.w.half()
    return pc.w1


def _w2(pc):
    """fp16 weight pairs [W_lo | W_hi] of a PackedConv (fp32 K-major ``w``), packed once on the device."""
    if pc.w2 is None:
        if pc.w.dtype != torch.float32:
            raise L.RsvldError("weight pairs: weights must be packed in fp32 (the owning network's compute_dtype)")
        taps = pc.kh * pc.kw
        w2 = torch.empty((pc.cout_p, taps * 2 * pc.cin_p), device=pc.w.device, dtype=torch.float16)
        L.check(L.load().rsvld_pack_weight_pairs(_ptr(pc.w), _ptr(w2), pc.cout_p, taps, pc.cin_p, _stream()), "rsvld_pack_weight_pairs")
        pc.w2 = w2
    return pc.w2


def _wq8(pc):
    """RSVLD_F16Q8 weight rows of a PackedConv (fp32 K-major ``w``, input channels a multiple of 32), packed once on the device."""
    if pc.wq8 is None:
        if pc.w.dtype != torch.float32 or pc.cin_p % 32:
            raise L.RsvldError("e4m3 cross-term weights: fp32-packed weights with a multiple of 32 input channels expected")
        taps = pc.kh * pc.kw
        wq = torch.empty((pc.cout_p, taps * 2 * pc.cin_p), device=pc.w.device, dtype=torch.float16)
        L.check(L.load().rsvld_pack_weight_hq8(_ptr(pc.w), _ptr(wq), pc.cout_p, taps, pc.cin_p, _stream()), "rsvld_pack_weight_hq8")
        pc.wq8 = wq
    return pc.wq8


def to_q8rows(x):
    """fp32 ``[..., C]`` (C % 32 == 0, contiguous) -> Q8Rows (rsvld_split_hq8): the activation format of RSVLD_F16Q8."""
    _need_gpu(x)
    Cc = x.shape[-1]
    if x.dtype != torch.float32 or not x.is_contiguous() or Cc % 32:
        raise L.RsvldError("to_q8rows: contiguous fp32 with a multiple of 32 channels expected")
    t = torch.empty(tuple(x.shape[:-1]) + (2, Cc), device=x.device, dtype=torch.float16)
    _launch("split_q8", 0.0, 8.0 * x.numel(), lambda: L.check(L.load().rsvld_split_hq8(_ptr(x), _ptr(t), x.numel() // Cc, Cc, _stream()),
                                                            "rsvld_split_hq8"))
    return Q8Rows(t)


def _q8_conv_eligible(B, H, W, Cc, pc, stride, pad, upsample, act, out_planes):
    """Can rsvld_conv3x3_halo_nhwc run this layer as RSVLD_F16Q8 (and would the halo kernel be chosen for it in the split precision)?"""
    if pc.kh != 3 or pc.kw != 3 or stride != 1 or upsample or out_planes or act != L.ACT_NONE or pad not in (None, 1, (1, 1, 1, 1)):
        return False
    if Cc % 64 or pc.cout_p <= 64 or pc.cout_p % 8 or W < 16 or H < 4 or pc.w.dtype != torch.float32:
        return False
    ctx = _CTX.get()
    Bp = -(-B // ctx.plan_div)
    return ctx.use_halo and Bp * ((H + 7) // 8) * ((W + 31) // 32) * ((pc.cout_p + 127) // 128) >= ctx.split_halo_min_wgs


def _conv2d_q8(xq, pc, *, rowvec, residual, alpha, beta, stats):
    """3x3 / stride 1 / pad 1 on Q8Rows: fp32 out (+ fp32 residual), epilogue statistics for the next GroupNorm."""
    ctx = _CTX.get()
    B, H, W, Cc = xq.shape
    if Cc != pc.cin_p:
        raise L.RsvldError(f"conv2d (q8): input channels {Cc} != packed {pc.cin_p}")
    out = torch.empty((B, H, W, pc.cout_p), device=xq.t.device, dtype=torch.float32)
    if residual is not None and (isinstance(residual, Planes) or residual.dtype != torch.float32 or tuple(residual.shape) != tuple(out.shape)
                                 or not residual.is_contiguous()):
        raise L.RsvldError("conv2d (q8): residual must be fp32 and match the output shape")
    rv_stride = 0
    if rowvec is not None:
        if tuple(rowvec.shape) != (B, pc.cout_p) or rowvec.dtype != torch.float32 or rowvec.stride(1) != 1:
            raise L.RsvldError("conv2d: rowvec must be fp32 [B, Cout] with unit inner stride")
        rv_stride = rowvec.stride(0) if B > 1 else pc.cout_p
    wq = _wq8(pc)
    d = L.ConvDesc(
        x=xq.t.data_ptr(), x2=None, w=wq.data_ptr(), bias=None if pc.bias is None else pc.bias.data_ptr(),
        rowvec=None if rowvec is None else rowvec.data_ptr(), residual=None if residual is None else residual.data_ptr(), out=out.data_ptr(),
        B=B, H=H, W=W, Cin=Cc, Cin2=0, Cout=pc.cout_p, KH=3, KW=3, stride=1, pad_t=1, pad_l=1, Ho=H, Wo=W, upsample=0,
        dtype=L.F16Q8, out_f32=1, act=L.ACT_NONE, alpha=alpha, beta=beta, rowvec_stride=rv_stride, plan_div=ctx.plan_div, tune=ctx.tune)
    lib = L.load()
    if not lib.rsvld_conv3x3_halo_supported(C.byref(d)):
        raise L.RsvldError("conv2d (q8): shape not supported by the halo kernel (checked by _q8_conv_eligible)")
    part_out, ntiles = None, ((H + 7) // 8) * ((W + 31) // 32)
    if stats:
        part_out = torch.empty((B, ntiles, pc.cout_p, 2), device=out.device, dtype=torch.float32)
    flops = 2.0 * B * H * W * pc.cout * pc.cin * 9
    nbytes = 4.0 * (B * H * W * Cc + out.numel() + (0 if residual is None else residual.numel())) + 4.0 * pc.w.numel()
    _launch("conv_halo_128_q8" + _detail(B, H, W, Cc, 0, pc, 1, False), flops, nbytes, lambda: L.check(
        lib.rsvld_conv3x3_halo_nhwc(C.byref(d), None, 0, _ptr(part_out), _stream()), "rsvld_conv3x3_halo_nhwc"))
    if part_out is not None:
        out._gn_part = (part_out, ntiles)
    out._nhwc = True
    return out


def _gn_scale_shift_f32(x, x2, gamma, nbeta, groups, eps):
    """(scale, shift) fp32 ``[B, C1+C2, 2]`` of a GroupNorm over fp32 NHWC ``[x | x2]``: from the producers' epilogue partials when
    every source carries them (no pass over the tensors), else one statistics pass."""
    B, H, W, C1 = x.shape
    C2 = 0 if x2 is None else x2.shape[-1]
    lib = L.load()
    ab = torch.empty((B, C1 + C2, 2), device=x.device, dtype=torch.float32)
    part1 = getattr(x, "_gn_part", None)
    part2 = None if x2 is None else getattr(x2, "_gn_part", None)
    if part1 is not None and (x2 is None or part2 is not None):
        _launch("groupnorm_ab_from_partials", 0.0, 0.0, lambda: L.check(lib.rsvld_groupnorm_scale_shift_from_partials(
            _ptr(part1[0]), part1[1], C1, None if part2 is None else _ptr(part2[0]), 0 if part2 is None else part2[1],
            C2, _ptr(gamma), _ptr(nbeta), _ptr(ab), B, H * W, groups, eps, _stream()), "rsvld_groupnorm_scale_shift_from_partials"))
        return ab
    if not x.is_contiguous() or (x2 is not None and not x2.is_contiguous()):
        raise L.RsvldError("group_norm (split): contiguous fp32 NHWC inputs expected")
    ws = torch.empty(lib.rsvld_groupnorm_ws_bytes(B, H * W, C1 + C2, groups), device=x.device, dtype=torch.uint8)
    _launch("groupnorm_stats_split", 0.0, 4.0 * (x.numel() + (0 if x2 is None else x2.numel())), lambda: L.check(
        lib.rsvld_groupnorm_scale_shift_f32(_ptr(x), _ptr(x2), _ptr(gamma), _ptr(nbeta), _ptr(ab), B, H * W, C1, C2, groups, eps,
                                            _ptr(ws), _stream()), "rsvld_groupnorm_scale_shift_f32"))
    return ab


def _gn_apply_split(x, x2, ab, silu, planes, mod_scale1p=None, mod_shift=None, f16=False, q8=False):
    B, H, W, C1 = x.shape
    C2 = 0 if x2 is None else x2.shape[-1]
    Cc = C1 + C2
    mod_stride = 0
    if mod_scale1p is not None:
        mod_stride = mod_scale1p.stride(-2)
        if (mod_shift.stride(-2) != mod_stride or mod_scale1p.stride(-1) != 1 or mod_shift.stride(-1) != 1
                or mod_scale1p.dtype != torch.float32 or mod_shift.dtype != torch.float32):
            raise L.RsvldError("group_norm: modulation tensors must be fp32, share a row stride and be channel-contiguous")
    f16 = f16 and planes          # the fp16 hand-over replaces a planes output only
    if q8:                        # RSVLD_F16Q8 rows (the input of a convolution with e4m3 cross terms)
        out = torch.empty((B, H, W, 2, Cc), device=x.device, dtype=torch.float16)
        _launch("groupnorm_apply_q8", 0.0, 8.0 * B * H * W * Cc, lambda: L.check(L.load().rsvld_groupnorm_apply_split(
            _ptr(x), _ptr(x2), _ptr(out), _ptr(ab), _ptr(mod_scale1p), _ptr(mod_shift), mod_stride, B, H * W, C1, C2, int(silu), 3, _stream()),
            "rsvld_groupnorm_apply_split"))
        return Q8Rows(out)
    if f16:
        out = torch.empty((B, H, W, Cc), device=x.device, dtype=torch.float16)
    elif planes:
        out = torch.empty((B, H, W, 2, Cc), device=x.device, dtype=torch.bfloat16)
    else:
        out = torch.empty((B, H, W, Cc), device=x.device, dtype=torch.float32)
    _launch("groupnorm_apply_split", 0.0, (6.0 if f16 else 8.0) * B * H * W * Cc, lambda: L.check(L.load().rsvld_groupnorm_apply_split(
        _ptr(x), _ptr(x2), _ptr(out), _ptr(ab), _ptr(mod_scale1p), _ptr(mod_shift), mod_stride, B, H * W, C1, C2, int(silu),
        2 if f16 else int(not planes), _stream()), "rsvld_groupnorm_apply_split"))
    if f16:
        return out
    if planes:
        return Planes(out)
    out._nhwc = True
    return out


def _detail(B, Ho, Wo, Cin, Cin2, pc, stride, upsample):
    if not _CTX.get().profile_detail:
        return ""
    return f" [{B}x{Ho}x{Wo} {Cin}+{Cin2}->{pc.cout_p} k{pc.kh} s{stride}{' up' if upsample else ''}]"


def _conv2d_split(x, pc, *, x2, stride, pad, upsample, rowvec, residual, act, alpha, beta, norm, stats, out_planes, out_group=None,
                  norm_group=None, group=None):
    """The split-operand product path of conv2d / linear: bf16 planes in (split here when the caller hands fp32), weight triples,
    the 16-bit kernels with dtype RSVLD_SPLIT; fp32 (or Planes, or -- a 1x1 layer whose consumer group takes fp16 -- fp16) out,
    fp32 residual."""

    if norm is not None:      # GroupNorm(+SiLU) over [x | x2]: its own apply pass writes ONE planes tensor (no concat, no fp32 copy)
        gamma, nbeta, groups, eps, silu = norm
        if isinstance(x, Planes) or isinstance(x2, Planes):
            raise L.RsvldError("conv2d (split): norm= needs the fp32 tensors")
        ab = _gn_scale_shift_f32(x, x2, gamma, nbeta, groups, eps)
        if f16_group(norm_group):     # the policy hands this convolution's input over in fp16: the weight-pair form (fp32 out, fp32 residual)
            x16 = _gn_apply_split(x, x2, ab, silu, planes=True, f16=True)
            return conv2d(x16, pc, stride=stride, pad=pad, upsample=upsample, rowvec=rowvec, residual=residual, act=act, alpha=alpha,
                          beta=beta, stats=stats, out_planes=out_planes, out_group=out_group, group=group)
        if q8_group(norm_group) and _q8_conv_eligible(x.shape[0], x.shape[1], x.shape[2], x.shape[3] + (0 if x2 is None else x2.shape[3]), pc,
                                                      stride, pad, upsample, act, out_planes):
            # both operands keep ~22 bits, the two cross terms of the product run in e4m3 (RSVLD_F16Q8; SplitPolicy.q8_convs)
            return _conv2d_q8(_gn_apply_split(x, x2, ab, silu, planes=True, q8=True), pc, rowvec=rowvec, residual=residual, alpha=alpha,
                              beta=beta, stats=stats)
        x, x2 = _gn_apply_split(x, x2, ab, silu, planes=True), None
    ctx = _CTX.get()
    x = to_planes(x)
    x2 = None if x2 is None else to_planes(x2)
    B, H, W, Cin = x.shape
    Cin2 = 0 if x2 is None else x2.shape[-1]
    if Cin + Cin2 != pc.cin_p:
        raise L.RsvldError(f"conv2d: input channels {Cin}+{Cin2} != packed {pc.cin_p}")
    if pad is None:
        pad = pc.kh // 2
    pt, pl, pb, pr = (pad,) * 4 if isinstance(pad, int) else pad
    Hin, Win = (2 * H, 2 * W) if upsample else (H, W)
    Ho = (Hin + pt + pb - pc.kh) // stride + 1
    Wo = (Win + pl + pr - pc.kw) // stride + 1
    geglu = act == L.ACT_GEGLU
    c_out = pc.cout_p // 2 if geglu else pc.cout_p
    if not x.is_contiguous() or (x2 is not None and not x2.is_contiguous()):
        raise L.RsvldError("conv2d (split): inputs must be contiguous planes")
    if residual is not None:
        if out_planes:
            raise L.RsvldError("conv2d (split): a planes output takes no residual (the residual stream stays fp32)")
        if isinstance(residual, Planes) or residual.dtype != torch.float32 or tuple(residual.shape) != (B, Ho, Wo, c_out) \
                or not residual.is_contiguous():
            raise L.RsvldError("conv2d (split): residual must be fp32 and match the output shape")
    # fp16 hand-over: a Linear / 1x1 layer (the implicit-GEMM kernels write it) whose consumer's layer group takes fp16 inputs
    out_f16 = bool(out_planes and f16_group(out_group) and pc.kh == 1 and pc.kw == 1 and stride == 1 and x2 is None and not upsample)
    if out_f16:
        out = torch.empty((B, Ho, Wo, c_out), device=x.device, dtype=torch.float16)
    elif out_planes:
        out = torch.empty((B, Ho, Wo, 2, c_out), device=x.device, dtype=torch.bfloat16)
    else:
        out = torch.empty((B, Ho, Wo, c_out), device=x.device, dtype=torch.float32)
    rv_stride = 0
    if rowvec is not None:
        if tuple(rowvec.shape) != (B, pc.cout_p) or rowvec.dtype != torch.float32 or rowvec.stride(1) != 1:
            raise L.RsvldError("conv2d: rowvec must be fp32 [B, Cout] with unit inner stride")
        rv_stride = rowvec.stride(0) if B > 1 else pc.cout_p
    w3 = _w3(pc)
    d = L.ConvDesc(
        x=x.t.data_ptr(), x2=None if x2 is None else x2.t.data_ptr(), w=w3.data_ptr(),
        bias=None if pc.bias is None else pc.bias.data_ptr(), rowvec=None if rowvec is None else rowvec.data_ptr(),
        residual=None if residual is None else residual.data_ptr(), out=out.data_ptr(),
        B=B, H=H, W=W, Cin=Cin, Cin2=Cin2, Cout=pc.cout_p, KH=pc.kh, KW=pc.kw, stride=stride, pad_t=pt, pad_l=pl, Ho=Ho, Wo=Wo,
        upsample=int(upsample), dtype=L.SPLIT, out_f32=2 if out_f16 else int(not out_planes), act=act, alpha=alpha, beta=beta,
        rowvec_stride=rv_stride, plan_div=ctx.plan_div, tune=ctx.tune)
    lib = L.load()
    flops = 2.0 * B * Ho * Wo * pc.cout * pc.cin * pc.kh * pc.kw
    nbytes = 4.0 * (x.numel() + (0 if x2 is None else x2.numel()) + out.numel() / (2 if (out_planes or out_f16) else 1)
                    + (0 if residual is None else residual.numel())) + 6.0 * pc.w.numel()
    Bp = -(-B // ctx.plan_div)
    halo = ctx.use_halo and bool(lib.rsvld_conv3x3_halo_supported(C.byref(d)))
    if halo:
        bn = 64 if pc.cout_p <= 64 else 128
        halo = Bp * ((Ho + 7) // 8) * ((Wo + 31) // 32) * ((pc.cout_p + bn - 1) // bn) >= ctx.split_halo_min_wgs
    if halo:
        part_out = None
        if stats and not out_planes:
            ntiles = ((Ho + 7) // 8) * ((Wo + 31) // 32)
            part_out = torch.empty((B, ntiles, pc.cout_p, 2), device=x.device, dtype=torch.float32)
        _launch(("conv_halo_64_split" if pc.cout_p <= 64 else "conv_halo_128_split") + _detail(B, Ho, Wo, Cin, Cin2, pc, stride, upsample), flops, nbytes, lambda: L.check(
            lib.rsvld_conv3x3_halo_nhwc(C.byref(d), None, 0, _ptr(part_out), _stream()), "rsvld_conv3x3_halo_nhwc"))
        if part_out is not None:
            out._gn_part = (part_out, ntiles)
    else:
        M = B * Ho * Wo
        Mp = -(-M // ctx.plan_div)
        g256 = (pc.kh == 1 and pc.kw == 1 and stride == 1 and (pt, pl) == (0, 0) and not upsample and x2 is None and rowvec is None
                and Cin % 32 == 0 and pc.cout_p >= 256 and Mp >= 4096 and ((Mp + 255) // 256) * ((pc.cout_p + 255) // 256) >= 128
                and 256 * Cin * 6 < 2 ** 32 and not (ctx.tune & L.TUNE_NO_GEMM256))   # mirrors rsvld_gemm256_try
        _launch(("gemm_256x256_split" if g256 else "conv_igemm_split") + _detail(B, Ho, Wo, Cin, Cin2, pc, stride, upsample), flops, nbytes,
                lambda: L.check(lib.rsvld_conv2d_nhwc(C.byref(d), _stream()), "rsvld_conv2d_nhwc"))
    if out_f16:
        return out
    if out_planes:
        return Planes(out)
    out._nhwc = True
    return out


def linear(x, pc, *, residual=None, act=L.ACT_NONE, alpha=1.0, beta=1.0, out_planes=False, out_group=None, group=None):
    """``[..., Cin] -> [..., Cout]`` on token-major tensors (a 1x1 conv over rows).  ``group``: the layer's weight group
    (``SplitPolicy.f16_weights``) where the consumer of its output does not name it."""
    shp = x.shape
    rows = x.numel() // shp[-1]
    res = None if residual is None else residual.reshape(1, 1, rows, residual.shape[-1])
    y = conv2d(x.reshape(1, 1, rows, shp[-1]), pc, pad=0, residual=res, act=act, alpha=alpha, beta=beta, out_planes=out_planes,
               out_group=out_group, group=group)
    return y.reshape(*shp[:-1], y.shape[-1])


# ----------------------------------------------------------------------------- norms
def group_norm(x, gamma, beta, groups, eps, *, x2=None, silu=False, mod_scale1p=None, mod_shift=None, planes=False, group=None):
    """GroupNorm(+SiLU) over NHWC ``x`` (or the channel concat [x | x2]).  ``mod_scale1p`` / ``mod_shift``
    (ZeroSFT) may be channel slices of one stacked tensor: only their row stride must agree.
    ``planes=True`` (split precision only, ignored otherwise): the result only feeds a matrix product -> ``Planes``, or fp16 when
    the policy hands the inputs of that product's layer group ``group`` over in fp16."""
    if x.dtype == torch.float32 and _split_fast() and not isinstance(x, Planes):
        _need_gpu(x, x2, gamma, beta)
        ab = _gn_scale_shift_f32(x, x2, gamma, beta, groups, eps)
        return _gn_apply_split(x, x2, ab, silu, planes, as_f32(mod_scale1p) if mod_scale1p is not None else None,
                               as_f32(mod_shift) if mod_shift is not None else None, f16=f16_group(group))
    mod_stride = 0
    if mod_scale1p is not None:
        mod_stride = mod_scale1p.stride(-2)
        if mod_shift.stride(-2) != mod_stride or mod_scale1p.stride(-1) != 1 or mod_shift.stride(-1) != 1:
            raise L.RsvldError("group_norm: modulation tensors must share a row stride and be channel-contiguous")
    _need_gpu(x, x2, gamma, beta)
    if x.dtype == torch.float32:       # fp32 family: the two-source form concatenates first; statistics, then apply (+ modulation)
        if x2 is not None:
            x = concat_c(x, x2)
        return group_norm_apply(x, group_norm_stats(x, groups), gamma, beta, groups, eps, silu=silu,
                                mod_scale1p=mod_scale1p, mod_shift=mod_shift)
    B, H, W, C1 = x.shape
    C2 = 0 if x2 is None else x2.shape[-1]
    lib = L.load()
    ws = torch.empty(lib.rsvld_groupnorm_ws_bytes(B, H * W, C1 + C2, groups), device=x.device, dtype=torch.uint8)
    y = torch.empty((B, H, W, C1 + C2), device=x.device, dtype=x.dtype)
    nbytes = 3 * y.numel() * y.element_size()  # stats read + apply read + write
    _launch("groupnorm(3 kernels)", 0.0, nbytes, lambda: L.check(
        lib.rsvld_groupnorm_nhwc(_ptr(x), _ptr(x2), _ptr(y), _ptr(gamma), _ptr(beta), _ptr(mod_scale1p),
                                 _ptr(mod_shift), mod_stride, B, H * W, C1, C2, groups, eps, int(silu), _dt(x),
                                 _ptr(ws), _stream()), "rsvld_groupnorm_nhwc"))
    return y


def group_norm_stats(x, groups, *, x2=None):
    """-> fp32 ``[B, groups, 2]`` (mean, biased variance)."""
    _need_gpu(x, x2)
    B, H, W, C1 = x.shape
    C2 = 0 if x2 is None else x2.shape[-1]
    lib = L.load()
    if x.dtype == torch.float32 and _split_fast():
        if not x.is_contiguous() or (x2 is not None and not x2.is_contiguous()):
            raise L.RsvldError("group_norm_stats (split): contiguous fp32 NHWC tensors expected")
        ws = torch.empty(lib.rsvld_groupnorm_ws_bytes(B, H * W, C1 + C2, groups), device=x.device, dtype=torch.uint8)
        st = torch.empty((B, groups, 2), device=x.device, dtype=torch.float32)
        _launch("groupnorm_stats_split", 0.0, 4.0 * (x.numel() + (0 if x2 is None else x2.numel())), lambda: L.check(
            lib.rsvld_groupnorm_stats_f32_fast(_ptr(x), _ptr(x2), _ptr(st), B, H * W, C1, C2, groups, _ptr(ws), _stream()),
            "rsvld_groupnorm_stats_f32_fast"))
        return st
    if x.dtype == torch.float32:
        if x2 is not None or not x.is_contiguous():
            raise L.RsvldError("group_norm_stats (fp32): one contiguous NHWC tensor expected")
        ws = torch.empty(lib.rsvld_groupnorm_f32_ws_bytes(B, H * W, C1, groups), device=x.device, dtype=torch.uint8)
        st = torch.empty((B, groups, 2), device=x.device, dtype=torch.float32)
        _launch("groupnorm_stats_f32", 0.0, 4.0 * x.numel(), lambda: L.check(
            lib.rsvld_groupnorm_stats_f32(_ptr(x), _ptr(st), B, H * W, C1, groups, _ptr(ws), _stream()), "rsvld_groupnorm_stats_f32"))
        return st
    ws = torch.empty(lib.rsvld_groupnorm_ws_bytes(B, H * W, C1 + C2, groups), device=x.device, dtype=torch.uint8)
    st = torch.empty((B, groups, 2), device=x.device, dtype=torch.float32)
    L.check(lib.rsvld_groupnorm_stats(_ptr(x), _ptr(x2), _ptr(st), B, H * W, C1, C2, groups, _dt(x), _ptr(ws),
                                      _stream()), "rsvld_groupnorm_stats")
    return st


def group_norm_apply(x, stats, gamma, beta, groups, eps, *, x2=None, silu=False, mod_scale1p=None, mod_shift=None, planes=False):
    """``mod_*`` (ZeroSFT modulation) with supplied statistics exists in the fp32 family only.
    ``planes=True`` (split precision only): the result only feeds a matrix product -> ``Planes``."""
    _need_gpu(x, x2, stats, gamma, beta)
    if x.dtype == torch.float32 and _split_fast() and not isinstance(x, Planes):
        B, H, W, C1 = x.shape
        Cc = C1 + (0 if x2 is None else x2.shape[-1])
        ab = torch.empty((B, Cc, 2), device=x.device, dtype=torch.float32)
        L.check(L.load().rsvld_groupnorm_scale_shift_from_stats(_ptr(stats), _ptr(gamma), _ptr(beta), _ptr(ab), B, Cc, groups, eps,
                                                                _stream()), "rsvld_groupnorm_scale_shift_from_stats")
        return _gn_apply_split(x, x2, ab, silu, planes, mod_scale1p, mod_shift)
    if mod_scale1p is not None and x.dtype != torch.float32:
        raise L.RsvldError("group_norm_apply: modulation with supplied statistics is an fp32-family feature (use group_norm)")
    B, H, W, C1 = x.shape
    C2 = 0 if x2 is None else x2.shape[-1]
    y = torch.empty((B, H, W, C1 + C2), device=x.device, dtype=x.dtype)
    if x.dtype == torch.float32:
        if x2 is not None or not x.is_contiguous():
            raise L.RsvldError("group_norm_apply (fp32): one contiguous NHWC tensor expected")
        mod_stride = 0
        if mod_scale1p is not None:
            mod_stride = mod_scale1p.stride(-2)
            if (mod_shift.stride(-2) != mod_stride or mod_scale1p.stride(-1) != 1 or mod_shift.stride(-1) != 1
                    or mod_scale1p.dtype != torch.float32 or mod_shift.dtype != torch.float32):
                raise L.RsvldError("group_norm: modulation tensors must be fp32, share a row stride and be channel-contiguous")
        _launch("groupnorm_apply_f32", 0.0, 8.0 * x.numel(), lambda: L.check(L.load().rsvld_groupnorm_apply_f32(
            _ptr(x), _ptr(y), _ptr(stats), _ptr(gamma), _ptr(beta), _ptr(mod_scale1p), _ptr(mod_shift), mod_stride,
            B, H * W, C1, groups, eps, int(silu), _stream()), "rsvld_groupnorm_apply_f32"))
        y._nhwc = True
        return y
    L.check(L.load().rsvld_groupnorm_apply(_ptr(x), _ptr(x2), _ptr(y), _ptr(stats), _ptr(gamma), _ptr(beta), None,
                                           None, 0, B, H * W, C1, C2, groups, eps, int(silu), _dt(x), _stream()),
            "rsvld_groupnorm_apply")
    return y


def layer_norm(x, gamma, beta, eps=1e-5, planes=False, group=None):
    """``planes=True`` (split precision only, ignored otherwise): the result only feeds matrix products -> ``Planes``, or fp16 when
    the policy hands the inputs of those products' layer group ``group`` over in fp16."""
    _need_gpu(x, gamma, beta)
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    if x.dtype == torch.float32 and _split_fast():
        if not x.is_contiguous():
            raise L.RsvldError("layer_norm (split): contiguous rows expected")
        f16 = planes and f16_group(group)
        if f16:
            y = torch.empty(x.shape, device=x.device, dtype=torch.float16)
        elif planes:
            y = torch.empty(tuple(x.shape[:-1]) + (2, Cc), device=x.device, dtype=torch.bfloat16)
        else:
            y = torch.empty_like(x)
        _launch("layernorm_split", 0.0, (6.0 if f16 else 8.0) * x.numel(), lambda: L.check(L.load().rsvld_layernorm_split(
            _ptr(x), _ptr(y), _ptr(gamma), _ptr(beta), rows, Cc, eps, 2 if f16 else int(not planes), _stream()), "rsvld_layernorm_split"))
        return y if f16 else (Planes(y) if planes else y)
    y = torch.empty_like(x)
    if x.dtype == torch.float32:
        if not x.is_contiguous():
            raise L.RsvldError("layer_norm (fp32): contiguous rows expected")
        _launch("layernorm_f32", 0.0, 8.0 * x.numel(), lambda: L.check(L.load().rsvld_layernorm_f32(
            _ptr(x), _ptr(y), _ptr(gamma), _ptr(beta), rows, Cc, eps, _stream()), "rsvld_layernorm_f32"))
        return y
    L.check(L.load().rsvld_layernorm(_ptr(x), _ptr(y), _ptr(gamma), _ptr(beta), rows, Cc, eps, _dt(x), _stream()),
            "rsvld_layernorm")
    return y


# ----------------------------------------------------------------------------- attention
def attention(q, k, v, heads, scale=None):
    """q ``[B, Nq, heads*D]``, k/v ``[B, Nk, heads*D]`` (views with a token stride are fine, e.g.
    slices of a fused qkv tensor) -> ``[B, Nq, heads*D]`` contiguous.
    In the split precision the policy decides (``SplitPolicy.f16_inputs``): with "attn" the operands are fp16 (handed over as fp16 by
    their producers, or converted here) and the 16-bit kernels run; the result is fp16 when "attn_out" is set (its consumer then runs
    as fp16 x weight pairs) and planes otherwise.  Without "attn" (the VAE's policy: its single-head attentions are a rounding error
    of the image's time and a third of the mode's Stage-2 distance from the reference when run in fp16, 4.4e-4 -> 3.0e-4 max after
    50 steps) the fused split kernels run on planes."""
    if any(isinstance(t, Planes) for t in (q, k, v)) or (q.dtype == torch.float32 and _split_fast()):
        return _attention_split(q, k, v, heads, scale)
    _need_gpu(q, k, v)
    B, Nq, HD = q.shape
    Nk = k.shape[1]
    D = HD // heads
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    for t in (q, k, v):
        if t.stride(2) != 1:
            raise L.RsvldError("attention: last dim must be contiguous")
    out = torch.empty((B, Nq, HD), device=q.device, dtype=q.dtype)
    lib = L.load()
    flops = 4.0 * B * heads * Nq * Nk * D
    nbytes = (q.shape[0] * Nq * HD * 2 + 2 * B * Nk * HD) * q.element_size()
    if q.dtype == torch.float32:
        if k.dtype != torch.float32 or v.dtype != torch.float32:
            raise L.RsvldError("attention (fp32): q, k, v must all be fp32")
        fn = lib.rsvld_attention_f32_split if _policy() is not None else lib.rsvld_attention_f32
        _launch(f"attention_f32{'_split' if _policy() is not None else ''}_d{D}", flops, nbytes, lambda: L.check(
            fn(_ptr(q), _ptr(k), _ptr(v), _ptr(out), B, heads, Nq, Nk, D,
               q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1),
               out.stride(0), out.stride(1), scale, _stream()), "rsvld_attention_f32"))
        return out
    ws_bytes = lib.rsvld_attention_ws_bytes(B, heads, Nq, Nk, D, _CTX.get().plan_div)   # split-KV partials (D = 512, small grids)
    ws = torch.empty(ws_bytes, device=q.device, dtype=torch.uint8) if ws_bytes > 0 else None
    tune = _CTX.get().d64_kernel
    # (profiler group: the short cross-attention launches -- 77 text keys -- are a different kernel and a different regime than
    #  the self-attention of the same layer: kept apart so that the roofline of the dominant group describes ONE kind of launch)
    _launch(f"attention_d{D}" + ("_cross" if (D == 64 and Nk != Nq) else ""), flops, nbytes, lambda: L.check(
        lib.rsvld_attention_tuned(_ptr(q), _ptr(k), _ptr(v), _ptr(out), B, heads, Nq, Nk, D,
                                  q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1),
                                  out.stride(0), out.stride(1), scale, _dt(q), _CTX.get().plan_div, _ptr(ws), _stream(), tune),
        "rsvld_attention"))
    if _split_fast() and not f16_group("attn_out"):   # a split-precision network whose to_out wants planes (exact: 11 bits fit hi + lo)
        return _f16_to_planes(out)
    return out


def _split_gemm(xt, w3, out, M, K, N, out_f32, name):
    """``out[M, N] = x[M, K] w[N, K]^T`` in the split precision: ``xt`` contiguous planes ``[M, 2, K]``, ``w3`` triples ``[N, 3K]``,
    ``out`` fp32 ``[M, N]`` or planes ``[M, 2, N]`` (rsvld_conv2d_nhwc as a 1x1 layer; plans on the whole call)."""
    d = L.ConvDesc(x=xt.data_ptr(), x2=None, w=w3.data_ptr(), bias=None, rowvec=None, residual=None, out=out.data_ptr(),
                   B=1, H=1, W=M, Cin=K, Cin2=0, Cout=N, KH=1, KW=1, stride=1, pad_t=0, pad_l=0, Ho=1, Wo=M, upsample=0,
                   dtype=L.SPLIT, out_f32=int(out_f32), act=L.ACT_NONE, alpha=1.0, beta=1.0, rowvec_stride=0, plan_div=1, tune=_CTX.get().tune)
    _launch(name, 2.0 * M * K * N, 4.0 * M * K + 6.0 * N * K + 4.0 * M * N,
            lambda: L.check(L.load().rsvld_conv2d_nhwc(C.byref(d), _stream()), "rsvld_conv2d_nhwc"))


def _planes_to_f16(x):
    """Planes / fp32 ``[B, N, C]`` (channel slices of a fused planes tensor are read in place) -> contiguous fp16 ``[B, N, C]``."""
    if not isinstance(x, Planes):
        return x.to(torch.float16)
    t = x.t                                             # [B, N, 2, C]
    B, N, _, C = t.shape
    if t.stride(3) != 1 or t.stride(0) != N * t.stride(1):
        t = t.contiguous()
    out = torch.empty((B, N, C), device=t.device, dtype=torch.float16)
    _launch("planes_to_f16", 0.0, 6.0 * B * N * C, lambda: L.check(L.load().rsvld_planes_to_f16(
        _ptr(t), t.stride(1), t.stride(2), _ptr(out), C, B * N, C, _stream()), "rsvld_planes_to_f16"))
    return out


def _f16_to_planes(x):
    """contiguous fp16 ``[..., C]`` -> Planes (exact)."""
    C = x.shape[-1]
    t = torch.empty(tuple(x.shape[:-1]) + (2, C), device=x.device, dtype=torch.bfloat16)
    _launch("f16_to_planes", 0.0, 6.0 * x.numel(), lambda: L.check(L.load().rsvld_f16_to_planes(
        _ptr(x), C, _ptr(t), x.numel() // C, C, _stream()), "rsvld_f16_to_planes"))
    return Planes(t)


def _attention_split(q, k, v, heads, scale):
    """Planes / fp32 operands in a split-precision network.  Policy "attn": the operands leave the planes as fp16 and the 16-BIT attention
    kernels run (attn_d64c / attn_d512b at 1 100-1 240 TFLOP/s instead of the three-MFMA kernels at 400-440 effective; measured against
    the reference's CPU runs after 50 + 50 steps, rounding ONLY q, k, v, P costs 4e-5 in Stage 1 and 2.5e-4 in Stage 2, where rounding
    the weights costs 2e-3 and every activation 3e-3: DESIGN.md).  Otherwise: the fused split attention kernels."""
    if f16_group("attn") and q.shape[-1] % 8 == 0:
        shared = k is v
        q16, k16 = _planes_to_f16(q), _planes_to_f16(k)
        v16 = k16 if shared else _planes_to_f16(v)
        return attention(q16, k16, v16, heads, scale)     # (-> fp16 or planes, by "attn_out")
    return _attention_split_kernels(q, k, v, heads, scale)


def _attention_split_kernels(q, k, v, heads, scale):
    """Attention in the split precision -> ``Planes [B, Nq, heads*D]`` (its consumer is always a projection).
    D = 64: the fused flash kernel on planes (rsvld_attention_split_d64).  Other head sizes (single-head d = 512 of SR3 and the
    VAE): two split GEMMs around a row softmax per block of query rows -- S = Q K^T (fp32) -> P = softmax(scale S) (planes) ->
    O = P V -- with K and V re-packed once as the GEMMs' weight triples."""
    shared = k is v
    def prep(t):
        if isinstance(t, Planes):
            return t
        return to_planes(t.contiguous())
    q = prep(q)
    k = prep(k)
    v = k if shared else prep(v)
    _need_gpu(q.t, k.t, v.t)
    B, Nq, HD = q.shape
    Nk = k.shape[1]
    D = HD // heads
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    lib = L.load()
    out = torch.empty((B, Nq, 2, HD), device=q.t.device, dtype=torch.bfloat16)
    flops = 4.0 * B * heads * Nq * Nk * D
    if D == 64:
        for t in (q.t, k.t, v.t):
            if t.stride(3) != 1:
                raise L.RsvldError("attention (split): last dim must be contiguous")
        nbytes = 4.0 * (2 * B * Nq * HD + 2 * B * Nk * HD)
        _launch("attention_split_d64" + ("_cross" if Nk != Nq else ""), flops, nbytes, lambda: L.check(lib.rsvld_attention_split_d64(
            _ptr(q.t), _ptr(k.t), _ptr(v.t), _ptr(out), B, heads, Nq, Nk,
            q.t.stride(0), q.t.stride(1), q.t.stride(2), k.t.stride(0), k.t.stride(1), k.t.stride(2),
            v.t.stride(0), v.t.stride(1), v.t.stride(2), out.stride(0), out.stride(1), out.stride(2), scale, 0, _stream()),
            "rsvld_attention_split_d64"))
        return Planes(out)
    if D == 512 and heads == 1 and shared and Nq >= _CTX.get().split_d512_fused_min:
        # keys and values are ONE planes tensor (SR3's re-associated SelfAttention): the fused kernel, two waves per 32 query rows
        nbytes = 4.0 * (2 * B * Nq * HD + B * Nk * HD)
        _launch("attention_split_d512", flops, nbytes, lambda: L.check(lib.rsvld_attention_split_d512_shared(
            _ptr(q.t), _ptr(k.t), _ptr(out), B, Nq, Nk, q.t.stride(0), q.t.stride(1), q.t.stride(2),
            k.t.stride(0), k.t.stride(1), k.t.stride(2), out.stride(0), out.stride(1), out.stride(2), scale, 0, _stream()),
            "rsvld_attention_split_d512_shared"))
        return Planes(out)
    if D % 8:
        raise L.RsvldError("attention (split): head dim must be a multiple of 8")
    Nk_p = (Nk + 7) // 8 * 8
    rows_blk = max(256, min(Nq, (_CTX.get().split_attn_s_bytes // (4 * Nk_p)) // 256 * 256))
    dev = q.t.device
    for b in range(B):
        for h in range(heads):
            def head(t):      # contiguous planes [N, 2, D] of (b, h)
                th = t[b, :, :, h * D:(h + 1) * D]
                return th if th.is_contiguous() else th.contiguous()
            qh, kh = head(q.t), head(k.t)
            vh = kh if shared else head(v.t)
            w1 = torch.empty((Nk_p, 3 * D), device=dev, dtype=torch.bfloat16)
            L.check(lib.rsvld_planes_to_triple(_ptr(kh), _ptr(w1), Nk, Nk_p, D, 2 * D, _stream()), "rsvld_planes_to_triple")
            w2 = torch.empty((D, 3 * Nk_p), device=dev, dtype=torch.bfloat16)
            L.check(lib.rsvld_planes_transpose_triple(_ptr(vh), _ptr(w2), Nk, Nk_p, D, 2 * D, _stream()), "rsvld_planes_transpose_triple")
            oh = out[b] if heads == 1 else torch.empty((Nq, 2, D), device=dev, dtype=torch.bfloat16)
            for r0 in range(0, Nq, rows_blk):
                m = min(rows_blk, Nq - r0)
                s_blk = torch.empty((m, Nk_p), device=dev, dtype=torch.float32)
                _split_gemm(qh[r0:r0 + m], w1, s_blk, m, D, Nk_p, True, f"attention_split_gemm_qk_d{D}")
                p_blk = torch.empty((m, 2, Nk_p), device=dev, dtype=torch.bfloat16)
                _launch("attention_split_softmax", 0.0, 12.0 * m * Nk_p, lambda: L.check(lib.rsvld_softmax_rows_split(
                    _ptr(s_blk), _ptr(p_blk), m, Nk, Nk_p, Nk_p, scale, _stream()), "rsvld_softmax_rows_split"))
                del s_blk
                _split_gemm(p_blk, w2, oh[r0:r0 + m], m, Nk_p, D, False, f"attention_split_gemm_pv_d{D}")
                del p_blk
            if heads != 1:
                out[b, :, :, h * D:(h + 1) * D] = oh
    return Planes(out)


def gemv(w, x, bias=None):
    """``w [N, K]`` (16-bit, rows contiguous) times ONE activation row ``x [K]`` (+ ``bias [N]``) -> ``[N]``: the weight-streaming
    products of the caption pass's token loop (rsvld_gemv; HBM-bound, no tile, no MFMA)."""
    _need_gpu(w, x, bias)
    N, K = w.shape
    if x.numel() != K or w.stride(1) != 1 or w.stride(0) != K or not x.is_contiguous():
        raise L.RsvldError("gemv: w must be a contiguous [N, K] matrix and x a contiguous row of K elements")
    if x.dtype != w.dtype or (bias is not None and bias.dtype != w.dtype):
        raise L.RsvldError("gemv: w, x and bias must share one 16-bit dtype")
    y = torch.empty(N, device=w.device, dtype=w.dtype)
    lib = L.load()
    _launch("gemv", 2.0 * N * K, (N * K + K + N) * 2, lambda: L.check(
        lib.rsvld_gemv(_ptr(w), _ptr(x), _ptr(bias), _ptr(y), N, K, _dt(w), _stream()), "rsvld_gemv"))
    return y


def gemv_fused(w, x, bias=None, *, norm=None, residual=None, glu=False):
    """``gemv`` with the element-wise neighbours of a Llama decode step folded in (rsvld_gemv_fused): ``norm=(weight, eps)``: the product runs
    on RMSNorm(x) * weight; ``glu``: x holds ``[gate | up]`` (2 K elements) and the product runs on silu(gate) * up; ``residual [N]``: the
    result is residual + (w x + bias)."""
    _need_gpu(w, x, bias, residual)
    N, K = w.shape
    if x.numel() != (2 * K if glu else K) or w.stride(1) != 1 or w.stride(0) != K or not x.is_contiguous():
        raise L.RsvldError("gemv_fused: w must be a contiguous [N, K] matrix and x a contiguous row of K (glu: 2 K) elements")
    nw, eps = (None, 0.0) if norm is None else norm
    for t in (x, bias, nw, residual):
        if t is not None and t.dtype != w.dtype:
            raise L.RsvldError("gemv_fused: every operand shares the weights' 16-bit dtype")
    if residual is not None and (residual.numel() != N or not residual.is_contiguous()):
        raise L.RsvldError("gemv_fused: residual must be a contiguous row of N elements")
    y = torch.empty(N, device=w.device, dtype=w.dtype)
    lib = L.load()
    _launch("gemv", 2.0 * N * K, (N * K + K + N) * 2, lambda: L.check(
        lib.rsvld_gemv_fused(_ptr(w), _ptr(x), _ptr(bias), _ptr(nw), float(eps), _ptr(residual), int(glu), _ptr(y), N, K, _dt(w), _stream()),
        "rsvld_gemv_fused"))
    return y


def llama_decode_attention(qkv, cos, sin, pos, kcache, vcache, n_q, n_kv, scale, ws=None):
    """One decode step of grouped-query attention over a static cache (rsvld_llama_decode_attention): ``qkv`` = the new token's q | k | v rows,
    ``pos`` a DEVICE int64 scalar; the caches ``[n_kv, max_len, 128]`` are updated at ``pos``; ``ws`` (optional, re-usable): an fp32
    tensor of rsvld_llama_decode_attention_ws_bytes (scratch: no initial state).  A position outside ``[0, max_len)`` is clamped by the kernel.  -> ``[n_q * 128]``."""
    _need_gpu(qkv, cos, sin, pos, kcache, vcache)
    hd, max_len = kcache.shape[-1], kcache.shape[-2]
    if pos.dtype != torch.int64 or not (kcache.is_contiguous() and vcache.is_contiguous() and qkv.is_contiguous()):
        raise L.RsvldError("llama_decode_attention: contiguous caches / qkv and an int64 position on the device")
    lib = L.load()
    if ws is None:
        ws = torch.empty(int(lib.rsvld_llama_decode_attention_ws_bytes(n_q, n_kv, max_len)) // 4, device=qkv.device, dtype=torch.float32)
    out = torch.empty(n_q * hd, device=qkv.device, dtype=qkv.dtype)
    _launch("llama_decode_attention", 0.0, 0.0, lambda: L.check(
        lib.rsvld_llama_decode_attention(_ptr(qkv), _ptr(cos), _ptr(sin), _ptr(pos), _ptr(kcache), _ptr(vcache), _ptr(out), _ptr(ws), n_q, n_kv,
                                         hd, max_len, float(scale), _dt(qkv), _stream()), "rsvld_llama_decode_attention"))
    return out


# ----------------------------------------------------------------------------- small fp32 layers
def linear_small(x, w, b, act_in=0, act_out=0):
    """fp32 ``[rows, in] -> [rows, out]`` with torch nn.Linear weight layout."""
    _need_gpu(x, w, b)
    rows, in_f = x.shape
    out_f = w.shape[0]
    y = torch.empty((rows, out_f), device=x.device, dtype=torch.float32)
    L.check(L.load().rsvld_linear_small_f32(_ptr(x), _ptr(w), _ptr(b), _ptr(y), rows, in_f, out_f, act_in, act_out,
                                            _stream()), "rsvld_linear_small_f32")
    return y


def sinusoidal(t, dim, kind):
    _need_gpu(t)
    t = t.reshape(-1).contiguous().float()
    out = torch.empty((t.numel(), dim), device=t.device, dtype=torch.float32)
    L.check(L.load().rsvld_sinusoidal_embedding(_ptr(t), _ptr(out), t.numel(), dim, kind, _stream()),
            "rsvld_sinusoidal_embedding")
    return out


# ----------------------------------------------------------------------------- layout / elementwise
def nchw_to_nhwc(src, dtype, c_dst=None, c_off=0, out=None, scale=1.0):
    """fp32 NCHW (* scale) -> 16-bit (or fp32) NHWC with channels padded to ``c_dst`` (default: next multiple of 8)."""
    _need_gpu(src)
    B, Cc, H, W = src.shape
    src = src.contiguous().float()
    if out is None:
        c_dst = pad8(Cc) if c_dst is None else c_dst
        out = torch.empty((B, H, W, c_dst), device=src.device, dtype=dtype)
        zero = 1
    else:
        c_dst, zero = out.shape[-1], 0
    if out.dtype == torch.float32:
        L.check(L.load().rsvld_nchw_f32_to_nhwc_f32(_ptr(src), _ptr(out), B, Cc, H, W, c_dst, c_off, zero, scale, _stream()),
                "rsvld_nchw_f32_to_nhwc_f32")
        out._nhwc = True     # an fp32 4-d tensor is otherwise taken for NCHW by the VAE's input adapter
        return out
    L.check(L.load().rsvld_nchw_f32_to_nhwc(_ptr(src), _ptr(out), B, Cc, H, W, c_dst, c_off, zero, scale, _DT[out.dtype],
                                            _stream()), "rsvld_nchw_f32_to_nhwc")
    return out


def nhwc_to_nchw(src, channels=None, c_off=0):
    """NHWC (16-bit or fp32) -> fp32 NCHW, keeping channels [c_off, c_off+channels)."""
    _need_gpu(src)
    B, H, W, Cs = src.shape
    Cc = Cs - c_off if channels is None else channels
    out = torch.empty((B, Cc, H, W), device=src.device, dtype=torch.float32)
    f32 = src.dtype == torch.float32
    L.check(L.load().rsvld_nhwc_to_nchw_f32(_ptr(src), _ptr(out), B, Cc, H, W, Cs, c_off, int(f32),
                                            0 if f32 else _dt(src), _stream()), "rsvld_nhwc_to_nchw_f32")
    return out


def axpby(a, b, sa=1.0, sb=1.0):
    _need_gpu(a, b)
    out = torch.empty_like(a)
    if a.dtype == torch.float32:
        if b.dtype != torch.float32 or not (a.is_contiguous() and b.is_contiguous()) or a.numel() != b.numel():
            raise L.RsvldError("axpby (fp32): two contiguous fp32 tensors of one size expected")
        L.check(L.load().rsvld_axpby_f32(_ptr(a), _ptr(b), _ptr(out), a.numel(), sa, sb, _stream()), "rsvld_axpby_f32")
        if a.dim() == 4:
            out._nhwc = True
        return out
    L.check(L.load().rsvld_axpby(_ptr(a), _ptr(b), _ptr(out), a.numel(), sa, sb, _dt(a), _stream()), "rsvld_axpby")
    return out


def geglu(x):
    _need_gpu(x)
    Cc = x.shape[-1] // 2
    rows = x.numel() // x.shape[-1]
    out = torch.empty((*x.shape[:-1], Cc), device=x.device, dtype=x.dtype)
    L.check(L.load().rsvld_geglu(_ptr(x), _ptr(out), rows, Cc, _dt(x), _stream()), "rsvld_geglu")
    return out


def ddpm_step(x, eps_nhwc, noise, c_recip, c_recipm1, coef1, coef2, sigma, clip=True):
    """SR3 ancestral step on fp32 NCHW ``x`` with the UNet's fp32 NHWC eps (diffusion.py:142-175)."""
    _need_gpu(x, eps_nhwc, noise)
    B, Cc, H, W = x.shape
    out = torch.empty_like(x)
    L.check(L.load().rsvld_ddpm_step(_ptr(x), _ptr(eps_nhwc), _ptr(noise), _ptr(out), B, Cc, H, W,
                                     eps_nhwc.shape[-1], c_recip, c_recipm1, coef1, coef2, sigma, int(clip),
                                     _stream()), "rsvld_ddpm_step")
    return out


# ----------------------------------------------------------------------------- Stage-2 sampler / cache / VAE posterior
def denoiser_out(net_out_nhwc, inp, c_out, c_skip):
    """fp32 NHWC network output -> fp32 NCHW ``net*c_out + input*c_skip`` (denoiser.py:77-78)."""
    _need_gpu(net_out_nhwc, inp)
    B, Cc, H, W = inp.shape
    if net_out_nhwc.dtype != torch.float32 or tuple(net_out_nhwc.shape[:3]) != (B, H, W):
        raise L.RsvldError("denoiser_out: network output must be fp32 NHWC matching the input")
    out = torch.empty_like(inp)
    L.check(L.load().rsvld_denoiser_out(_ptr(net_out_nhwc), _ptr(inp), _ptr(out), B, Cc, H, W, net_out_nhwc.shape[-1],
                                        c_out, c_skip, _stream()), "rsvld_denoiser_out")
    return out


def lerp_f32(a, b, w):
    _need_gpu(a, b)
    if not (a.is_contiguous() and b.is_contiguous()):
        raise L.RsvldError("lerp_f32: contiguous fp32 tensors expected")
    out = torch.empty_like(a)
    L.check(L.load().rsvld_lerp_f32(_ptr(a), _ptr(b), _ptr(out), a.numel(), w, _stream()), "rsvld_lerp_f32")
    return out


def axpy_f32(x, y, s):
    """x + s*y on fp32 tensors; ``x=None`` gives s*y."""
    _need_gpu(x, y)
    out = torch.empty_like(y)
    L.check(L.load().rsvld_axpy_f32(_ptr(x), _ptr(y), _ptr(out), y.numel(), s, _stream()), "rsvld_axpy_f32")
    return out


def euler_step(x_hat, denoised, x_center, restore_w, sigma_hat, dt):
    _need_gpu(x_hat, denoised, x_center)
    out = torch.empty_like(x_hat)
    L.check(L.load().rsvld_euler_step(_ptr(x_hat), _ptr(denoised), _ptr(x_center), _ptr(out), x_hat.numel(),
                                      restore_w, sigma_hat, dt, _stream()), "rsvld_euler_step")
    return out


def tile_blend_accumulate(acc, cnt, tile, weights, y0, x0):
    """In place: ``acc[:, :, y0:y0+th, x0:x0+tw] += tile * weights``, ``cnt[...] += weights`` (fp32 NCHW; sampling.py:733-734)."""
    _need_gpu(acc, cnt, tile, weights)
    B, Cc, H, W = acc.shape
    th, tw = tile.shape[-2:]
    if (tuple(cnt.shape) != tuple(acc.shape) or tuple(tile.shape[:2]) != (B, Cc) or tuple(weights.shape) != (th, tw)
            or any(t.dtype != torch.float32 or not t.is_contiguous() for t in (acc, cnt, tile, weights))):
        raise L.RsvldError("tile_blend_accumulate: fp32 contiguous acc/cnt [B,C,H,W], tile [B,C,th,tw], weights [th,tw]")
    L.check(L.load().rsvld_tile_blend_accumulate(_ptr(acc), _ptr(cnt), _ptr(tile), _ptr(weights), B, Cc, H, W, int(y0), int(x0),
                                                 th, tw, _stream()), "rsvld_tile_blend_accumulate")


def tile_blend_finish(acc, cnt):
    """-> ``acc / cnt`` (sampling.py:735)."""
    _need_gpu(acc, cnt)
    out = torch.empty_like(acc)
    L.check(L.load().rsvld_tile_blend_finish(_ptr(acc), _ptr(cnt), _ptr(out), acc.numel(), _stream()), "rsvld_tile_blend_finish")
    return out


def absdiff_sums(a, b):
    """Per batch row: fp32 ``[rows, 2]`` = (sum|a-b|, sum|a|)  (DFBCache.py:98-112)."""
    _need_gpu(a, b)
    rows = a.shape[0]
    n = a.numel() // rows
    lib = L.load()
    out = torch.empty((rows, 2), device=a.device, dtype=torch.float32)
    if a.dtype == torch.float32:
        if b.dtype != torch.float32 or not (a.is_contiguous() and b.is_contiguous()) or a.numel() != b.numel():
            raise L.RsvldError("absdiff_sums (fp32): two contiguous fp32 tensors of one size expected")
        L.check(lib.rsvld_absdiff_sums_f32(_ptr(a), _ptr(b), _ptr(out), rows, n, _stream()), "rsvld_absdiff_sums_f32")
        return out
    ws = torch.empty(lib.rsvld_absdiff_ws_bytes(rows, n), device=a.device, dtype=torch.uint8)
    L.check(lib.rsvld_absdiff_sums(_ptr(a), _ptr(b), _ptr(out), rows, n, _dt(a), _ptr(ws), _stream()),
            "rsvld_absdiff_sums")
    return out


def gaussian_sample(moments, channels, noise, scale):
    """NHWC moments (16-bit or fp32) -> fp32 NCHW z; ``noise=None`` gives mode()."""
    _need_gpu(moments, noise)
    B, H, W, mc = moments.shape
    z = torch.empty((B, channels, H, W), device=moments.device, dtype=torch.float32)
    f32 = moments.dtype == torch.float32
    L.check(L.load().rsvld_gaussian_sample(_ptr(moments), _ptr(noise), _ptr(z), B, channels, H, W, mc, scale, int(f32),
                                           0 if f32 else _dt(moments), _stream()), "rsvld_gaussian_sample")
    return z


def wavelet_blur(img, radius, high_accum=None):
    _need_gpu(img, high_accum)
    B, Cc, H, W = img.shape
    low = torch.empty_like(img)
    L.check(L.load().rsvld_wavelet_blur(_ptr(img), _ptr(low), _ptr(high_accum), B * Cc, H, W, radius, _stream()),
            "rsvld_wavelet_blur")
    return low


def add_f32(a, b):
    _need_gpu(a, b)
    out = torch.empty_like(a)
    L.check(L.load().rsvld_add_f32(_ptr(a), _ptr(b), _ptr(out), a.numel(), _stream()), "rsvld_add_f32")
    return out


def adain(content, style):
    _need_gpu(content, style)
    B, Cc, H, W = content.shape
    out = torch.empty_like(content)
    ws = torch.empty(4 * B * Cc, device=content.device, dtype=torch.float32)
    L.check(L.load().rsvld_adain(_ptr(content), _ptr(style), _ptr(out), _ptr(ws), B * Cc, H * W, _stream()),
            "rsvld_adain")
    return out


def concat_c(a, b):
    _need_gpu(a, b)
    rows = a.numel() // a.shape[-1]
    out = torch.empty((*a.shape[:-1], a.shape[-1] + b.shape[-1]), device=a.device, dtype=a.dtype)
    if a.dtype == torch.float32:
        if b.dtype != torch.float32 or not (a.is_contiguous() and b.is_contiguous()):
            raise L.RsvldError("concat_c (fp32): two contiguous fp32 tensors expected")
        L.check(L.load().rsvld_concat_c_f32(_ptr(a), _ptr(b), _ptr(out), rows, a.shape[-1], b.shape[-1], _stream()), "rsvld_concat_c_f32")
        if a.dim() == 4:
            out._nhwc = True
        return out
    L.check(L.load().rsvld_concat_c(_ptr(a), _ptr(b), _ptr(out), rows, a.shape[-1], b.shape[-1], _dt(a), _stream()),
            "rsvld_concat_c")
    return out
