"""First-block feature cache ("dynamic inference acceleration") — reference: models/modules/DFBCache.py.

Same module-global context protocol (``cache_context`` / ``get_current_cache_context``,
:37-57) and the same ``MyCacheContext`` slots (``prev``, ``final_decode``, :59-69).  The similarity
test (:98-112)  diff = mean|prev - cur| / (mean|prev| + 1e-6)  is evaluated by
``rsvld_absdiff_sums`` (fp32 partial sums, fp64 merge, deterministic) and decided PER IMAGE: with
a CFG batch [uc_0..uc_{B-1}; c_0..c_{B-1}] image b owns rows b and B+b, which for B = 1 (the only
way the reference calls it, infer.py:172,199) is exactly the reference's whole-tensor mean.
"""
import contextlib

from ... import ops

_current_cache_context = None


def get_current_cache_context():
    return _current_cache_context


def set_current_cache_context(cache_context=None):
    global _current_cache_context
    _current_cache_context = cache_context


@contextlib.contextmanager
def cache_context(cache_context):
    global _current_cache_context
    old = _current_cache_context
    _current_cache_context = cache_context
    try:
        yield cache_context
    finally:
        _current_cache_context = old


class MyCacheContext:
    def __init__(self):
        self._buffers = {}
        self.prev = None
        self.final_decode = None

    def get_buffer(self, name):
        return self._buffers.get(name, None)

    def set_buffer(self, name, val):
        self._buffers[name] = val


def relative_l1(t1, t2, images=None):
    """Per-image relative L1 of two ``[2B, ...]`` (or ``[B, ...]``) 16-bit tensors -> python floats.
    One device->host copy of 2*rows floats (the reference syncs twice per step through .item())."""
    rows = t1.shape[0]
    sums = ops.absdiff_sums(t1, t2).cpu()
    n = t1.numel() // rows
    images = rows // 2 if images is None else images
    per = rows // images
    out = []
    for b in range(images):
        sd = sum(float(sums[b + j * images, 0]) for j in range(per))
        sa = sum(float(sums[b + j * images, 1]) for j in range(per))
        cnt = n * per
        out.append((sd / cnt) / (sa / cnt + 1e-6))
    return out


class RowSubset:
    """A row selection of a per-sample tensor that is NOT materialised: consumers that cache work per source tensor
    (the text-context K/V projections, sgm/modules/attention.py) keep their cache on ``full`` and select ``rows``."""

    def __init__(self, full, rows):
        self.full, self.rows = full, rows


def select_rows(t, rows):
    """``t[rows]`` for an activation, carrying the producer's GroupNorm partial statistics along (ops.conv2d attaches
    them as ``_gn_part``), so the sub-batch takes the same statistics path as the full batch."""
    out = t.index_select(0, rows)
    part = getattr(t, "_gn_part", None)
    if part is not None:
        out._gn_part = (part[0].index_select(0, rows), part[1])
    return out


def select_partial_info(p, rows):
    """The ``partial_info`` dict of ``input_stage1`` (SR_modules.py:674-683) restricted to batch rows ``rows``."""
    ctx = p["context"]
    return dict(p, h=select_rows(p["h"], rows), hs=[select_rows(t, rows) for t in p["hs"]],
                emb={k: v.index_select(0, rows) for k, v in p["emb"].items()},
                context=None if ctx is None else RowSubset(ctx, rows),
                control=None if p["control"] is None else [select_rows(t, rows) for t in p["control"]])


def are_two_tensors_similar(t1, t2, *, threshold, parallelized=False):
    """Reference signature (:98-112): whole-tensor decision -> (bool, diff)."""
    diff = relative_l1(t1, t2, images=1)[0]
    return diff < threshold, diff


def get_can_use_cache_multi(first_residual, threshold, parallelized=False):
    """Reference signature (:115-134)."""
    context = get_current_cache_context()
    if context.prev is None:
        return False, threshold
    use_cache, diff = are_two_tensors_similar(context.prev, first_residual, threshold=threshold)
    return (True, diff) if use_cache else (False, diff)
