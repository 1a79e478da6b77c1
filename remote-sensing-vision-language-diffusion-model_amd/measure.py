"""Measurement hooks of bench.py, kept out of the reference-compatible call signatures.

    with measure.hooks(model, stamp=phases, max_steps=2):
        model.just_sampling(...)            # or GaussianDiffusion.super_resolution(...)

``stamp(name)`` is called at every phase border of the sampling entry point; ``max_steps`` truncates its sampler
loop (the per-image fixed part still runs in full).  Outside the ``with`` block neither can be reached."""
import contextlib


@contextlib.contextmanager
def hooks(model, stamp=None, max_steps=None):
    if not hasattr(model, "_measure"):
        raise TypeError(f"{type(model).__name__} has no measurement hooks")
    old = model._measure
    model._measure = (stamp, max_steps)
    try:
        yield model
    finally:
        model._measure = old
