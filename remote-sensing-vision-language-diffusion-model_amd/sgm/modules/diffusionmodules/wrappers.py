"""Network wrapper (reference: sgm/modules/diffusionmodules/wrappers.py:68-110).

``ControlWrapper.forward(x, t, c, control_scale, fbcache_mode, partial_info)`` runs the ControlNet
and then the UNet, like the reference — with one structural difference: on an ``*_stage2`` call
the reference re-runs the ControlNet and throws the result away (the UNet takes ``control`` from
``partial_info``, SR_modules.py:694); here the ControlNet is not launched at all in that case.
``dtype`` is assigned by SR_backbone (SR_model.py:41) and selects the 16-bit compute type.
Inputs may be fp32 NCHW (reference layout; converted once) or already-packed 16-bit NHWC.
Full outputs are the UNet's fp32 NHWC epsilon ``[N,H,W,8]`` (4 channels + padding); the denoiser
turns them back into fp32 NCHW.

hipGraph replay (``use_graph``; SR_backbone turns it on for latents up to 128, where a denoiser call is ~2 000 launches of
tens of microseconds and the host cannot keep the queue full): the launches of one call -- ControlNet + UNet, or either
half of the cache split -- are captured once per (mode, shape, control scale, packed-weight version) over STATIC input
buffers (noisy latent, timestep, text context, vector, LQ latent) and replayed; the per-step inputs are copied in, the
per-image ones only when their source tensors change.  The second half is captured over the first half's static outputs
(``partial_info``), so the two graphs hand over in place.  Anything else (profiling, sub-batches of the per-image cache,
foreign ``partial_info``) runs eagerly."""
import torch
import torch.nn as nn

from .... import ops
from ...._lib import RsvldError

OPENAIUNETWRAPPER = "rsvld_amd.sgm.modules.diffusionmodules.wrappers.ControlWrapper"


class ControlWrapper(nn.Module):
    def __init__(self, diffusion_model, compile_model: bool = False, dtype=torch.float32):
        super().__init__()
        self.diffusion_model = diffusion_model
        self.control_model = None
        self.dtype = dtype

    def load_control_model(self, control_model):
        self.control_model = control_model

    def _compute_dtype(self):
        return torch.bfloat16 if self.dtype == torch.bfloat16 else torch.float16

    def _nhwc(self, t, dt):
        if t.dtype in (torch.float16, torch.bfloat16) and t.dim() == 4 and t.shape[-1] % 8 == 0:
            return t if t.dtype == dt else t.to(dt)
        return ops.nchw_to_nhwc(t, dt)

    use_graph = False
    _graphs = None

    def forward(self, x, t, c, control_scale=1, fbcache_mode="none", partial_info=None, **kwargs):
        if not x.is_cuda:
            raise RsvldError("ControlWrapper runs on the GPU only")
        dt = self._compute_dtype()
        for net in (self.diffusion_model, self.control_model):
            if net.compute_dtype != dt:          # the packed 16-bit weights are per dtype
                net.compute_dtype = dt
                net.invalidate_packed()
        if self.use_graph and ops._PROFILER is None and not kwargs and torch.cuda.is_current_stream_capturing() is False:
            out = self._forward_graph(x, t, c, control_scale, fbcache_mode, partial_info, dt)
            if out is not None:
                return out
        return self._forward_eager(x, t, c, control_scale, fbcache_mode, partial_info, dt, **kwargs)

    # ---------------------------------------------------------------------------------------------- hipGraph replay
    def _forward_graph(self, x, t, c, control_scale, mode, partial_info, dt):
        """-> the call's result from a replayed graph, or None when this call must run eagerly."""
        if mode not in ("none", "input_stage1", "input_stage2") or c.get("crossattn") is None or c.get("vector") is None:
            return None
        xh = self._nhwc(x, dt)
        unet, ctrl = self.diffusion_model, self.control_model
        if self._graphs is None:
            self._graphs = {}
        key = (mode, tuple(xh.shape), float(control_scale), dt, unet.pack_version, ctrl.pack_version, ops._PLAN_DIV)
        if mode == "input_stage2":      # captured over the static outputs of the matching first half
            first = self._graphs.get(("input_stage1",) + key[1:])
            if first is None or partial_info is None or partial_info.get("h") is not first["out"]["h"]:
                return None
        ent = self._graphs.get(key)
        if ent is None:
            self._graphs = {k: v for k, v in self._graphs.items() if k[4:6] == key[4:6]}      # drop captures of old weights
            ent = self._graphs[key] = self._capture(xh, t, c, control_scale, mode, partial_info, dt)
        # per-step inputs
        ent["x"].copy_(xh)
        ent["t"].copy_(t)
        # per-image inputs: copied when the source tensor (object or contents) changed
        for name, src in (("ctx", c["crossattn"]), ("y", c["vector"]), ("ctl", c.get("control"))):
            if mode == "input_stage2" and name == "ctl":
                continue
            tag = (id(src), src._version)
            if ent["tag"].get(name) != tag:
                ent[name].copy_(src)
                ent["tag"][name] = tag
                ent["keep"][name] = src           # keeps id() unique while the tag is in use
        ent["graph"].replay()
        out = ent["out"]
        if isinstance(out, dict):                 # consumers pop from ``hs``: hand out a fresh list every call
            out = dict(out, hs=list(out["hs"]))
        return out

    def _capture(self, xh, t, c, control_scale, mode, partial_info, dt):
        ctx_src, y_src, ctl_src = c["crossattn"], c["vector"], c.get("control")
        ent = {"x": torch.empty_like(xh), "t": torch.empty_like(t), "ctx": ctx_src.to(dt).contiguous().clone(),
               "y": y_src.detach().clone(), "ctl": None if ctl_src is None else ctl_src.detach().clone(), "tag": {}, "keep": {}}
        ent["x"].copy_(xh)
        ent["t"].copy_(t)
        sc = {"crossattn": ent["ctx"], "vector": ent["y"], "control": ent["ctl"]}
        nets = (self.diffusion_model, self.control_model)
        saved = [n.cache_context_kv for n in nets]
        for n in nets:                          # the text K/V projections become part of the graph (they read the static context)
            n.cache_context_kv = False
        def fresh(p):                       # the second half pops from ``hs``: every call gets its own list
            return None if p is None else dict(p, hs=list(p["hs"]))
        try:
            self._forward_eager(ent["x"], ent["t"], sc, control_scale, mode, fresh(partial_info), dt)   # warm-up: packs weights
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = self._forward_eager(ent["x"], ent["t"], sc, control_scale, mode, fresh(partial_info), dt)
        finally:
            for n, v in zip(nets, saved):
                n.cache_context_kv = v
        if isinstance(out, dict):
            out = dict(out, hs=list(out["hs"]))
        ent["graph"], ent["out"] = g, out
        return ent

    def _forward_eager(self, x, t, c, control_scale, fbcache_mode, partial_info, dt, **kwargs):
        context = c.get("crossattn", None)
        if context is not None and context.dtype != dt:
            cached = getattr(self, "_ctx_cast", None)
            if cached is None or cached[0] is not context or cached[1] != context._version:
                cached = self._ctx_cast = (context, context._version, context.to(dt).contiguous())
            context = cached[2]
        y = c.get("vector", None)
        xh = self._nhwc(x, dt)
        if "stage2" in fbcache_mode:
            control = None
        else:
            control = self.control_model(x=self._nhwc(c.get("control", None), dt), timesteps=t, xt=xh,
                                         context=context, y=y)
        return self.diffusion_model(xh, timesteps=t, context=context, y=y, control=control,
                                    control_scale=control_scale, fbcache_mode=fbcache_mode,
                                    partial_info=partial_info, **kwargs)
