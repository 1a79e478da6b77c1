"""Runtime shared by the Stage-2 networks (UNet, ControlNet, VAE): lazy 16-bit weight packing and
the stacked timestep-embedding projection.

A ``HipNet`` is the root nn.Module of one network.  Its layers are plain parameter containers
named like the reference's (so reference checkpoints load unchanged); their ``run(rt, ...)``
methods take the root as ``rt`` and issue kernels of librsvld_hip.so through ``rsvld_amd.ops``.
"""
import torch
from torch import nn

from . import ops
from ._lib import RsvldError


class HipNet(nn.Module):
    compute_dtype = torch.float16
    cache_context_kv = True   # reuse cross-attention K/V while the SAME context tensor object is passed

    def __init__(self):
        super().__init__()
        self._pk = {}
        self._pk_other = {}      # packed weights of the compute dtypes this network was switched away from

    # ---- cache invalidation whenever the fp32 master parameters move or change.  ``pack_version`` counts them: captured
    # hipGraphs hold raw pointers into the packed tensors and key themselves on it.
    pack_version = 0

    def invalidate_packed(self):
        self._pk = {}
        self._pk_other = {}
        self.pack_version += 1

    def set_compute_dtype(self, dt):
        """Switch the operand type (fp16 / bf16: 16-bit kernels, fp32: the fp32-operand family).  The packed weights of the
        previous type are kept, so switching back and forth (precision A/B runs, tests) packs each type once."""
        if dt == self.compute_dtype:
            return
        self._pk_other[self.compute_dtype] = self._pk
        self.compute_dtype = dt
        self._pk = self._pk_other.pop(dt, {})
        self.pack_version += 1

    def load_state_dict(self, *a, **k):
        self.invalidate_packed()
        return super().load_state_dict(*a, **k)

    def _load_from_state_dict(self, *a, **k):
        self.invalidate_packed()
        return super()._load_from_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self.invalidate_packed()
        return super()._apply(fn, *a, **k)

    # ---- packing
    def _dev(self, t):
        if t.device.type != "cuda":
            raise RsvldError(f"{type(self).__name__}: parameters must be on the GPU (there is no CPU execution path)")
        return t.device

    def pk(self, m, tag="", **kw):
        """PackedConv of a Conv2d / Linear ``m`` (packed on first use)."""
        key = (id(m), tag)
        p = self._pk.get(key)
        if p is None:
            p = self._pk[key] = ops.pack_conv(m.weight, m.bias, self.compute_dtype, self._dev(m.weight), **kw)
        return p

    def pk_cat(self, mods, tag, **kw):
        """One PackedConv for several Linear/Conv layers that share their input (rows concatenated):
        q|k|v of a self-attention, k|v of a cross-attention, gamma|beta of ZeroSFT."""
        key = (tuple(id(m) for m in mods), tag)
        p = self._pk.get(key)
        if p is None:
            w = torch.cat([m.weight.detach().float() for m in mods], 0)      # (stays on the masters' device: pack_conv re-lays it there)
            if all(m.bias is None for m in mods):
                b = None
            else:
                b = torch.cat([torch.zeros(m.weight.shape[0], device=w.device) if m.bias is None else m.bias.detach().float()
                               for m in mods], 0)
            p = self._pk[key] = ops.pack_conv(w, b, self.compute_dtype, self._dev(mods[0].weight), **kw)
        return p

    # ---- stacked emb projections: every ResBlock's emb_layers Linear in ONE launch
    def emb_rows(self, emb):
        """emb fp32 [B, E] -> {id(resblock): fp32 view [B, Cout]} = Linear(SiLU(emb)) for all ResBlocks."""
        ent = self._pk.get("emb_stack")
        if ent is None:
            offs, ws, bs, off = {}, [], [], 0
            for m in self.modules():
                lin = getattr(m, "_emb_linear", None)
                if lin is not None:
                    offs[id(m)] = (off, lin.out_features)
                    ws.append(lin.weight.detach().float())
                    bs.append(lin.bias.detach().float())
                    off += lin.out_features
            dev = self._dev(ws[0])
            ent = self._pk["emb_stack"] = (torch.cat(ws, 0).contiguous().to(dev), torch.cat(bs, 0).contiguous().to(dev), offs)
        w, b, offs = ent
        table = ops.linear_small(emb, w, b, act_in=1)
        return {k: table[:, o:o + n] for k, (o, n) in offs.items()}
