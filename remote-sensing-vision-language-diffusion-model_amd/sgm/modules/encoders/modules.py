"""Conditioners (boundary of the hot path: their OUTPUTS are its inputs).

``PreparedConditioner`` is the reference's cached-embedding conditioner
(sgm/modules/encoders/modules.py:237-281) and runs unchanged in PyTorch.  The text towers
(FrozenCLIPEmbedder / FrozenOpenCLIPEmbedder2, :436-612) stay on stock PyTorch-ROCm per
BASELINE.json's north_star and need ``transformers`` / ``open_clip`` checkpoints that are not
available offline; ``GeneralConditionerWithControl`` therefore only wires embedders given to it.
"""
from typing import Dict, List, Optional

import torch
from torch import nn

from ...util import instantiate_from_config


class GeneralConditioner(nn.Module):
    OUTPUT_DIM2KEYS = {2: "vector", 3: "crossattn", 4: "concat", 5: "concat"}
    KEY2CATDIM = {"vector": 1, "crossattn": 2, "concat": 1}

    def __init__(self, emb_models):
        super().__init__()
        embedders = []
        for n, cfg in enumerate(emb_models):
            emb = instantiate_from_config(cfg)
            emb.is_trainable = cfg.get("is_trainable", False)
            emb.ucg_rate = cfg.get("ucg_rate", 0.0)
            if "input_key" in cfg:
                emb.input_key = cfg["input_key"]
            elif "input_keys" in cfg:
                emb.input_keys = cfg["input_keys"]
            else:
                raise KeyError(f"need either 'input_key' or 'input_keys' for embedder {emb.__class__.__name__}")
            emb.legacy_ucg_val = cfg.get("legacy_ucg_value", None)
            embedders.append(emb)
        self.embedders = nn.ModuleList(embedders)

    @torch.no_grad()
    def forward(self, batch: Dict, force_zero_embeddings: Optional[List] = None) -> Dict:
        output = {}
        force_zero_embeddings = force_zero_embeddings or []
        for emb in self.embedders:
            if getattr(emb, "input_key", None) is not None:
                out = emb(batch[emb.input_key])
            else:
                out = emb(*[batch[k] for k in emb.input_keys])
            for e in (out if isinstance(out, (list, tuple)) else [out]):
                key = "control_vector" if "control_vector" in getattr(emb, "input_key", "") else self.OUTPUT_DIM2KEYS[e.dim()]
                if getattr(emb, "input_key", None) in force_zero_embeddings:
                    e = torch.zeros_like(e)
                output[key] = torch.cat((output[key], e), self.KEY2CATDIM[key]) if key in output else e
        return output

    def get_unconditional_conditioning(self, batch_c, batch_uc=None, force_uc_zero_embeddings=None):
        c = self(batch_c)
        uc = self(batch_c if batch_uc is None else batch_uc, force_uc_zero_embeddings or [])
        return c, uc


class GeneralConditionerWithControl(GeneralConditioner):
    """Adds the pass-through of ``control`` (encoders/modules.py:184-234)."""

    def forward(self, batch: Dict, force_zero_embeddings: Optional[List] = None) -> Dict:
        output = super().forward(batch, force_zero_embeddings)
        output["control"] = batch["control"]
        return output


class PreparedConditioner(nn.Module):
    """Cached embeddings: ``cond_pth`` / ``un_cond_pth`` are torch-saved dicts {crossattn, vector}
    (or the dicts themselves); each call repeats them to the batch of ``batch['control']``."""

    def __init__(self, cond_pth, un_cond_pth=None):
        super().__init__()
        cond = torch.load(cond_pth) if isinstance(cond_pth, str) else cond_pth
        for k, v in cond.items():
            self.register_buffer(k, v)
        self.un_cond_pth = un_cond_pth
        if un_cond_pth is not None:
            unc = torch.load(un_cond_pth) if isinstance(un_cond_pth, str) else un_cond_pth
            for k, v in unc.items():
                self.register_buffer(k + "_uc", v)

    @torch.no_grad()
    def forward(self, batch: Dict, return_uc=False) -> Dict:
        n = batch["control"].shape[0]
        output = {}
        for k, v in self.state_dict().items():
            if k.endswith("_uc") != return_uc:
                continue
            output[k[:-3] if return_uc else k] = v.detach().clone().repeat(n, *[1] * (v.ndim - 1))
        output["control"] = batch["control"]
        return output

    def get_unconditional_conditioning(self, batch_c, batch_uc=None, force_uc_zero_embeddings=None):
        c = self(batch_c)
        uc = self(batch_c, return_uc=True) if self.un_cond_pth is not None else None
        return c, uc
