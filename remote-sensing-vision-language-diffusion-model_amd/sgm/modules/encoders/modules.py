"""Conditioners (boundary of the hot path: their OUTPUTS are its inputs).

``PreparedConditioner`` is the reference's cached-embedding conditioner
(sgm/modules/encoders/modules.py:237-281).  ``GeneralConditionerWithControl`` wires the live
embedders of model_configs/juggernautXL.yaml:67-105: ``FrozenCLIPEmbedder`` (CLIP-L, hidden layer
11 -> crossattn[...,:768]), ``FrozenOpenCLIPEmbedder2`` (OpenCLIP bigG, penultimate layer ->
crossattn[...,768:] and the pooled vector) and three ``ConcatTimestepEmbedderND`` (size / crop /
target size -> 3 x 512 entries of ``vector``).  The text towers stay stock PyTorch-ROCm modules
(BASELINE.json north_star): they run twice per image on 77 tokens.  Their third-party packages
(``transformers`` / ``open_clip``) and checkpoints are resolved at construction, or injected
(``tokenizer=`` / ``transformer=`` / ``model=`` / ``tokenize=``) -- which is how the offline parity
tests drive them with small seeded towers (tests/golden/gen_cond_golden.py).
"""
import math
from typing import Dict, List, Optional

import torch
from torch import nn

from ...util import instantiate_from_config


class AbstractEmbModel(nn.Module):
    """encoders/modules.py:28-68: the three attributes GeneralConditioner sets on every embedder."""

    def __init__(self):
        super().__init__()
        self._is_trainable = None
        self._ucg_rate = None
        self._input_key = None

    is_trainable = property(lambda self: self._is_trainable, lambda self, v: setattr(self, "_is_trainable", v))
    ucg_rate = property(lambda self: self._ucg_rate, lambda self, v: setattr(self, "_ucg_rate", v))
    input_key = property(lambda self: self._input_key, lambda self, v: setattr(self, "_input_key", v))


def _timestep_embedding(t, dim, max_period=10000):
    """sgm/modules/diffusionmodules/util.py:206-230 in plain torch (the conditioner runs where its inputs live;
    the denoiser's own timestep embedding is the HIP kernel behind diffusionmodules.util.timestep_embedding)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half).to(device=t.device)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


class Timestep(nn.Module):
    """openaimodel.py:491-497"""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, t):
        return _timestep_embedding(t, self.dim)


class ConcatTimestepEmbedderND(AbstractEmbModel):
    """encoders/modules.py:1031-1047: embeds each entry of ``[b, dims]`` independently and concatenates -> ``[b, dims*outdim]``."""

    def __init__(self, outdim):
        super().__init__()
        self.timestep = Timestep(outdim)
        self.outdim = outdim

    def forward(self, x):
        if x.ndim == 1:
            x = x[:, None]
        if x.ndim != 2:
            raise ValueError("ConcatTimestepEmbedderND expects [b] or [b, dims]")
        b, dims = x.shape
        emb = self.timestep(x.reshape(b * dims))
        return emb.reshape(b, dims * self.outdim)


def _freeze(module):
    module.eval()
    for q in module.parameters():
        q.requires_grad_(False)
    return module


class FrozenCLIPEmbedder(AbstractEmbModel):
    """CLIP-L text tower of the conditioner (yaml: ``layer: hidden, layer_idx: 11``; reference encoders/modules.py:436-499,
    whose constructor signature and attribute names -- ``tokenizer``, ``transformer`` = checkpoint prefix -- are kept).

    One forward of a HuggingFace ``CLIPTextModel`` on the 77-token padded prompt; what comes back is selected by a tiny
    table instead of a branch ladder: ``last`` = final-norm states, ``pooled`` = the pooled vector as a length-1 sequence,
    ``hidden`` = hidden state ``layer_idx`` of the ``[embeddings, block 1 .. block 12]`` list.  ``always_return_pooled``
    appends the pooled vector.  ``version`` (or ``CKPT_PTH.SDXL_CLIP1_PATH`` when that module exists) names the checkpoint;
    ``tokenizer`` / ``transformer`` may be injected (the offline parity tests do)."""

    LAYERS = ["last", "pooled", "hidden"]

    def __init__(self, version="openai/clip-vit-large-patch14", device="cuda", max_length=77, freeze=True, layer="last",
                 layer_idx=None, always_return_pooled=False, tokenizer=None, transformer=None):
        super().__init__()
        if layer not in self.LAYERS:
            raise ValueError(f"layer must be one of {self.LAYERS}")
        if layer == "hidden" and (layer_idx is None or abs(layer_idx) > 12):
            raise ValueError("layer='hidden' needs 0 <= |layer_idx| <= 12")
        if tokenizer is None or transformer is None:
            from transformers import CLIPTextModel, CLIPTokenizer
            try:
                from CKPT_PTH import SDXL_CLIP1_PATH as src
            except ImportError:
                src = None
            src = src or version        # a CKPT_PTH that sets SDXL_CLIP1_PATH = None means "use `version`" (reference modules.py:453-454)
            tokenizer = tokenizer or CLIPTokenizer.from_pretrained(src)
            transformer = transformer or CLIPTextModel.from_pretrained(src)
        self.tokenizer, self.transformer = tokenizer, transformer
        self.device, self.max_length = device, max_length
        self.layer, self.layer_idx, self.return_pooled = layer, layer_idx, always_return_pooled
        if freeze:
            self.freeze()

    def freeze(self):
        _freeze(self)

    @torch.no_grad()
    def forward(self, text):
        ids = self.tokenizer(text, padding="max_length", truncation=True, max_length=self.max_length,
                             return_tensors="pt")["input_ids"].to(self.device)
        res = self.transformer(input_ids=ids, output_hidden_states=(self.layer == "hidden"))
        pick = {"last": lambda: res.last_hidden_state,
                "pooled": lambda: res.pooler_output.unsqueeze(1),
                "hidden": lambda: res.hidden_states[self.layer_idx]}
        z = pick[self.layer]()
        return (z, res.pooler_output) if self.return_pooled else z

    def encode(self, text):
        return self(text)        # through nn.Module.__call__ (hooks), as the reference does


class FrozenOpenCLIPEmbedder2(AbstractEmbModel):
    """OpenCLIP bigG text tower of the conditioner (yaml: ``layer: penultimate, always_return_pooled: True, legacy: False``;
    reference encoders/modules.py:502-612, constructor signature and ``model`` = checkpoint prefix kept).

    ``_states(tokens)`` walks the residual blocks ONCE and returns the pair (input of the last block, output of the last
    block), batch-first.  The selected one is returned as it is -- NOT through ``ln_final``: only the pooled vector is
    normalised, ``pooled = ln_final(last)[eot] @ text_projection`` with eot = the highest token id of a row (:571-585).
    ``legacy=True`` (the reference's default, not the yaml's) returns ``ln_final`` of the selected state and has no pooled
    output.

    ``model`` must expose open_clip's text-tower attributes: ``token_embedding``, ``positional_embedding``,
    ``transformer.resblocks`` (called as ``r(x, attn_mask=...)`` on sequence-first tensors), ``ln_final``,
    ``text_projection``, ``attn_mask``.  Without ``model`` / ``tokenize`` the ``open_clip`` package builds them (the
    reference pins the ``laion2b_s39b_b160k`` weights whatever ``version`` says, :524-528)."""

    LAYERS = ["pooled", "last", "penultimate"]

    def __init__(self, arch="ViT-H-14", version="laion2b_s32b_b79k", device="cuda", max_length=77, freeze=True, layer="last",
                 always_return_pooled=False, legacy=True, model=None, tokenize=None):
        super().__init__()
        if layer not in self.LAYERS:
            raise ValueError(f"layer must be one of {self.LAYERS}")
        if layer == "pooled":
            raise NotImplementedError("layer='pooled' is listed but not implemented by the reference either (:545-546)")
        if model is None or tokenize is None:
            try:
                import open_clip
            except ImportError as e:
                raise ImportError("FrozenOpenCLIPEmbedder2 needs the open_clip package (or model= / tokenize= passed in); "
                                  "use PreparedConditioner with cached embeddings otherwise") from e
            if model is None:
                model = open_clip.create_model_and_transforms(arch, device=torch.device("cpu"), pretrained="laion2b_s39b_b160k")[0]
                del model.visual
            tokenize = tokenize or open_clip.tokenize
        self.model, self.tokenize = model, tokenize
        self.device, self.max_length = device, max_length
        self.layer, self.legacy, self.return_pooled = layer, legacy, always_return_pooled
        self.layer_idx = {"last": 0, "penultimate": 1}[layer]
        if freeze:
            self.freeze()

    def freeze(self):
        _freeze(self)

    def _states(self, tokens):
        """-> (penultimate, last), each ``[b, 77, width]``"""
        tower = self.model
        h = (tower.token_embedding(tokens) + tower.positional_embedding).transpose(0, 1)      # sequence-first for resblocks
        *body, head = list(tower.transformer.resblocks)
        for blk in body:
            h = blk(h, attn_mask=tower.attn_mask)
        return h.transpose(0, 1), head(h, attn_mask=tower.attn_mask).transpose(0, 1)

    @torch.no_grad()
    def forward(self, text):
        tokens = self.tokenize(text).to(self.device)
        penultimate, last = self._states(tokens)
        z = penultimate if self.layer == "penultimate" else last
        if self.legacy:
            if self.return_pooled:
                raise ValueError("always_return_pooled needs legacy=False")
            return self.model.ln_final(z)
        if not self.return_pooled:
            return z
        eot = tokens.argmax(dim=-1)
        rows = torch.arange(tokens.shape[0], device=tokens.device)
        return z, self.model.ln_final(last)[rows, eot] @ self.model.text_projection

    def encode(self, text):
        return self(text)        # through nn.Module.__call__ (hooks), as the reference does


class GeneralConditioner(nn.Module):
    OUTPUT_DIM2KEYS = {2: "vector", 3: "crossattn", 4: "concat", 5: "concat"}
    KEY2CATDIM = {"vector": 1, "crossattn": 2, "concat": 1, "control_vector": 1}

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
        if not getattr(self, "_warned_txt", False) and any(str(t).strip() for t in batch.get("txt", [])):
            import logging   # same behaviour as the reference class (it never reads batch['txt']) -- but say so once
            logging.getLogger("rsvld_amd").warning(
                "PreparedConditioner: the caption / a_prompt / n_prompt text is NOT used; conditioning comes from the cached "
                "embeddings only (build them for your prompts with tools/build_cached_cond.py, or use juggernautXL.yaml)")
            self._warned_txt = True
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
