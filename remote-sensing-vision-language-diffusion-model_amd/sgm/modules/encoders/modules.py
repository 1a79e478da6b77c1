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


class FrozenCLIPEmbedder(AbstractEmbModel):
    """encoders/modules.py:436-499: HuggingFace CLIP text transformer; ``layer`` in {last, pooled, hidden}.

    ``version`` (or ``CKPT_PTH.SDXL_CLIP1_PATH`` when that module exists, as in the reference) names the checkpoint
    for ``from_pretrained``; ``tokenizer`` / ``transformer`` may be passed in instead."""

    LAYERS = ["last", "pooled", "hidden"]

    def __init__(self, version="openai/clip-vit-large-patch14", device="cuda", max_length=77, freeze=True, layer="last",
                 layer_idx=None, always_return_pooled=False, tokenizer=None, transformer=None):
        super().__init__()
        if layer not in self.LAYERS:
            raise ValueError(f"layer must be one of {self.LAYERS}")
        if tokenizer is None or transformer is None:
            from transformers import CLIPTextModel, CLIPTokenizer
            try:
                from CKPT_PTH import SDXL_CLIP1_PATH
            except ImportError:
                SDXL_CLIP1_PATH = None
            src = version if SDXL_CLIP1_PATH is None else SDXL_CLIP1_PATH
            tokenizer = CLIPTokenizer.from_pretrained(src) if tokenizer is None else tokenizer
            transformer = CLIPTextModel.from_pretrained(src) if transformer is None else transformer
        self.tokenizer = tokenizer
        self.transformer = transformer
        self.device = device
        self.max_length = max_length
        if freeze:
            self.freeze()
        self.layer = layer
        self.layer_idx = layer_idx
        self.return_pooled = always_return_pooled
        if layer == "hidden":
            if layer_idx is None or not 0 <= abs(layer_idx) <= 12:
                raise ValueError("layer='hidden' needs 0 <= |layer_idx| <= 12")

    def freeze(self):
        self.transformer = self.transformer.eval()
        for param in self.parameters():
            param.requires_grad = False

    @torch.no_grad()
    def forward(self, text):
        enc = self.tokenizer(text, truncation=True, max_length=self.max_length, return_length=True,
                             return_overflowing_tokens=False, padding="max_length", return_tensors="pt")
        tokens = enc["input_ids"].to(self.device)
        outputs = self.transformer(input_ids=tokens, output_hidden_states=self.layer == "hidden")
        if self.layer == "last":
            z = outputs.last_hidden_state
        elif self.layer == "pooled":
            z = outputs.pooler_output[:, None, :]
        else:
            z = outputs.hidden_states[self.layer_idx]
        if self.return_pooled:
            return z, outputs.pooler_output
        return z

    def encode(self, text):
        return self(text)


class FrozenOpenCLIPEmbedder2(AbstractEmbModel):
    """encoders/modules.py:502-612: OpenCLIP text transformer; ``layer`` in {last, penultimate}; with ``legacy=False``
    and ``always_return_pooled=True`` (the yaml's setting) returns (hidden states of that layer WITHOUT ln_final --
    only ``last`` passes through it for the pooling, :571-576 -- , pooled = ln_final(last)[eot] @ text_projection).

    ``model`` must expose open_clip's text-tower attributes: ``token_embedding``, ``positional_embedding``,
    ``transformer.resblocks`` (called as ``r(x, attn_mask=...)`` on LND tensors), ``ln_final``, ``text_projection``,
    ``attn_mask``.  Without ``model`` / ``tokenize`` the ``open_clip`` package builds them (the reference pins the
    ``laion2b_s39b_b160k`` weights whatever ``version`` says, :524-528)."""

    LAYERS = ["pooled", "last", "penultimate"]

    def __init__(self, arch="ViT-H-14", version="laion2b_s32b_b79k", device="cuda", max_length=77, freeze=True, layer="last",
                 always_return_pooled=False, legacy=True, model=None, tokenize=None):
        super().__init__()
        if layer not in self.LAYERS:
            raise ValueError(f"layer must be one of {self.LAYERS}")
        if model is None or tokenize is None:
            try:
                import open_clip
            except ImportError as e:
                raise ImportError("FrozenOpenCLIPEmbedder2 needs the open_clip package (or model= / tokenize= passed in); "
                                  "use PreparedConditioner with cached embeddings otherwise") from e
            if model is None:
                model, _, _ = open_clip.create_model_and_transforms(arch, device=torch.device("cpu"), pretrained="laion2b_s39b_b160k")
                del model.visual
            tokenize = open_clip.tokenize if tokenize is None else tokenize
        self.model = model
        self.tokenize = tokenize
        self.device = device
        self.max_length = max_length
        self.return_pooled = always_return_pooled
        if freeze:
            self.freeze()
        self.layer = layer
        if layer == "last":
            self.layer_idx = 0
        elif layer == "penultimate":
            self.layer_idx = 1
        else:
            raise NotImplementedError()
        self.legacy = legacy

    def freeze(self):
        self.model = self.model.eval()
        for param in self.parameters():
            param.requires_grad = False

    @torch.no_grad()
    def forward(self, text):
        tokens = self.tokenize(text)
        z = self.encode_with_transformer(tokens.to(self.device))
        if not self.return_pooled and self.legacy:
            return z
        if self.return_pooled:
            if self.legacy:
                raise ValueError("always_return_pooled needs legacy=False")
            return z[self.layer], z["pooled"]
        return z[self.layer]

    def encode_with_transformer(self, text):
        x = self.model.token_embedding(text)          # [b, n_ctx, width]
        x = x + self.model.positional_embedding
        x = x.permute(1, 0, 2)                        # NLD -> LND
        x = self.text_transformer_forward(x, attn_mask=self.model.attn_mask)
        if self.legacy:
            return self.model.ln_final(x[self.layer])
        o = self.model.ln_final(x["last"])
        x["pooled"] = self.pool(o, text)
        return x

    def pool(self, x, text):   # features at the eot token (the highest id of each sequence)
        return x[torch.arange(x.shape[0]), text.argmax(dim=-1)] @ self.model.text_projection

    def text_transformer_forward(self, x, attn_mask=None):
        outputs = {}
        blocks = self.model.transformer.resblocks
        for i, r in enumerate(blocks):
            if i == len(blocks) - 1:
                outputs["penultimate"] = x.permute(1, 0, 2)   # LND -> NLD
            x = r(x, attn_mask=attn_mask)
        outputs["last"] = x.permute(1, 0, 2)
        return outputs

    def encode(self, text):
        return self(text)


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
