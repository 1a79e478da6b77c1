"""LLaVA-NeXT caption pass on stock PyTorch-ROCm (BASELINE configs[3] "live LLaVA-Next prompt").

The reference runs a vendored LLaVA-NeXT tree (``llava/``, ~20 k lines) whose loader defaults to
``attn_implementation="flash_attention_2"`` (llava/model/builder.py:30) and wraps the model in a ``peft`` adapter
(models/util.py:111-117).  Per BASELINE.json north_star this pass stays a PyTorch module (it is conditioning, not the
hot path); what the build owns is a ROCm-clean way to load and drive it:

  * ``LlavaNextLlama``: transformers' own ``LlamaForCausalLM`` + ``CLIPVisionModel`` (SDPA attention -- no flash-attn, no
    bitsandbytes) with the multimodal members named as in the reference model (``model.vision_tower.vision_tower.*``,
    ``model.mm_projector.{0,2}``, ``model.image_newline``; llava/model/llava_arch.py:36-49), so the reference's
    ``lmms-lab/llama3-llava-next-8b`` checkpoint loads unchanged;
  * the "anyres" image path: tile selection, resize+pad, tiling (llava/mm_utils.py:121-296), CLIP layer -2 patch features
    -> 2-layer GELU MLP (multimodal_encoder/clip_encoder.py:48-82, multimodal_projector/builder.py:41-48), un-padding of
    the tile grid and one ``image_newline`` embedding per feature row (llava_arch.py:129-161,372-409), spliced into the
    token embeddings where the prompt holds IMAGE_TOKEN_INDEX (llava_arch.py:440-497);
  * ``get_img_describe`` with the reference's signature (models/util.py:17-66): Llama-3 chat template of
    ``conv_llava_llama_3`` (llava/conversation.py:98-110,387-398), sampling at temperature 0.2 -- plus an explicit
    ``seed`` so captions are reproducible;
  * ``merge_lora``: the adapter of ``./CKPT_PTH/Llava-next`` folded into the base weights (W += B A * alpha / r) when
    ``peft`` is not installed.

Pinned offline by tests/test_llava_next.py against token ids, input embeddings and logits the REFERENCE model produced on a
tiny seeded LLaMA + CLIP configuration (tests/golden/gen_llava_golden.py).
"""
import ast
import json
import math
import os
import re

import torch
from torch import nn

IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200            # llava/constants.py
DEFAULT_IMAGE_TOKEN = "<image>"
LLAMA3_SYSTEM = ("You are a helpful language and vision assistant. You are able to understand the visual content that the user "
                 "provides, and assist the user with a variety of tasks using natural language.")   # conversation.py:388
DEFAULT_MODEL = "lmms-lab/llama3-llava-next-8b"     # models/util.py:112-114
DEFAULT_ADAPTER = "./CKPT_PTH/Llava-next"           # models/util.py:115


# ------------------------------------------------------------------------------------------------ image side (host)
def _resolutions(grid_pinpoints, patch_size):
    """``image_grid_pinpoints`` as a list of (width, height): a list, its string form, or the "(1x1),...,(3x3)" range form."""
    if isinstance(grid_pinpoints, str) and "x" in grid_pinpoints:
        spans = [(int(a), int(b)) for a, b in re.findall(r"\((\d+)x(\d+)\)", grid_pinpoints)]
        (a0, b0), (a1, b1) = spans[0], spans[-1]
        return [[i * patch_size, j * patch_size] for i in range(a0, a1 + 1) for j in range(b0, b1 + 1)]
    return grid_pinpoints if isinstance(grid_pinpoints, list) else ast.literal_eval(grid_pinpoints)


def select_best_resolution(original_size, possible_resolutions):
    """The candidate (width, height) that keeps most of the image's pixels after an aspect-preserving fit, ties broken by
    the least wasted canvas (mm_utils.py:121-151)."""
    ow, oh = original_size
    best, best_key = None, None
    for w, h in possible_resolutions:
        s = min(w / ow, h / oh)
        eff = min(int(ow * s) * int(oh * s), ow * oh)
        key = (eff, -(w * h - eff))
        if best_key is None or key > best_key:
            best, best_key = (w, h), key
    return best


def resize_and_pad_image(image, target_resolution):
    """Aspect-preserving resize (PIL default filter, like the reference) centred on a black canvas (mm_utils.py:154-190)."""
    from PIL import Image
    ow, oh = image.size
    tw, th = target_resolution
    sw, sh = tw / ow, th / oh
    if sw < sh:
        nw, nh = tw, min(math.ceil(oh * sw), th)
    else:
        nw, nh = min(math.ceil(ow * sh), tw), th
    canvas = Image.new("RGB", (tw, th), (0, 0, 0))
    canvas.paste(image.resize((nw, nh)), ((tw - nw) // 2, (th - nh) // 2))
    return canvas


def _proc_sizes(processor):
    """(shortest_edge, crop) of a CLIP image processor across transformers versions (dict, SizeDict or int pair)."""
    def get(obj, key):
        try:
            return obj[key]
        except (KeyError, TypeError, IndexError):
            return getattr(obj, key, None)
    size, crop = processor.size, processor.crop_size
    edge = get(size, "shortest_edge")
    if edge is None:
        edge = min(v for v in (get(size, "height"), get(size, "width")) if v is not None) if not isinstance(size, (tuple, list)) \
            else min(size)
    return int(edge), int(get(crop, "height"))


def process_anyres_image(image, processor, grid_pinpoints):
    """-> ``[1 + tiles, 3, S, S]``: the whole image squashed to S x S (the reference resizes, it does not pad,
    mm_utils.py:281-293), then the row-major S x S tiles of the best-fitting padded canvas."""
    edge, crop = _proc_sizes(processor)
    canvas = resize_and_pad_image(image, select_best_resolution(image.size, _resolutions(grid_pinpoints, edge)))
    w, h = canvas.size
    views = [image.resize((edge, edge))]
    views += [canvas.crop((x, y, x + crop, y + crop)) for y in range(0, h, crop) for x in range(0, w, crop)]
    return torch.stack([processor.preprocess(v, return_tensors="pt")["pixel_values"][0] for v in views], dim=0)


def process_images(images, image_processor, model_cfg):
    """mm_utils.py:316-340 for the aspect modes the shipped LLaVA-NeXT checkpoints use ("anyres"; plain preprocess otherwise)."""
    mode = getattr(model_cfg, "image_aspect_ratio", None)
    if mode == "anyres" or (mode is not None and "anyres_max" in mode):
        out = [process_anyres_image(im, image_processor, model_cfg.image_grid_pinpoints) for im in images]
        return torch.stack(out, dim=0) if all(o.shape == out[0].shape for o in out) else out
    if mode in ("highres", "crop_split", "pad"):
        raise NotImplementedError(f"image_aspect_ratio={mode!r}: the LLaVA-NeXT checkpoints of the pipeline use 'anyres'")
    return image_processor.preprocess(images, return_tensors="pt")["pixel_values"]


def anyres_grid_shape(image_size, grid_pinpoints, patch_size):
    """(tiles across, tiles down) of the canvas chosen for ``image_size`` = (width, height) (mm_utils.py:215-242)."""
    w, h = select_best_resolution(image_size, _resolutions(grid_pinpoints, patch_size))
    return w // patch_size, h // patch_size


def unpad_image(feat, original_size):
    """Crop a ``[C, H, W]`` feature map of a padded canvas back to the image's aspect (llava_arch.py:129-161; note the
    truncating ``int`` and the symmetric crop, which can leave one padded row / column)."""
    ow, oh = original_size
    ch, cw = feat.shape[1:]
    if ow / oh > cw / ch:                       # padded above / below
        pad = (ch - int(oh * (cw / ow))) // 2
        return feat[:, pad:ch - pad, :]
    pad = (cw - int(ow * (ch / oh))) // 2       # padded left / right
    return feat[:, :, pad:cw - pad]


# ------------------------------------------------------------------------------------------------ text side (host)
def tokenizer_image_token(prompt, tokenizer, image_token_index=IMAGE_TOKEN_INDEX, return_tensors=None):
    """Tokenise around every "<image>" and put ``image_token_index`` in its place; a BOS the tokenizer prepends to each
    chunk is kept once (mm_utils.py:343-362)."""
    chunks = [tokenizer(c).input_ids for c in prompt.split(DEFAULT_IMAGE_TOKEN)]
    has_bos = bool(chunks and chunks[0] and chunks[0][0] == tokenizer.bos_token_id)
    ids = [chunks[0][0]] if has_bos else []
    skip = 1 if has_bos else 0
    for i, c in enumerate(chunks):
        if i > 0:
            ids.append(image_token_index)
        ids.extend(c[skip:])
    if return_tensors is None:
        return ids
    if return_tensors != "pt":
        raise ValueError(f"Unsupported tensor type: {return_tensors}")
    return torch.tensor(ids, dtype=torch.long)


def llama3_prompt(tokenizer, question, system=LLAMA3_SYSTEM):
    """The prompt ``conv_templates["llava_llama_3"]`` builds for one user turn (conversation.py:98-110): the tokenizer's own
    chat template over [system, user], with the generation header appended."""
    messages = [{"role": "system", "content": system}, {"role": "user", "content": question}]
    return tokenizer.apply_chat_template(messages, tokenize=False, add_generation_prompt=True)


class _Llama3Template:
    """Minimal stand-in for an entry of the reference's ``conv_templates`` dict, for callers that pass one through
    ``get_img_describe(conv_templates=...)``."""
    system, roles = LLAMA3_SYSTEM, ("user", "assistant")


conv_templates = {"llava_llama_3": _Llama3Template()}


# ------------------------------------------------------------------------------------------------ the model
def _llama_config_cls():
    from transformers import LlamaConfig

    class LlavaNextConfig(LlamaConfig):
        model_type = "llava_llama"          # what config.json of the reference checkpoints says (llava_llama.py:30-37)

    return LlavaNextConfig


class _VisionTower(nn.Module):
    """``model.vision_tower``: holds the CLIP vision model as ``.vision_tower`` like the reference's CLIPVisionTower, and
    returns the hidden states of ``select_layer`` without the class token (clip_encoder.py:48-82)."""

    def __init__(self, clip, select_layer, select_feature):
        super().__init__()
        self.vision_tower = clip
        self.select_layer, self.select_feature = select_layer, select_feature
        if select_feature not in ("patch", "cls_patch"):
            raise NotImplementedError(f"mm_vision_select_feature={select_feature!r}")

    @property
    def config(self):
        return self.vision_tower.config

    @property
    def image_size(self):
        return self.config.image_size

    @property
    def num_patches_per_side(self):
        return self.config.image_size // self.config.patch_size

    def forward(self, images):
        p = next(self.vision_tower.parameters())
        hs = self.vision_tower(images.to(device=p.device, dtype=p.dtype), output_hidden_states=True).hidden_states[self.select_layer]
        return (hs[:, 1:] if self.select_feature == "patch" else hs).to(images.dtype)


def build_model(config, clip=None):
    """``LlavaNextLlama(config)``.  ``config``: a LlamaConfig carrying the reference's multimodal fields (mm_vision_tower,
    mm_projector_type, mm_hidden_size, mm_vision_select_layer, image_aspect_ratio, image_grid_pinpoints,
    mm_patch_merge_type).  ``clip``: a ready CLIPVisionModel, else one is built from ``config.mm_vision_tower``'s config."""
    from transformers import CLIPVisionConfig, CLIPVisionModel, LlamaForCausalLM

    class LlavaNextLlama(LlamaForCausalLM):
        config_class = _llama_config_cls()

        def __init__(self, cfg):
            super().__init__(cfg)
            tower = clip if clip is not None else CLIPVisionModel(CLIPVisionConfig.from_pretrained(cfg.mm_vision_tower))
            tower.requires_grad_(False)
            self.model.vision_tower = _VisionTower(tower, cfg.mm_vision_select_layer, getattr(cfg, "mm_vision_select_feature", "patch"))
            m = re.match(r"^mlp(\d+)x_gelu$", getattr(cfg, "mm_projector_type", "linear"))
            if m is None and cfg.mm_projector_type != "linear":
                raise NotImplementedError(f"mm_projector_type={cfg.mm_projector_type!r}")
            layers = [nn.Linear(cfg.mm_hidden_size, cfg.hidden_size)]
            for _ in range(1, int(m.group(1)) if m else 1):
                layers += [nn.GELU(), nn.Linear(cfg.hidden_size, cfg.hidden_size)]
            self.model.mm_projector = nn.Sequential(*layers) if m else layers[0]
            if "unpad" in getattr(cfg, "mm_patch_merge_type", ""):
                self.model.image_newline = nn.Parameter(torch.zeros(cfg.hidden_size))

        # -------- image features -> one [tokens, hidden] block per image
        def encode_images(self, pixel_values):
            return self.model.mm_projector(self.model.vision_tower(pixel_values))

        def image_blocks(self, images, image_sizes):
            """``images``: list of ``[1 + tiles, 3, S, S]`` (or a 5-D stack).  Base view first, then the un-padded tile grid
            with a newline embedding closing every feature row (mm_patch_merge_type "spatial_unpad", llava_arch.py:339-409)."""
            cfg = self.config
            merge = getattr(cfg, "mm_patch_merge_type", "flat")
            views = [im if im.ndim == 4 else im.unsqueeze(0) for im in images]
            feats = torch.split(self.encode_images(torch.cat(views, dim=0)), [v.shape[0] for v in views])
            if merge == "flat":
                return [f.flatten(0, 1) for f in feats]
            if merge != "spatial_unpad":
                raise NotImplementedError(f"mm_patch_merge_type={merge!r}")
            side = self.model.vision_tower.num_patches_per_side
            nl = self.model.image_newline
            out = []
            for f, size in zip(feats, image_sizes):
                if f.shape[0] == 1:
                    out.append(torch.cat([f[0], nl[None].to(f.dtype)], dim=0))
                    continue
                nx, ny = anyres_grid_shape(size, cfg.image_grid_pinpoints, self.model.vision_tower.image_size)
                grid = f[1:].view(ny, nx, side, side, -1).permute(4, 0, 2, 1, 3).reshape(f.shape[-1], ny * side, nx * side)
                grid = unpad_image(grid, size)
                grid = torch.cat([grid, nl.to(grid.dtype)[:, None, None].expand(-1, grid.shape[1], 1)], dim=-1)
                out.append(torch.cat([f[0], grid.flatten(1, 2).transpose(0, 1)], dim=0))
            return out

        def multimodal_embeds(self, input_ids, images, image_sizes):
            """``input_ids [1, T]`` holding IMAGE_TOKEN_INDEX placeholders -> ``inputs_embeds [1, T', hidden]``
            (llava_arch.py:440-497, single sequence, no padding)."""
            if input_ids.shape[0] != 1:
                raise NotImplementedError("the caption pass runs one prompt at a time (models/util.py:39-43)")
            ids = input_ids[0]
            blocks = self.image_blocks(images, image_sizes)
            where = (ids == IMAGE_TOKEN_INDEX).nonzero().flatten().tolist()
            if len(where) != len(blocks):
                raise ValueError(f"{len(where)} image placeholders in the prompt for {len(blocks)} images")
            embed = self.model.embed_tokens
            parts, start = [], 0
            for pos, blk in zip(where, blocks):
                parts += [embed(ids[start:pos]), blk.to(embed.weight.dtype)]
                start = pos + 1
            parts.append(embed(ids[start:]))
            emb = torch.cat(parts, dim=0)
            limit = getattr(self.config, "tokenizer_model_max_length", None)
            return emb[:limit][None]

        @torch.no_grad()
        def generate(self, inputs=None, images=None, image_sizes=None, **kw):
            """``generate(input_ids, images=[...], image_sizes=[(w, h)], do_sample=..., ...)`` like the reference's override
            (llava_llama.py:118-137): the prompt is handed to the decoder as embeddings."""
            kw.pop("modalities", None)
            if images is None:
                return super().generate(inputs, **kw)
            return super().generate(inputs_embeds=self.multimodal_embeds(inputs, images, image_sizes), **kw)

    return LlavaNextLlama(config)


class FastDecoder:
    """Token loop of the caption pass over a static key/value cache, with the decode step replayed from a hipGraph.

    ``transformers``' ``generate`` launches ~500 kernels per token from Python (8 B Llama: 32 layers) and is host-bound at
    20-30 ms per token; a caption is 256 tokens.  This class runs the SAME module weights (``model.model.layers[i].self_attn
    .{q,k,v,o}_proj``, ``.mlp.{gate,up,down}_proj``, the two RMSNorms, ``model.model.rotary_emb``, ``lm_head``) through a
    functional forward written out in stock torch ops -- RMSNorm, rotary embedding (half-split form), grouped-query SDPA
    over a pre-allocated ``[1, kv_heads, max_len, head_dim]`` cache, SwiGLU -- so that one decode step has static shapes
    and addresses and can be captured once (``torch.cuda.CUDAGraph`` = hipGraph) and replayed per token.  Sampling follows
    ``generate`` exactly (temperature, then one ``torch.multinomial`` per step; greedy = argmax), which is what
    tests/test_llava_next.py pins token for token against the reference's generations.  On the CPU (tests) the same code
    runs eagerly."""

    def __init__(self, model, max_len):
        cfg = model.config
        self.model, self.max_len = model, int(max_len)
        self.layers = list(model.model.layers)
        self.n_q, self.n_kv = cfg.num_attention_heads, cfg.num_key_value_heads
        self.hd = getattr(cfg, "head_dim", None) or cfg.hidden_size // cfg.num_attention_heads
        p = next(model.parameters())
        self.dev, self.dt = p.device, p.dtype
        shape = (1, self.n_kv, self.max_len, self.hd)
        self.k = [torch.zeros(shape, device=self.dev, dtype=self.dt) for _ in self.layers]
        self.v = [torch.zeros(shape, device=self.dev, dtype=self.dt) for _ in self.layers]
        self.cols = torch.arange(self.max_len, device=self.dev)
        self._graph = None
        self._ws = None
        self.fused = True        # the decode step through the fused kernels where their conditions hold (``_fused_ok``); False = the torch-op sequence
        self.profile = None      # a dict: generate() fills in device-synchronised seconds of its prefill and its token loop
        # One weight-streaming GEMV for q | k | v and one for gate | up: the three (two) weight matrices of a layer become views
        # of one concatenated tensor (no copy is kept, ``state_dict`` is unchanged), so a decode step launches 4 GEMVs per layer
        # instead of 7.
        self.wqkv, self.bqkv, self.wgu = [], [], []
        for layer in self.layers:
            at, mlp = layer.self_attn, layer.mlp
            self.wqkv.append(self._fuse([at.q_proj, at.k_proj, at.v_proj], "weight"))
            self.bqkv.append(self._fuse([at.q_proj, at.k_proj, at.v_proj], "bias") if at.q_proj.bias is not None else None)
            if mlp.gate_proj.bias is not None:
                raise NotImplementedError("FastDecoder: biased MLP projections")
            self.wgu.append(self._fuse([mlp.gate_proj, mlp.up_proj], "weight"))

    def stale(self):
        """The fused q | k | v and gate | up tensors are VIEWED by the model's own modules; ``model.to()`` / ``.half()`` / a LoRA merge /
        ``load_state_dict(assign=True)`` give the modules new storage and leave this decoder (weights, KV cache, captured graph) on
        the old one.  Cheap check: first and last layer still alias the fused tensors, on the same device and dtype."""
        for i in (0, len(self.layers) - 1):
            q = self.layers[i].self_attn.q_proj.weight
            g = self.layers[i].mlp.gate_proj.weight
            if (q.data_ptr() != self.wqkv[i].data_ptr() or g.data_ptr() != self.wgu[i].data_ptr() or q.device != self.wqkv[i].device
                    or q.dtype != self.wqkv[i].dtype):
                return True
        return False

    @staticmethod
    def _fuse(mods, name):
        with torch.no_grad():
            parts = [getattr(m, name) for m in mods]
            fused = torch.cat([q.data for q in parts], dim=0)
            o = 0
            for q in parts:
                q.data = fused[o:o + q.shape[0]]      # the module keeps working (and saving) through a view of the fused tensor
                o += q.shape[0]
        return fused

    def _rms(self, x, norm):
        return torch.nn.functional.rms_norm(x, (x.shape[-1],), norm.weight, norm.variance_epsilon)

    def _lin(self, x, w, b=None):
        """``x [T, K] @ w[N, K]^T``.  The decode step (ONE row, 16-bit, on the GPU) streams the weights through this library's
        HBM-bound matrix-vector kernel (rsvld_gemv: the PyTorch GEMV of the same shapes ran at ~3 TB/s); everything else is
        ``torch.nn.functional.linear``."""
        if x.shape[0] == 1 and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and w.is_contiguous() and w.shape[1] % 8 == 0:
            from . import ops
            return ops.gemv(w, x[0].contiguous(), b)[None]
        return torch.nn.functional.linear(x, w, b)

    def forward(self, embeds, pos):
        """``embeds [1, T, H]`` at absolute positions ``pos [T]`` (long, on the device) -> logits of the LAST position
        ``[1, vocab]``; writes the keys / values of these positions into the cache.  One code path for the prefill (T > 1)
        and the decode step (T = 1): scores in the activation type, softmax in fp32 (``transformers``' eager attention),
        grouped-query heads as a batched matmul over the kv heads -- no expanded copy of the cache, no mask tensor beyond an
        additive ``[T, max_len]`` row."""
        m = self.model.model
        T = embeds.shape[1]
        nq, nkv, hd = self.n_q, self.n_kv, self.hd
        g = nq // nkv
        if T == 1 and self._fused_ok(embeds):
            return self._forward_fused(embeds, pos)
        cos, sin = m.rotary_emb(embeds, pos[None])                                 # [1, T, hd]
        cos, sin = cos[0][:, None], sin[0][:, None]                                # [T, 1, hd]
        bias = torch.zeros((T, self.max_len), device=self.dev, dtype=torch.float32)
        bias.masked_fill_(self.cols[None, :] > pos[:, None], float("-inf"))       # causal over the filled prefix
        scale = hd ** -0.5
        h = embeds[0]                                                               # [T, H]
        for i, layer in enumerate(self.layers):
            x = self._rms(h, layer.input_layernorm)
            qkv = self._lin(x, self.wqkv[i], self.bqkv[i]).view(T, nq + 2 * nkv, hd)
            qk = qkv[:, :nq + nkv]
            half = hd // 2
            qk = qk * cos + torch.cat((-qk[..., half:], qk[..., :half]), dim=-1) * sin     # rotary embedding on q and k at once
            self.k[i][0].index_copy_(1, pos, qk[:, nq:].transpose(0, 1))
            self.v[i][0].index_copy_(1, pos, qkv[:, nq + nkv:].transpose(0, 1))
            q = qk[:, :nq].reshape(T, nkv, g, hd).permute(1, 0, 2, 3).reshape(nkv, T * g, hd)  # [kv head, (t, q-in-group), hd]
            sc = torch.matmul(q, self.k[i][0].transpose(1, 2)).float().view(nkv, T, g, self.max_len)
            pr = torch.softmax(sc * scale + bias[None, :, None, :], dim=-1).to(self.dt).view(nkv, T * g, self.max_len)
            o = torch.matmul(pr, self.v[i][0]).view(nkv, T, g, hd).permute(1, 0, 2, 3).reshape(T, nq * hd)
            h = h + self._lin(o, layer.self_attn.o_proj.weight, layer.self_attn.o_proj.bias)
            x = self._rms(h, layer.post_attention_layernorm)
            gu = self._lin(x, self.wgu[i])
            inter = gu.shape[-1] // 2
            h = h + self._lin(torch.nn.functional.silu(gu[:, :inter]) * gu[:, inter:], layer.mlp.down_proj.weight)
        return self._lin(self._rms(h[-1:], m.norm), self.model.lm_head.weight, self.model.lm_head.bias)

    def _fused_ok(self, embeds):
        """The decode step as the library's fused kernels (rsvld_gemv_fused, rsvld_llama_decode_attention): 16-bit tensors on the GPU, head_dim
        128, at most eight query heads per kv head, unbiased MLP, 16-byte aligned rows."""
        return (self.fused and embeds.is_cuda and embeds.dtype in (torch.float16, torch.bfloat16) and self.hd == 128
                and self.n_q % self.n_kv == 0 and self.n_q // self.n_kv <= 8 and embeds.shape[-1] % 8 == 0)

    def _forward_fused(self, embeds, pos):
        """ONE new token through the decoder in 7 launches per layer instead of ~26: RMSNorm folded into the q|k|v and gate|up products, rotary
        embedding + cache write + grouped-query attention in one operator, the residual additions in the o_proj / down_proj epilogues, SwiGLU in
        down_proj's prologue.  Same arithmetic and the same roundings as ``forward`` up to the order of two fp32 sums (the RMSNorm's mean, the
        softmax over key chunks) and the 16-bit rounding of P, which this path skips."""
        from . import ops
        m = self.model.model
        cos, sin = m.rotary_emb(embeds, pos[None])                                 # [1, 1, hd]
        cos, sin = cos.reshape(-1).contiguous(), sin.reshape(-1).contiguous()
        h = embeds.reshape(-1).contiguous()
        scale = self.hd ** -0.5
        if self._ws is None:
            lib = ops.L.load()
            self._ws = torch.empty(int(lib.rsvld_llama_decode_attention_ws_bytes(self.n_q, self.n_kv, self.max_len)) // 4, device=self.dev,
                                   dtype=torch.float32)          # (per-chunk partial results: written before they are read, no initial state)
        for i, layer in enumerate(self.layers):
            n1, n2 = layer.input_layernorm, layer.post_attention_layernorm
            qkv = ops.gemv_fused(self.wqkv[i], h, self.bqkv[i], norm=(n1.weight, n1.variance_epsilon))
            o = ops.llama_decode_attention(qkv, cos, sin, pos, self.k[i][0], self.v[i][0], self.n_q, self.n_kv, scale, ws=self._ws)
            op = layer.self_attn.o_proj
            h = ops.gemv_fused(op.weight, o, op.bias, residual=h)
            gu = ops.gemv_fused(self.wgu[i], h, None, norm=(n2.weight, n2.variance_epsilon))
            h = ops.gemv_fused(layer.mlp.down_proj.weight, gu, None, glu=True, residual=h)
        head = self.model.lm_head
        return ops.gemv_fused(head.weight, h, head.bias, norm=(m.norm.weight, m.norm.variance_epsilon))[None]

    def _pick(self, logits, do_sample, temperature, top_k=0, top_p=1.0):
        """``generate``'s sampling chain (transformers' TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper -> multinomial).
        The reference calls ``generate(do_sample=True, temperature=0.2)`` (models/util.py:50-60) and thereby inherits the generation
        config's ``top_k`` (transformers' default: 50): sampling the full softmax instead would be a different sampler."""
        if not do_sample:
            return logits.argmax(-1)
        scores = logits.float() / temperature
        if top_k and top_k > 0:
            kth = torch.topk(scores, min(int(top_k), scores.shape[-1]))[0][..., -1, None]
            scores = scores.masked_fill(scores < kth, float("-inf"))
        if top_p is not None and top_p < 1.0:
            srt, idx = torch.sort(scores, descending=False)
            remove = srt.softmax(dim=-1).cumsum(dim=-1) <= (1.0 - top_p)
            remove[..., -1:] = False                                   # min_tokens_to_keep = 1
            scores = scores.masked_fill(remove.scatter(-1, idx, remove), float("-inf"))
        return torch.multinomial(torch.softmax(scores, dim=-1), 1)[:, 0]

    def _sampling_defaults(self, do_sample, top_k, top_p):
        """``None`` = what ``generate`` would take from the model's generation_config for a sampling call (top_k 50, top_p 1.0 unless
        the checkpoint says otherwise)."""
        g = getattr(self.model, "generation_config", None)
        if top_k is None:
            top_k = getattr(g, "top_k", 50) if do_sample else 0
        if top_p is None:
            top_p = getattr(g, "top_p", 1.0) if do_sample else 1.0
        return (top_k or 0), (1.0 if top_p is None else top_p)

    @torch.no_grad()
    def generate(self, inputs_embeds, max_new_tokens, do_sample=False, temperature=1.0, eos_token_id=None, use_graph=None,
                 top_k=None, top_p=None):
        """-> ``[n]`` generated token ids (n <= max_new_tokens; stops after an ``eos_token_id``, which is included).
        ``top_k`` / ``top_p``: None = the generation config's values, as ``generate`` resolves them."""
        top_k, top_p = self._sampling_defaults(do_sample, top_k, top_p)
        T0 = inputs_embeds.shape[1]
        if T0 + max_new_tokens > self.max_len:
            raise ValueError(f"FastDecoder: prompt {T0} + {max_new_tokens} new tokens exceed the cache ({self.max_len})")
        use_graph = (self.dev.type == "cuda") if use_graph is None else use_graph
        eos = set([eos_token_id] if isinstance(eos_token_id, int) else (eos_token_id or []))
        embed = self.model.model.embed_tokens
        if self.profile is not None:
            import time
            torch.cuda.current_stream().synchronize()   # (this stream only: a caller may run other work on a second stream beside the token loop)
            t_start = time.perf_counter()
        logits = self.forward(inputs_embeds.to(self.dt), torch.arange(T0, device=self.dev))
        tok = self._pick(logits, do_sample, temperature, top_k, top_p)
        out = [tok]
        if self.profile is not None:
            torch.cuda.current_stream().synchronize()   # (this stream only: a caller may run other work on a second stream beside the token loop)
            t_prefill = time.perf_counter()
        if use_graph and self._graph is None:
            self._tok, self._pos = tok.clone(), torch.tensor([T0], device=self.dev)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                              # warm-up off the capture stream (allocator, lazy init);
                self.forward(embed(self._tok)[None], self._pos)        # it rewrites cache slot T0 with the same values
            torch.cuda.current_stream().wait_stream(side)
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._logits = self.forward(embed(self._tok)[None], self._pos)
        for n in range(1, max_new_tokens):
            if eos and int(tok) in eos:
                break
            if use_graph:
                self._tok.copy_(tok.view(-1))
                self._pos.fill_(T0 + n - 1)
                self._graph.replay()
                logits = self._logits
            else:
                logits = self.forward(embed(tok)[None], torch.tensor([T0 + n - 1], device=self.dev))
            tok = self._pick(logits, do_sample, temperature, top_k, top_p)
            out.append(tok)
        if self.profile is not None:
            torch.cuda.current_stream().synchronize()   # (this stream only: a caller may run other work on a second stream beside the token loop)
            self.profile.update(prompt_tokens=T0, prefill_s=t_prefill - t_start, decode_s=time.perf_counter() - t_prefill,
                                new_tokens=len(out))
        return torch.cat(out)


def normalise_checkpoint_keys(state_dict, model):
    """Checkpoints written by transformers 4.x name the CLIP weights ``...vision_tower.vision_tower.vision_model.*``;
    transformers 5 modules drop the ``vision_model.`` level (and vice versa).  Rename towards what ``model`` has."""
    have = set(model.state_dict().keys())
    mid = ".vision_tower.vision_tower."
    out = {}
    for k, v in state_dict.items():
        if k not in have and mid in k:
            alt = k.replace(mid + "vision_model.", mid) if mid + "vision_model." in k else k.replace(mid, mid + "vision_model.")
            if alt in have:
                k = alt
        out[k] = v
    return out


def merge_lora(model, adapter_dir):
    """Fold a PEFT LoRA adapter (adapter_config.json + adapter_model.safetensors / .bin) into ``model``'s Linear weights:
    ``W += (B @ A) * lora_alpha / r``.  Equivalent to ``PeftModel.from_pretrained(...).merge_and_unload()`` for plain LoRA on
    Linear layers; anything else in the adapter raises."""
    cfg = json.load(open(os.path.join(adapter_dir, "adapter_config.json")))
    if cfg.get("peft_type", "LORA") != "LORA" or cfg.get("use_dora") or cfg.get("modules_to_save"):
        raise NotImplementedError("merge_lora handles plain LoRA adapters on Linear layers")
    scale = cfg["lora_alpha"] / cfg["r"]
    st = os.path.join(adapter_dir, "adapter_model.safetensors")
    if os.path.exists(st):
        import safetensors.torch
        sd = safetensors.torch.load_file(st)
    else:
        sd = torch.load(os.path.join(adapter_dir, "adapter_model.bin"), map_location="cpu")
    mods = dict(model.named_modules())
    merged = 0
    for ka, a in sd.items():
        if ".lora_A" not in ka:
            if ".lora_B" not in ka:
                raise NotImplementedError(f"unexpected adapter tensor {ka}")
            continue
        name = re.sub(r"^base_model\.model\.", "", ka.split(".lora_A")[0])
        b = sd[ka.replace(".lora_A", ".lora_B")]
        lin = mods[name]
        lin.weight.data += (b.to(torch.float32) @ a.to(torch.float32)).to(lin.weight) * scale
        merged += 1
    return merged


def resolve_model_dir(model_path):
    """A local directory as is; a hub id (``lmms-lab/llama3-llava-next-8b``, models/util.py:112-114) through the local
    Hugging Face cache only -- this build never fetches from the network."""
    if os.path.isdir(model_path):
        return model_path
    try:
        from huggingface_hub import snapshot_download
        return snapshot_download(model_path, local_files_only=True)
    except Exception as exc:   # not cached (LocalEntryNotFoundError), malformed id, hub library absent
        raise FileNotFoundError(
            f"load_llava: {model_path!r} is neither a directory nor a snapshot in the local Hugging Face cache "
            f"({type(exc).__name__}); download the checkpoint first, or skip the captioner with --no_llava / --caption") from exc


def load_llava(device="cuda", model_path=DEFAULT_MODEL, adapter_path=DEFAULT_ADAPTER, dtype=torch.float16):
    """-> (tokenizer, model, image_processor), the tuple of models/util.py:111-117.  SDPA attention (flash-attn does not
    exist on this platform and is not needed); the adapter is applied by ``peft`` when installed, else merged here.
    The skeleton is built on the meta device (no 32 GB fp32 random init of an 8 B model on the host) and EVERY parameter
    and buffer must then come from the checkpoint: a tensor no shard supplies raises instead of staying random."""
    import glob

    import safetensors.torch
    from transformers import AutoTokenizer, CLIPImageProcessor
    model_dir = resolve_model_dir(model_path)
    cfg_cls = _llama_config_cls()
    config = cfg_cls.from_pretrained(model_dir)
    config._attn_implementation = "sdpa"
    tokenizer = AutoTokenizer.from_pretrained(model_dir, use_fast=False)
    shards = sorted(glob.glob(os.path.join(model_dir, "*.safetensors")))
    if not shards:
        raise FileNotFoundError(f"load_llava: no *.safetensors under {model_dir!r} (download the checkpoint there first; "
                                f"this build does not fetch from the network)")
    with torch.device("meta"):
        model = build_model(config)
    wanted = set(model.state_dict().keys())
    loaded = set()
    for sh in shards:
        sd = normalise_checkpoint_keys(safetensors.torch.load_file(sh), model)
        extra = [k for k in sd if k not in wanted]
        if extra:
            raise RuntimeError(f"load_llava: unexpected keys in {sh}: {extra[:5]}")
        model.load_state_dict({k: v.to(dtype) if v.is_floating_point() else v for k, v in sd.items()}, strict=False, assign=True)
        loaded.update(sd.keys())
    # buffers that are derived, not stored (rotary inv_freq, CLIP position_ids) are re-created below; everything else
    # the checkpoint must hold
    missing = sorted(k for k in wanted - loaded if not k.endswith(("inv_freq", "position_ids")))
    if missing:
        raise RuntimeError(f"load_llava: {len(missing)} tensors of the model are in no shard of {model_dir!r} and would stay "
                           f"uninitialised, e.g. {missing[:5]} (vision tower / mm_projector / image_newline must be in the "
                           f"checkpoint, as in lmms-lab/llama3-llava-next-8b)")
    _materialise_derived_buffers(model)
    if adapter_path is not None:
        if not os.path.isdir(adapter_path):
            raise FileNotFoundError(f"load_llava: adapter directory {adapter_path!r} does not exist")
        try:
            from peft import PeftModel
            model = PeftModel.from_pretrained(model, adapter_path, device_map="cpu").merge_and_unload()
        except ImportError:
            merge_lora(model, adapter_path)
    model.eval().to(device=device, dtype=dtype)
    image_processor = CLIPImageProcessor.from_pretrained(config.mm_vision_tower)
    return tokenizer, model, image_processor


def _materialise_derived_buffers(model, force=False):
    """After a meta-device construction: re-create the buffers a checkpoint does not store (non-persistent ones) on the CPU.
    ``force``: re-create them even when they are not on the meta device (after ``to_empty`` they hold garbage)."""
    for mod in model.modules():
        for name, buf in list(mod._buffers.items()):
            if buf is None or not (buf.is_meta or (force and name in ("position_ids", "inv_freq", "original_inv_freq"))):
                continue
            if name == "position_ids":
                mod._buffers[name] = torch.arange(buf.shape[-1]).expand(buf.shape).clone()
            elif name in ("inv_freq", "original_inv_freq"):
                fn = getattr(mod, "rope_init_fn", None)
                if fn is not None:
                    inv, _ = fn(mod.config, "cpu")
                else:
                    base, dim = getattr(mod, "base", 10000.0), buf.shape[0] * 2
                    inv = 1.0 / (base ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))
                mod._buffers[name] = inv.to(torch.float32)
            else:
                raise RuntimeError(f"load_llava: buffer {name!r} of {type(mod).__name__} is neither in the checkpoint nor derivable")


def _eos_ids(model, tokenizer):
    ids = getattr(getattr(model, "generation_config", None), "eos_token_id", None)
    if ids is None:
        ids = getattr(model.config, "eos_token_id", None)
    if ids is None:
        ids = getattr(tokenizer, "eos_token_id", None)
    return [ids] if isinstance(ids, int) else list(ids or [])


def get_img_describe(image_tensor, image, model, tokenizer, prompt, conv_templates=conv_templates,
                     image_token_index=IMAGE_TOKEN_INDEX, conv_template="llava_llama_3", num_beams=1, temperature=0.2,
                     do_sample=True, max_new_tokens=512, device="cuda", seed=None, fast=None):
    """models/util.py:17-66 -> ``[caption]``.  ``seed`` (an addition) makes the caption a function of (image, prompt,
    weights, seed): sampling then runs inside ``torch.random.fork_rng`` with the CPU and the model's device generator seeded
    THERE, so the caller's generators -- which Stage 2 draws its noise from right afterwards, and in the reference live on
    another device (infer.py:145-166: LLaVA on cuda:1) -- are exactly where they were, however many tokens were sampled.
    ``seed=None`` samples from the current generator state without touching any seed.
    ``fast`` (default: on a GPU, with one beam): the token loop runs through ``FastDecoder`` (static cache, decode step
    replayed from a hipGraph) instead of ``transformers``' ``generate``; same tokens (tests/test_llava_next.py)."""
    if conv_template != "llava_llama_3":
        raise NotImplementedError("the pipeline's captioner is the Llama-3 LLaVA-NeXT (conv_template 'llava_llama_3')")
    system = getattr(conv_templates[conv_template], "system", LLAMA3_SYSTEM)
    text = llama3_prompt(tokenizer, prompt, system)
    input_ids = tokenizer_image_token(text, tokenizer, image_token_index, return_tensors="pt").unsqueeze(0).to(device)
    mdev = next(model.parameters()).device
    fast = (mdev.type == "cuda" and num_beams == 1) if fast is None else fast
    if fast and _has_logits_warpers(model):
        # FastDecoder implements generate()'s temperature / top_k / top_p chain; a generation_config with typical_p, min_p, penalties or
        # banned tokens goes through transformers' generate() (~4 x slower per token), and says so
        import warnings
        warnings.warn("llava_next: the checkpoint's generation_config asks for logits processors FastDecoder does not implement; "
                      "captioning through transformers' generate()")
        fast = False

    def run():
        if fast:   # (no_grad, not inference_mode: tensors made in inference mode cannot be updated in place by a later call
            with torch.no_grad():   # outside it, and a hipGraph capture that fails on that leaves the generator in capture state)
                return caption_tokens_fast(model, input_ids, image_tensor, [image.size], max_new_tokens, do_sample, temperature,
                                           _eos_ids(model, tokenizer))
        with torch.inference_mode():
            return model.generate(input_ids, images=image_tensor, image_sizes=[image.size], do_sample=do_sample,
                                  temperature=temperature, num_beams=num_beams, max_new_tokens=max_new_tokens,
                                  return_dict_in_generate=True, output_scores=True)[0][0]

    if seed is None:
        out = run()
    else:
        with torch.random.fork_rng(devices=[mdev] if mdev.type == "cuda" else []):
            torch.manual_seed(seed)
            out = run()
    return [tokenizer.decode(out.cpu().tolist(), skip_special_tokens=True).lstrip()]


def _has_logits_warpers(model):
    """True when the checkpoint's generation_config asks for logits processing FastDecoder does not implement.  It implements what the
    reference's call uses: temperature, top_k (the config's value; transformers' default 50 IS applied by a sampling generate())
    and top_p; typical_p / min_p / penalties / banned or suppressed tokens are not."""
    g = getattr(model, "generation_config", None)
    if g is None:
        return False
    def differs(name, neutral):
        v = getattr(g, name, None)
        return v is not None and v != neutral
    return (differs("typical_p", 1.0) or differs("repetition_penalty", 1.0) or differs("no_repeat_ngram_size", 0)
            or differs("min_p", None) or differs("encoder_repetition_penalty", 1.0) or bool(getattr(g, "bad_words_ids", None))
            or bool(getattr(g, "suppress_tokens", None)))


def caption_tokens_fast(model, input_ids, images, image_sizes, max_new_tokens, do_sample, temperature, eos_ids=None):
    """Multimodal prompt -> generated token ids through the model's ``FastDecoder`` (built once per model and cache size)."""
    embeds = model.multimodal_embeds(input_ids, images, image_sizes)
    need = embeds.shape[1] + max_new_tokens
    dec = getattr(model, "_fast_decoder", None)
    if dec is not None and dec.stale():     # the model was moved / cast / reloaded since the decoder fused its weights: rebuild
        dec = None
    if dec is None or dec.max_len < need:
        dec = FastDecoder(model, -(-need // 256) * 256)
        model.__dict__["_fast_decoder"] = dec          # not a submodule: plain attribute
    return dec.generate(embeds, max_new_tokens, do_sample=do_sample, temperature=temperature, eos_token_id=eos_ids)
