"""Full-size, seeded random-weight stand-ins for the checkpoints bench.py cannot download (no network): the LLaVA-NeXT
captioner (Llama-3-8B decoder + CLIP ViT-L/14-336 tower, the architecture of lmms-lab/llama3-llava-next-8b named by
models/util.py:112-114) and the two text towers of the live conditioner (CLIP ViT-L/14 text, OpenCLIP ViT-bigG/14 text;
model_configs/juggernautXL.yaml:67-105).  Only shapes, layer counts and dtypes matter for timing; tokenizers are replaced
by deterministic hash tokenizers (the vocabulary files are not available offline either).  Nothing here is product code:
the product loads the real checkpoints through rsvld_amd.llava_next.load_llava / the embedders' constructors."""
import zlib

import torch
from torch import nn

LLAMA3_8B = dict(vocab_size=128256, hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32,
                 num_key_value_heads=8, max_position_embeddings=8192, rms_norm_eps=1e-5, rope_theta=500000.0,
                 bos_token_id=128000, eos_token_id=128009, pad_token_id=128001, tie_word_embeddings=False)
CLIP_L_336_VISION = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=336,
                         patch_size=14, projection_dim=768)
LLAVA_MM = dict(mm_projector_type="mlp2x_gelu", mm_hidden_size=1024, mm_vision_select_layer=-2, mm_vision_select_feature="patch",
                image_aspect_ratio="anyres", mm_patch_merge_type="spatial_unpad", mm_use_im_start_end=False,
                mm_use_im_patch_token=False, tokenizer_padding_side="right", tokenizer_model_max_length=8192,
                image_grid_pinpoints=[[336, 672], [672, 336], [672, 672], [1008, 336], [336, 1008]],
                mm_vision_tower="openai/clip-vit-large-patch14-336")
PROMPT_TEXT_TOKENS = 120     # Llama-3 chat template + system prompt + the shipped img_prompt (prompts/prompt_config.yaml)


def _seed_on_device(module, device, dtype, seed):
    """Materialise a meta-device module on ``device``: matrices ~ N(0, 0.02), vectors 0, norm gains 1 (device generator)."""
    module.to_empty(device=device)
    g = torch.Generator(device=device).manual_seed(seed)
    with torch.no_grad():
        for name, p_ in module.named_parameters():
            if p_.dim() >= 2:
                p_.normal_(0.0, 0.02, generator=g)
            elif "norm" in name.lower() or name.endswith("ln_1.weight") or name.endswith("ln_2.weight") or "ln_final.weight" in name \
                    or "layer_norm" in name or "layrnorm" in name:
                p_.fill_(0.0 if name.endswith("bias") else 1.0)
            else:
                p_.zero_()
    module.to(dtype)
    return module


def llava_full(device, seed=5, dtype=torch.float16):
    """-> (model, image_processor, prompt_ids): the 8 B LLaVA-NeXT architecture on ``device`` with seeded random weights,
    a CLIP image processor with the ViT-L/14-336 settings, and a synthetic prompt of PROMPT_TEXT_TOKENS ids around one
    image placeholder."""
    from transformers import CLIPImageProcessor, CLIPVisionConfig, CLIPVisionModel
    from rsvld_amd import llava_next as LN
    cfg = LN._llama_config_cls()(**LLAMA3_8B, **LLAVA_MM)
    cfg._attn_implementation = "sdpa"
    with torch.device("meta"):
        model = LN.build_model(cfg, clip=CLIPVisionModel(CLIPVisionConfig(**CLIP_L_336_VISION)))
    _seed_on_device(model, device, dtype, seed)
    LN._materialise_derived_buffers(model, force=True)     # to_empty left inv_freq / position_ids uninitialised
    model.to(device).eval()
    proc = CLIPImageProcessor(do_resize=True, size={"shortest_edge": 336}, do_center_crop=True, crop_size={"height": 336, "width": 336},
                              do_rescale=True, rescale_factor=1 / 255, do_normalize=True, do_convert_rgb=True, resample=3,
                              image_mean=[0.48145466, 0.4578275, 0.40821073], image_std=[0.26862954, 0.26130258, 0.27577711])
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1000, 100000, (PROMPT_TEXT_TOKENS,), generator=g)
    ids[0] = LLAMA3_8B["bos_token_id"]
    ids[60] = LN.IMAGE_TOKEN_INDEX
    return model, proc, ids[None]


class HashTokenizer:
    """77-id rows derived from the text's bytes (CRC-seeded), in both call conventions the embedders use: the HuggingFace
    one (``tok(text, padding=..., return_tensors="pt")["input_ids"]``) and open_clip's (``tokenize(text) -> LongTensor``)."""

    def __init__(self, vocab=49408, ctx=77):
        self.vocab, self.ctx = vocab, ctx

    def ids(self, text):
        rows = []
        for t in ([text] if isinstance(text, str) else list(text)):
            n = min(self.ctx - 2, max(1, len(t.split())))
            g = torch.Generator().manual_seed(zlib.crc32(t.encode()) % (2 ** 31))
            row = torch.zeros(self.ctx, dtype=torch.long)
            row[0] = self.vocab - 2                                     # bos
            row[1:1 + n] = torch.randint(1, self.vocab - 2, (n,), generator=g)
            row[1 + n] = self.vocab - 1                                 # eot = the highest id (what the pooling looks for)
            rows.append(row)
        return torch.stack(rows)

    def __call__(self, text, **kw):
        ids = self.ids(text)
        return {"input_ids": ids} if kw else ids


class _ResBlock(nn.Module):
    def __init__(self, width, heads):
        super().__init__()
        self.ln_1 = nn.LayerNorm(width)
        self.attn = nn.MultiheadAttention(width, heads)
        self.ln_2 = nn.LayerNorm(width)
        self.mlp = nn.Sequential()
        self.mlp.c_fc, self.mlp.gelu, self.mlp.c_proj = nn.Linear(width, 4 * width), nn.GELU(), nn.Linear(4 * width, width)

    def forward(self, x, attn_mask=None):
        h = self.ln_1(x)
        x = x + self.attn(h, h, h, need_weights=False, attn_mask=attn_mask)[0]
        return x + self.mlp(self.ln_2(x))


class OpenClipTextTower(nn.Module):
    """The attribute layout FrozenOpenCLIPEmbedder2 reads (open_clip's text tower), ViT-bigG/14 text sizes by default."""

    def __init__(self, vocab=49408, width=1280, heads=20, layers=32, proj=1280, ctx=77):
        super().__init__()
        self.token_embedding = nn.Embedding(vocab, width)
        self.positional_embedding = nn.Parameter(torch.empty(ctx, width))
        self.transformer = nn.Module()
        self.transformer.resblocks = nn.ModuleList([_ResBlock(width, heads) for _ in range(layers)])
        self.ln_final = nn.LayerNorm(width)
        self.text_projection = nn.Parameter(torch.empty(width, proj))
        self.register_buffer("attn_mask", torch.full((ctx, ctx), float("-inf")).triu_(1), persistent=False)


def live_conditioner_config(device, seed=6, dtype=torch.float32):
    """``conditioner_config`` of model_configs/juggernautXL.yaml:67-105 (GeneralConditionerWithControl over CLIP-L hidden
    layer 11, OpenCLIP bigG penultimate + pooled, three size embedders) with full-size seeded towers injected."""
    from transformers import CLIPTextConfig, CLIPTextModel
    with torch.device("meta"):
        hf = CLIPTextModel(CLIPTextConfig(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                                          num_attention_heads=12, max_position_embeddings=77, projection_dim=768,
                                          bos_token_id=49406, eos_token_id=49407, pad_token_id=1))
        oc = OpenClipTextTower()
    _seed_on_device(hf, device, dtype, seed)
    _seed_on_device(oc, device, dtype, seed + 1)
    from rsvld_amd import llava_next as LN
    LN._materialise_derived_buffers(hf, force=True)         # CLIP text position_ids
    hf.to(device)
    oc.attn_mask = torch.full((77, 77), float("-inf"), device=device, dtype=dtype).triu_(1)
    tok = HashTokenizer()
    mod = "rsvld_amd.sgm.modules.encoders.modules."
    size = {"target": mod + "ConcatTimestepEmbedderND", "params": {"outdim": 256}, "is_trainable": False}
    return {"target": "rsvld_amd.sgm.modules.GeneralConditionerWithControl", "params": {"emb_models": [
        {"is_trainable": False, "input_key": "txt", "target": mod + "FrozenCLIPEmbedder",
         "params": {"layer": "hidden", "layer_idx": 11, "device": str(device), "tokenizer": tok, "transformer": hf.eval()}},
        {"is_trainable": False, "input_key": "txt", "target": mod + "FrozenOpenCLIPEmbedder2",
         "params": {"arch": "ViT-bigG-14", "version": "laion2b_s39b_b160k", "freeze": True, "layer": "penultimate",
                    "always_return_pooled": True, "legacy": False, "device": str(device), "model": oc.eval(), "tokenize": tok}},
        dict(size, input_key="original_size_as_tuple"), dict(size, input_key="crop_coords_top_left"),
        dict(size, input_key="target_size_as_tuple")]}}
