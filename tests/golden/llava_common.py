"""Shared by the LLaVA-Next golden generator and the tests: a tiny seeded LLaMA + CLIP-vision configuration, an in-memory
tokenizer with the Llama-3 chat template, and the input recipes.  Nothing here needs a checkpoint or the network."""
import json
import os

import numpy as np
import torch

WEIGHT_SEED = 2468
VISION = dict(hidden_size=32, intermediate_size=64, num_hidden_layers=3, num_attention_heads=4, image_size=56, patch_size=14,
              projection_dim=16)
LLAMA = dict(vocab_size=384, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
             num_key_value_heads=2, max_position_embeddings=2048, rms_norm_eps=1e-5, rope_theta=10000.0,
             bos_token_id=1, eos_token_id=2, pad_token_id=0, tie_word_embeddings=False)
MM = dict(mm_projector_type="mlp2x_gelu", mm_hidden_size=32, mm_vision_select_layer=-2, mm_vision_select_feature="patch",
          image_aspect_ratio="anyres", mm_patch_merge_type="spatial_unpad", mm_use_im_start_end=False,
          mm_use_im_patch_token=False, unfreeze_mm_vision_tower=True, tokenizer_padding_side="right",
          image_grid_pinpoints=[[56, 112], [112, 56], [112, 112], [168, 56], [56, 168]])
SYSTEM = ("You are a helpful language and vision assistant. You are able to understand the visual content that the user provides, "
          "and assist the user with a variety of tasks using natural language.")   # llava/conversation.py:388
QUESTION = "<image>\ndescribe every visible detail of the aerial image : terrain , roads and buildings ."
# the Llama-3 chat template (tokenizer_config.json of meta-llama/Meta-Llama-3-8B-Instruct), which conv_llava_llama_3 applies
CHAT_TEMPLATE = ("{% set loop_messages = messages %}{% for message in loop_messages %}{% set content = '<|start_header_id|>' + "
                 "message['role'] + '<|end_header_id|>\n\n'+ message['content'] | trim + '<|eot_id|>' %}{% if loop.index0 == 0 %}"
                 "{% set content = bos_token + content %}{% endif %}{{ content }}{% endfor %}{% if add_generation_prompt %}"
                 "{{ '<|start_header_id|>assistant<|end_header_id|>\n\n' }}{% endif %}")
IMAGE_SIZES = [(100, 70), (60, 150), (56, 56)]     # (width, height): landscape, portrait, exactly one tile


def build_tokenizer():
    """A 384-entry word-level tokenizer with Llama-3's special tokens and chat template."""
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    specials = ["<pad>", "<|begin_of_text|>", "<|eot_id|>", "<unk>", "<|start_header_id|>", "<|end_header_id|>"]
    words = sorted(set((SYSTEM + " " + QUESTION.replace("<image>", " ") + " user assistant system \n \n\n").replace(",", " , ")
                       .replace(".", " . ").split(" ")) - {""})
    vocab = {t: i for i, t in enumerate(specials)}
    for w in words + ["\n", "\n\n", "Ċ", "ĊĊ"]:
        vocab.setdefault(w, len(vocab))
    i = 0
    while len(vocab) < LLAMA["vocab_size"]:
        vocab.setdefault(f"w{i}", len(vocab))
        i += 1
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.Split(" ", "removed"), pre_tokenizers.Split("\n", "isolated")])
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, bos_token="<|begin_of_text|>", eos_token="<|eot_id|>", pad_token="<pad>",
                                   unk_token="<unk>", additional_special_tokens=["<|start_header_id|>", "<|end_header_id|>"])
    fast.chat_template = CHAT_TEMPLATE
    return fast


def test_image(size, seed):
    """uint8 RGB PIL image of ``size`` = (width, height): seeded low-pass noise."""
    from PIL import Image
    rng = np.random.default_rng(seed)
    a = rng.random((size[1] // 4 + 2, size[0] // 4 + 2, 3))
    a = np.kron(a, np.ones((4, 4, 1)))[:size[1], :size[0]]
    return Image.fromarray((a * 255).astype(np.uint8))


def save_tiny_clip(path):
    """Write a tiny CLIP vision tower (config + seeded weights + image processor config) to ``path``."""
    from transformers import CLIPVisionConfig, CLIPVisionModel
    os.makedirs(path, exist_ok=True)
    torch.manual_seed(11)
    m = CLIPVisionModel(CLIPVisionConfig(**VISION))
    m.save_pretrained(path)
    json.dump({"crop_size": {"height": VISION["image_size"], "width": VISION["image_size"]}, "do_center_crop": True,
               "do_convert_rgb": True, "do_normalize": True, "do_rescale": True, "do_resize": True,
               "image_mean": [0.48145466, 0.4578275, 0.40821073], "image_std": [0.26862954, 0.26130258, 0.27577711],
               "image_processor_type": "CLIPImageProcessor", "resample": 3, "rescale_factor": 1 / 255,
               "size": {"shortest_edge": VISION["image_size"]}}, open(os.path.join(path, "preprocessor_config.json"), "w"))
    return path


def name_seeded_state(module, seed):
    """Seeded weights that depend on each parameter's NAME only (not on registration order, which differs between the
    reference's module tree and the product's): N(0, 1/fan_in) matrices, 1 + 0.1 N gains, 0.05 N biases / vectors."""
    import zlib
    out = {}
    for name, p in module.named_parameters():
        key = name.replace(".vision_tower.vision_tower.vision_model.", ".vision_tower.vision_tower.")   # transformers 4 / 5 naming
        g = torch.Generator().manual_seed((zlib.crc32(key.encode()) + seed) % (2 ** 31))
        if p.dim() >= 2:
            fan_in = int(np.prod(p.shape[1:]))
            out[name] = torch.randn(p.shape, generator=g) / np.sqrt(fan_in)
        elif name.endswith("weight") and p.dim() == 1:
            out[name] = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
        else:
            out[name] = 0.05 * torch.randn(p.shape, generator=g)
    with torch.no_grad():
        for name, p in module.named_parameters():
            p.copy_(out[name])
    return out


class PlainProcessor:
    """CLIP image processor seen through plain dicts: transformers 5 exposes SizeDict objects where the reference
    (written against transformers 4.43) indexes dicts."""

    def __init__(self, p):
        self._p = p
        self.size = {k: v for k, v in dict(p.size).items() if v is not None}
        self.crop_size = {k: v for k, v in dict(p.crop_size).items() if v is not None}
        self.image_mean = p.image_mean

    def preprocess(self, *a, **k):
        return self._p.preprocess(*a, **k)
