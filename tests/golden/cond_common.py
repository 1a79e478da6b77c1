"""Small seeded text towers + fixed token tables shared by the conditioner golden generator (which runs the
REFERENCE's embedder code over them) and tests/test_conditioner.py (which runs this repo's embedders over the same
objects).  The towers only mimic the attribute layout of the real ones (HuggingFace CLIPTextModel is used as is;
the OpenCLIP stand-in exposes token_embedding / positional_embedding / transformer.resblocks / ln_final /
text_projection / attn_mask): what is pinned is the embedder logic around them (layer selection, ln_final placement,
eot pooling, concatenation order), not the towers' weights."""
import torch
from torch import nn

CTX = 77
PROMPTS = {"a satellite image of a harbour, sharp": 0, "blurry, low quality": 1, "": 2}


def token_table(vocab, seed, eot):
    """ids [3, 77] for the three prompts: bos, random words, eot (= highest id, as in CLIP), padding"""
    g = torch.Generator().manual_seed(seed)
    ids = torch.zeros(len(PROMPTS), CTX, dtype=torch.long)
    for r, n in enumerate((9, 5, 0)):
        ids[r, 0] = eot - 1
        ids[r, 1:1 + n] = torch.randint(1, eot - 1, (n,), generator=g)
        ids[r, 1 + n] = eot
    return ids


class FakeHFTokenizer:
    """callable with the CLIPTokenizer call signature the embedder uses; looks the prompt up in PROMPTS"""

    def __init__(self, ids):
        self.ids = ids

    def __call__(self, text, **kw):
        assert kw.get("padding") == "max_length" and kw.get("max_length") == CTX and kw.get("return_tensors") == "pt"
        text = [text] if isinstance(text, str) else list(text)
        return {"input_ids": torch.stack([self.ids[PROMPTS[t]] for t in text])}


def make_hf_clip(seed=11):
    from transformers import CLIPTextConfig, CLIPTextModel
    cfg = CLIPTextConfig(vocab_size=64, hidden_size=32, intermediate_size=64, num_hidden_layers=3, num_attention_heads=2,
                         max_position_embeddings=CTX, projection_dim=16, bos_token_id=62, eos_token_id=63, pad_token_id=0)
    torch.manual_seed(seed)
    m = CLIPTextModel(cfg).eval()
    return m, FakeHFTokenizer(token_table(64, seed, 63))


class _Block(nn.Module):
    def __init__(self, width, heads):
        super().__init__()
        self.ln_1 = nn.LayerNorm(width)
        self.attn = nn.MultiheadAttention(width, heads)
        self.ln_2 = nn.LayerNorm(width)
        self.mlp = nn.Sequential(nn.Linear(width, 4 * width), nn.GELU(), nn.Linear(4 * width, width))

    def forward(self, x, attn_mask=None):
        h = self.ln_1(x)
        x = x + self.attn(h, h, h, need_weights=False, attn_mask=attn_mask)[0]
        return x + self.mlp(self.ln_2(x))


class TinyOpenClipText(nn.Module):
    def __init__(self, vocab=80, width=48, heads=3, layers=4, proj=40):
        super().__init__()
        self.token_embedding = nn.Embedding(vocab, width)
        self.positional_embedding = nn.Parameter(torch.randn(CTX, width) * 0.01)
        self.transformer = nn.Module()
        self.transformer.resblocks = nn.ModuleList([_Block(width, heads) for _ in range(layers)])
        self.transformer.grad_checkpointing = False
        self.ln_final = nn.LayerNorm(width)
        self.text_projection = nn.Parameter(torch.randn(width, proj) * width ** -0.5)
        self.register_buffer("attn_mask", torch.full((CTX, CTX), float("-inf")).triu_(1), persistent=False)


def make_open_clip(seed=12):
    torch.manual_seed(seed)
    m = TinyOpenClipText().eval()
    ids = token_table(80, seed, 79)

    def tokenize(text):
        text = [text] if isinstance(text, str) else list(text)
        return torch.stack([ids[PROMPTS[t]] for t in text])

    return m, tokenize


def batches():
    """the conditioning batch SR_backbone.prepare_condition builds (models/SR_model.py:127-156), two images"""
    g = torch.Generator().manual_seed(13)
    control = torch.randn(2, 4, 8, 8, generator=g)
    size = torch.tensor([[1024.0, 1024.0], [1024.0, 1024.0]])
    batch = {"original_size_as_tuple": size, "crop_coords_top_left": torch.zeros(2, 2), "target_size_as_tuple": size.clone(),
             "control": control, "txt": ["a satellite image of a harbour, sharp", ""]}
    batch_uc = dict(batch, txt=["blurry, low quality", "blurry, low quality"])
    return batch, batch_uc
