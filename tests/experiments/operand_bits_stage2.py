"""CPU experiment, not a test and not product code (see operand_bits_stage1.py): Stage 2 (the reduced networks of the golden), 50 EDM
steps at 64 px on the CPU restatement of the reference, with chosen tensors rounded to fp16 (fp32 arithmetic, fp32 weights).
MODE: none | attn (q, k, v, the probabilities and the output of every attention) | in16_lin / in16_conv / in16 (attn + the INPUT of every
Linear / convolution / both) | all (attn + every layer output).  40-75 s on 8 cores.  Results: DESIGN.md."""
import os, sys, time
import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT); sys.path.insert(0, GOLDEN)
torch.set_num_threads(8)
import s2_common as S
from oracle import seeded, s2_oracle as O
from rsvld_amd.sgm.util import instantiate_from_config
m = instantiate_from_config({"target": "rsvld_amd.models.SR_model.SR_backbone", "params": S.product_params()})
seeded.seed_module(m, S.WEIGHT_SEED)
sd = {k: v.detach().clone() for k, v in m.eval().state_dict().items()}
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
r16 = lambda t: t.half().float()
if mode in ("attn", "all", "unet_attn"):
    def attention_core(q, k, v, heads):
        B, N, Cq = q.shape; d = Cq // heads
        q, k, v = (r16(t).view(B, -1, heads, d).transpose(1, 2) for t in (q, k, v))
        a = r16(torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5, dim=-1))
        return r16((a @ v).transpose(1, 2).reshape(B, N, Cq))
    O.attention_core = attention_core
if mode in ("in16", "in16_lin", "in16_conv"):
    # attention in fp16 (as the shipped hybrid) + the INPUT of every Linear / conv rounded to fp16 (weights exact, fp32 accumulate,
    # outputs and the residual stream fp32)
    O.attention_core = (lambda f: f)(O.attention_core)
    def attention_core(q, k, v, heads):
        B, N, Cq = q.shape; d = Cq // heads
        q, k, v = (r16(t).view(B, -1, heads, d).transpose(1, 2) for t in (q, k, v))
        a = r16(torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5, dim=-1))
        return r16((a @ v).transpose(1, 2).reshape(B, N, Cq))
    O.attention_core = attention_core
    _conv, _lin = O.conv, O.lin
    if mode in ("in16", "in16_conv"):
        O.conv = lambda sd, p, x, **k: _conv(sd, p, r16(x), **k)
    if mode in ("in16", "in16_lin"):
        O.lin = lambda sd, p, x: _lin(sd, p, r16(x))
if mode.startswith("lin:"):
    # attention in fp16 + the input of the Linear layers whose parameter path contains one of the given substrings (comma-separated)
    def attention_core(q, k, v, heads):
        B, N, Cq = q.shape; d = Cq // heads
        q, k, v = (r16(t).view(B, -1, heads, d).transpose(1, 2) for t in (q, k, v))
        a = r16(torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5, dim=-1))
        return r16((a @ v).transpose(1, 2).reshape(B, N, Cq))
    O.attention_core = attention_core
    keys = mode[4:].split(",")
    _lin = O.lin
    hits = set()
    def lin(sd, p, x):
        if any(k in p for k in keys):
            hits.add(p.split(".")[-2] + "." + p.split(".")[-1] if p.split(".")[-1].isdigit() else p.split(".")[-1])
            return _lin(sd, p, r16(x))
        return _lin(sd, p, x)
    O.lin = lin
if mode in ("attn_self", "attn_cross"):
    # fp16 operands only for the self-attentions (keys = the tokens themselves) or only for the attentions over another sequence
    # (the 77 text tokens; ZeroCrossAttn's control tokens)
    _core = O.attention_core
    def attention_core(q, k, v, heads):
        is_self = q.shape[1] == k.shape[1]
        if is_self != (mode == "attn_self"):
            return _core(q, k, v, heads)
        B, N, Cq = q.shape; d = Cq // heads
        q, k, v = (r16(t).view(B, -1, heads, d).transpose(1, 2) for t in (q, k, v))
        a = r16(torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5, dim=-1))
        return r16((a @ v).transpose(1, 2).reshape(B, N, Cq))
    O.attention_core = attention_core
if mode == "all":
    for name in ("gn", "conv", "lin", "layer_norm"):
        f = getattr(O, name)
        setattr(O, name, (lambda f: (lambda *a, **k: r16(f(*a, **k))))(f))
img = seeded.synthetic_image((1, 3, 64, 64), seed=80, smooth=3)
c, uc = S.cond_dicts()
z = np.load(os.path.join(GOLDEN, "s2_pipeline_50.npz"))
torch.manual_seed(7)
t0 = time.time()
with torch.no_grad():
    out = O.just_sampling(sd, img, c, uc, dict(S.PIPE_OPT, num_steps=50, img_threshold=0.0))
d = (out - torch.tensor(z["nocache50.final"])).abs()
print(mode, "max", float(d.max()), "mean", float(d.mean()), "range", float(torch.tensor(z["nocache50.final"]).abs().max()), f"{time.time()-t0:.0f} s")
