"""CPU experiment, not a test and not product code (the oracle is test infrastructure, which is why this lives under tests/): which
tensors of Stage 1 need more than 16 bits.  T = 50 at 256 px on the CPU restatement of the reference with chosen tensors rounded to fp16
on the way (fp32 arithmetic, fp32 weights), against the reference-generated golden -- the distance a precision mode built that way
would have from the reference's CPU run.  Usage:  python tests/experiments/operand_bits_stage1.py none | f16 | bf16 | attnonly
(KEEP=attn / KEEP=gn exempt the attention / GroupNorm outputs in the f16 / bf16 modes).  ~25 s on 8 cores.  Results: DESIGN.md,
"Which operands need more than 16 bits"."""
import os, sys, math, time
import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT); sys.path.insert(0, GOLDEN)
torch.set_num_threads(8)
from oracle import seeded, sr3_oracle as O
from rsvld_amd.sr3_model.sr3_modules.diffusion import GaussianDiffusion
from rsvld_amd.sr3_model.sr3_modules.unet import UNet
c = O.SR3_CFG
unet = UNet(in_channel=c["in_channel"], out_channel=c["out_channel"], inner_channel=c["inner_channel"], norm_groups=c["norm_groups"],
            channel_mults=c["channel_mults"], attn_res=list(c["attn_res"]), res_blocks=c["res_blocks"], dropout=0.2, image_size=c["image_size"])
net = GaussianDiffusion(unet, image_size=c["image_size"], channels=3, conditional=True)
seeded.seed_module(net, 1234)
sd = {k: v.float() for k, v in net.state_dict().items()}
sch = O.schedule(dict(schedule="linear", n_timestep=50, linear_start=1e-6, linear_end=1e-2))
z = np.load(os.path.join(GOLDEN, "sr3_pipeline_t50.npz"))
cond = torch.tensor(np.load(os.path.join(GOLDEN, "sr3_pipeline_c1.npz"))["cond"])
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
r16 = (lambda t: t.half().float()) if mode == "f16" else (lambda t: t.bfloat16().float()) if mode == "bf16" else (lambda t: t)
if mode == "attnonly":
    r = lambda t: t.half().float()
    def self_attention(sd, p, x, groups, q_chunk=1024):
        import math
        B, Cc, H, W = x.shape
        n = F.group_norm(x, groups, sd[p + ".norm.weight"], sd[p + ".norm.bias"], eps=1e-5)
        qkv = r(F.conv2d(n, sd[p + ".qkv.weight"])).view(B, 1, Cc * 3, H * W)
        q, k, v = qkv[:, 0].chunk(3, dim=1)
        out = torch.empty_like(q)
        for s_ in range(0, H * W, q_chunk):
            a = torch.einsum("bcq,bck->bqk", q[:, :, s_:s_ + q_chunk], k) / math.sqrt(Cc)
            a = r(torch.softmax(a, dim=-1))
            out[:, :, s_:s_ + q_chunk] = torch.einsum("bqk,bck->bcq", a, v)
        out = F.conv2d(r(out).view(B, Cc, H, W), sd[p + ".out.weight"], sd[p + ".out.bias"])
        return out + x
    O.self_attention = self_attention
elif mode != "none":
    _conv, _lin, _gn, _ein, _sm = F.conv2d, F.linear, F.group_norm, torch.einsum, torch.softmax
    F.conv2d = lambda *a, **k: r16(_conv(*a, **k))
    F.linear = lambda *a, **k: r16(_lin(*a, **k))
    if "gn" not in os.environ.get("KEEP", ""):
        F.group_norm = lambda *a, **k: r16(_gn(*a, **k))
    if "attn" not in os.environ.get("KEEP", ""):
        torch.einsum = lambda *a, **k: r16(_ein(*a, **k))
        torch.softmax = lambda *a, **k: r16(_sm(*a, **k))
    _sw = O.swish
    O.swish = lambda x: r16(_sw(x))
torch.manual_seed(int(z["torch_seed"]))
t0 = time.time()
with torch.no_grad():
    out = O.p_sample_loop(sd, c, sch, cond, continous=True)
d = (out[-1:] - torch.tensor(z["final"])).abs()
print(mode, os.environ.get("KEEP", ""), "max", float(d.max()), "mean", float(d.mean()), f"{time.time()-t0:.0f} s")
