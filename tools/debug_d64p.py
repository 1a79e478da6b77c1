import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsvld_amd import ops
dev = torch.device("cuda:0")
D, heads = 64, 1
for Nk, spike in [(128, None), (192, None), (256, None), (320, None), (384, None), (320, 300), (320, 200), (320, 100), (320, 70), (384, 300), (384, 370), (256, 250), (256, 130), (192, 190), (192, 100), (640, 600), (640, 330)]:
    g = torch.Generator().manual_seed(Nk)
    q = torch.randn(1, 64, heads * D, generator=g)
    k = torch.randn(1, Nk, heads * D, generator=g)
    v = torch.randn(1, Nk, heads * D, generator=g)
    if spike is not None:
        k[0, spike] = q[0, 17] * 4.0
    q, k, v = (t.to(dev, torch.float16) for t in (q, k, v))
    outs = {}
    for kern in ("b", "p"):
        os.environ["RSVLD_D64_KERNEL"] = kern
        outs[kern] = ops.attention(q, k, v, heads=heads).float().cpu()
    d = (outs["b"] - outs["p"]).abs().amax(-1)[0]
    bad = torch.nonzero(d > 0).flatten().tolist()
    print(f"Nk={Nk} nt={(Nk+63)//64} spike={spike}: max diff {float(d.max()):.3e}, differing rows {bad[:10]}{'...' if len(bad) > 10 else ''}", flush=True)
print("---- detail: Nk=192, spike=190")
Nk, spike = 192, 190
g = torch.Generator().manual_seed(Nk)
q = torch.randn(1, 64, heads * D, generator=g); k = torch.randn(1, Nk, heads * D, generator=g); v = torch.randn(1, Nk, heads * D, generator=g)
k[0, spike] = q[0, 17] * 4.0
qd, kd, vd = (t.to(dev, torch.float16) for t in (q, k, v))
outs = {}
for kern in ("b", "p"):
    os.environ["RSVLD_D64_KERNEL"] = kern
    outs[kern] = ops.attention(qd, kd, vd, heads=heads).float().cpu()[0]
qf, kf, vf = (t.to(torch.float16).double()[0] for t in (q, k, v))
ref = torch.softmax(qf @ kf.T / 8.0, -1) @ vf
# reference with the last tile's contribution doubled / missing etc.
s = (qf @ kf.T / 8.0)
w = torch.softmax(s, -1)
last = w[:, 128:] @ vf[128:]
first = w[:, :128] @ vf[:128]
for r in (0, 1, 17, 40):
    b_, p_ = outs["b"][r], outs["p"][r]
    print(r, "b-ref", float((b_ - ref[r]).abs().max()), "p-ref", float((p_ - ref[r]).abs().max()),
          "p-(first only, renormalised)", float((p_ - first[r] / w[r, :128].sum()).abs().max()),
          "p-(ref + last again)", float((p_ - (ref[r] + last[r])).abs().max()),
          "p-(last only renorm)", float((p_ - last[r] / w[r, 128:].sum()).abs().max()))
print("---- hypotheses")
import math
LOG2E = 1.4426950408889634
sc = (qf * (torch.tensor(0.125 * LOG2E, dtype=torch.float16).double())).to(torch.float16).double() @ kf.T      # log2 units, Q pre-scaled in fp16
for r in (0, 1, 2, 3, 17, 40):
    s_ = sc[r]
    m_old = float(torch.tensor(float(s_[:64].max())).to(torch.float16))     # tile 0 sets it; tiles 1 stays unless > 2^14
    e_prev = torch.exp2(s_[:128] - m_old)
    O_prev = e_prev @ vf[:128]; l_prev = e_prev.sum()
    mx = float(s_[128:].max()) - m_old
    m_new = float(torch.tensor(m_old + mx).to(torch.float16)) if mx > 0 else m_old
    alpha = 2.0 ** (-(m_new - m_old))
    e_last = torch.exp2(s_[128:] - m_new)
    PV = e_last @ vf[128:]; rs = e_last.sum()
    good = (alpha * O_prev + PV) / (alpha * l_prev + rs)
    h1 = (O_prev + PV) / (alpha * l_prev + rs)
    h2 = (alpha * O_prev + PV) / (l_prev + rs)
    h3 = (alpha * alpha * O_prev + PV) / (alpha * l_prev + rs)
    e0_ = torch.exp2(s_[:64] - m_old); e1_ = torch.exp2(s_[64:128] - m_old)
    O0 = e0_ @ vf[:64]; O1 = e1_ @ vf[64:128]
    den = alpha * l_prev + rs
    p_ = outs["p"][r].double()
    print(r, "PV1 unscaled", float((p_ - (alpha * O0 + O1 + PV) / den).abs().max()), "PV1 lost", float((p_ - (alpha * O0 + PV) / den).abs().max()),
          "P_last relative to m_old", float((p_ - (alpha * O_prev + (torch.exp2(s_[128:] - m_old).to(torch.float16).double() @ vf[128:])) / den).abs().max()),
          "l: rs relative to m_old", float((p_ - (alpha * O_prev + PV) / (alpha * l_prev + torch.exp2(s_[128:] - m_old).sum())).abs().max()))
    print(r, f"alpha {alpha:.4f}", "good", float((p_ - good).abs().max()), "O unscaled", float((p_ - h1).abs().max()), "l unscaled", float((p_ - h2).abs().max()), "O scaled twice", float((p_ - h3).abs().max()))
print("---- per-element |p - b| rows 0..3 (64 d values)")
for r in (0, 1, 2):
    d = (outs["p"][r] - outs["b"][r]).abs()
    print(r, " ".join(f"{float(x):.0e}" if x > 0 else "0" for x in d))
