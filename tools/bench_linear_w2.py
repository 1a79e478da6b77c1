"""Times ops.linear in the weight-pair form (fp16 activation x fp16 weight pair [W_lo | W_hi], RSVLD_F16W2) on the Stage-2 transformer GEMMs of
the headline (latent 512, CFG pair: 32 768 tokens x 1 280 channels, 131 072 x 640) under ops.UNET_POLICY (PAIRS=1: every weight a pair).
TF/s = algorithmic rate x the MFMAs per product of the form that ran (1 with fp16 weights, 2 with pairs).
RSVLD_LIB=<another build> selects a different library for A/B runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsvld_amd import ops
from rsvld_amd import _lib as L

dev = torch.device("cuda:0")
torch.manual_seed(0)
# M, K, N, kind: "f16" = fp16 out (q | k | v, the GEGLU feed-forward), "res" = fp32 out + fp32 residual (to_out, ff.net.2)
SHAPES = [(32768, 1280, 3840, "f16"), (32768, 1280, 10240, "geglu"), (32768, 1280, 1280, "res"), (32768, 5120, 1280, "res"),
          (131072, 640, 1920, "f16"), (131072, 640, 5120, "geglu"), (131072, 640, 640, "res"), (131072, 2560, 640, "res")]
reps = int(os.environ.get("REPS", 10))
POLICY = ops.SplitPolicy(f16_weights=()) if os.environ.get("PAIRS") else ops.UNET_POLICY
NM = 2.0 if os.environ.get("PAIRS") else 1.0
print("library:", os.environ.get("RSVLD_LIB", "in-tree build"))
tot = 0.0
for (M, K, N, kind) in SHAPES:
    x = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K) / K ** 0.5
    geglu = kind == "geglu"
    pc = ops.pack_conv(w, torch.zeros(N), torch.float32, dev, geglu=geglu)
    res = torch.randn(M, N, device=dev) if kind == "res" else None

    def run():
        with ops.f32_split(POLICY):
            return ops.linear(x, pc, residual=res, act=L.ACT_GEGLU if geglu else L.ACT_NONE, out_planes=kind != "res", out_group="ff",
                              group="ff_out" if kind == "res" else None)   # (the policy decides pairs or fp16 weights: UNET_POLICY rounds all four groups)
    for _ in range(3):
        y = run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        y = run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tot += ms
    fl = NM * 2.0 * M * K * N
    print(f"M{M} K{K} N{N} {kind:5s} -> {str(y.dtype)[6:]}: {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} MFMA-TF/s", flush=True)
    del x, res, y
print(f"sum {tot:.3f} ms")
