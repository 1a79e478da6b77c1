"""gemm_stamps.py for the persistent weight-pair GEMMs (RSVLD_F16W2): per tile, the K loop and the epilogue (100 MHz s_memrealtime stamps of
thread 0 of every workgroup; -DG_STAMP=1 build selected through RSVLD_LIB).  A diagnostic, not part of the product."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rsvld_amd import ops
from rsvld_amd import _lib as L

dev = torch.device("cuda:0")
torch.manual_seed(0)
SHAPES = [(32768, 1280, 3840, "f16"), (32768, 1280, 10240, "geglu"), (32768, 1280, 1280, "res"), (32768, 5120, 1280, "res"),
          (131072, 640, 1920, "f16"), (131072, 640, 640, "res")]
lib = ctypes.CDLL(os.environ["RSVLD_LIB"])
lib.rsvld_debug_gemm_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
for (M, K, N, kind) in SHAPES:
    x = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K) / K ** 0.5
    geglu = kind == "geglu"
    pc = ops.pack_conv(w, torch.zeros(N), torch.float32, dev, geglu=geglu)
    res = torch.randn(M, N, device=dev) if kind == "res" else None

    def run():
        with ops.f32_split(ops.UNET_POLICY):
            return ops.linear(x, pc, residual=res, act=L.ACT_GEGLU if geglu else L.ACT_NONE, out_planes=kind != "res", out_group="ff")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    ntile = ((M + 255) // 256) * ((N + 255) // 256)
    buf = np.zeros(16384 * 8, dtype=np.uint64)
    assert lib.rsvld_debug_gemm_stamps(buf.ctypes.data, buf.nbytes) == 0
    s = buf.reshape(-1, 8)[:min(ntile, 16384)].astype(np.int64)
    s = s[s[:, 1] > 0]
    kl = (s[:, 2] - s[:, 1]) * 0.01
    ep = (s[:, 5] - s[:, 2]) * 0.01
    cyc = s[:, 6] / np.maximum(kl, 1e-3)
    nkp = 2 * K // 32
    ms = e0.elapsed_time(e1)
    print(f"M{M} K{K} N{N} {kind}: launch {ms*1e3:7.1f} us ({4.0*M*K*N/ms/1e9:6.1f} MFMA-TF/s), {len(s)} stamped tiles ({ntile} tiles / 256 CUs = {ntile/256:.2f} rounds)\n"
          f"    per tile, us (mean / p90): K loop {kl.mean():6.2f}/{np.percentile(kl,90):6.2f} = {kl.mean()/nkp*1e3:5.1f} ns = {cyc.mean()*kl.mean()/nkp:5.0f} cycles per K' tile (512 = the matrix pipe's 16 MFMAs per wave)"
          f"   epilogue {ep.mean():6.2f}/{np.percentile(ep,90):6.2f}   [{cyc.mean():5.0f} cycles/us]", flush=True)
    del x, res
