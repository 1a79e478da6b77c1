"""Where a gemm256 tile's time goes: runs the G_STAMP build (tools/build_variant.sh gstamp gemm.hip -DG_STAMP=1, selected through
RSVLD_LIB) on the headline shapes and prints, per shape, the mean duration of each phase of a workgroup (100 MHz s_memrealtime stamps
of thread 0: start | ring filled | K loop done | tile staged in LDS | stores issued | stores acknowledged) and the gap between two
consecutive workgroups on the same CU.  A diagnostic, not part of the product."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rsvld_amd import ops
from rsvld_amd import _lib as L

dev = torch.device("cuda:0")
torch.manual_seed(0)
SHAPES = [(32768, 1280, 1280, 0), (32768, 1280, 3840, 0), (32768, 1280, 10240, 2), (32768, 5120, 1280, 0), (32768, 2048, 2560, 0),
          (131072, 640, 640, 0), (131072, 640, 1920, 0), (131072, 640, 5120, 2), (131072, 2560, 640, 0)]
lib = ctypes.CDLL(os.environ["RSVLD_LIB"])
lib.rsvld_debug_gemm_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
for (M, K, N, act) in SHAPES:
    for res in (False, True):
        if res and act == 2:
            continue
        x = torch.randn(M, K, device=dev, dtype=torch.float16)
        w = torch.randn(N, K) / K ** 0.5
        pc = ops.pack_conv(w, torch.zeros(N), torch.float16, dev, geglu=(act == 2))
        r = torch.randn(M, N, device=dev, dtype=torch.float16) if res else None
        for _ in range(3):
            ops.linear(x, pc, act=act, residual=r)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.linear(x, pc, act=act, residual=r)
        e1.record()
        torch.cuda.synchronize()
        nwg = ((M + 255) // 256) * ((N + 255) // 256)
        buf = np.zeros(16384 * 8, dtype=np.uint64)
        assert lib.rsvld_debug_gemm_stamps(buf.ctypes.data, buf.nbytes) == 0
        s = buf.reshape(-1, 8)[:min(nwg, 16384)].astype(np.int64)
        t = (s[:, :6] - s[:, :1].min()) * 0.01          # us since the first workgroup's start
        ph = np.diff(t, axis=1)                         # fill | K loop | stage | store issue | store ack
        key = ((s[:, 7] >> 32) << 8) | ((s[:, 7] >> 8) & 0xff)
        gaps, per_cu = [], []
        for k in np.unique(key):
            tt = t[key == k]
            tt = tt[np.argsort(tt[:, 0])]
            per_cu.append(len(tt))
            gaps += list(tt[1:, 0] - tt[:-1, 5])
        gaps = np.array(gaps) if gaps else np.zeros(1)
        rr_ok = float(((s[:, 7] >> 32) & 0xf == (np.arange(len(s)) & 7)).mean())      # share of workgroups on XCD = linear id mod 8
        cyc = s[:, 6] / np.maximum(ph[:, 1], 1e-3)                                     # s_memtime ticks per us over the K loop
        fl = 2.0 * M * K * N
        print(f"M{M} K{K} N{N} act{act} res{int(res)}: launch {e0.elapsed_time(e1)*1e3:7.1f} us ({fl/e0.elapsed_time(e1)/1e9:6.1f} TF/s), stamps span {t[:, 5].max():7.1f} us, "
              f"{nwg} WGs on {len(per_cu)} CUs (max {max(per_cu)} per CU)\n"
              f"    per workgroup, us (mean / p90):  ring fill {ph[:,0].mean():5.2f}/{np.percentile(ph[:,0],90):5.2f}   K loop {ph[:,1].mean():6.2f}/{np.percentile(ph[:,1],90):6.2f}"
              f"   stage {ph[:,2].mean():5.2f}/{np.percentile(ph[:,2],90):5.2f}   store issue {ph[:,3].mean():5.2f}/{np.percentile(ph[:,3],90):5.2f}"
              f"   store ack {ph[:,4].mean():5.2f}/{np.percentile(ph[:,4],90):5.2f}   gap to next WG on the CU {gaps.mean():5.2f}/{np.percentile(gaps,90):5.2f}"
              f"   [K loop ideal at 2.4 GHz: {2.0*256*256*K/ (2.5e15/256) * 1e6:5.2f}; s_memtime {cyc.mean():6.0f} ticks/us; XCD = id mod 8 for {rr_ok:.3f}]", flush=True)
