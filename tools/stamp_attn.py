"""Where a loop iteration of the d = 64 attention kernel spends its cycles (diagnostic build, tools/ablate/librsvld_stamp.so =
attention.hip built with -DA6B_STAMP=1; cdna_hip_programming.md section 7 "In-kernel stamps").  Prints the median over waves of
the per-iteration cycles of each phase at the two headline shapes.  Read the shares, not the total (stamps forbid overlaps).
    RSVLD_LIB=$PWD/tools/ablate/librsvld_stamp.so python3 tools/stamp_attn.py"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsvld_amd import _lib as L

dev = torch.device("cuda:0")
lib = L.load()
PH = ["request (LDS-DMA issue)", "K reads + S chain (10 MFMA)", "softmax VALU", "V reads + PV chain (8 MFMA)", "wait + barrier"]
PINGPONG = os.environ.get("RSVLD_D64_KERNEL") == "c"     # attn_d64c: 8 waves x 64 query rows, two segments per tile
if PINGPONG:
    PH = ["M segment: request, PV (16 MFMA), K + Q reads, S (20 MFMA)", "V segment: softmax of 2 x 32 rows, V reads", "vmcnt wait + barrier"]
for (B, heads, N) in [(2, 20, 16384), (2, 10, 65536)]:
    D = 64
    torch.manual_seed(0)
    qkv = torch.randn(B, N, 3 * heads * D, device=dev, dtype=torch.float16)
    q, k, v = qkv[..., :heads * D], qkv[..., heads * D:2 * heads * D], qkv[..., 2 * heads * D:]
    out = torch.empty(B, N, heads * D, device=dev, dtype=torch.float16)
    nw = 8 if N >= 16384 else 4      # A6B_NW8_MIN
    rows = 64 if PINGPONG else 32
    nw = 8 if PINGPONG else nw
    nwg = (N + rows * nw - 1) // (rows * nw) * heads * B
    dbg = torch.zeros(nwg * nw * 8, device=dev, dtype=torch.int64)
    p = lambda t: C.c_void_p(t.data_ptr())
    for _ in range(3):   # the last launch's stamps are read (clock settled)
        rc = lib.rsvld_attention_tuned(p(q), p(k), p(v), p(out), B, heads, N, N, D, q.stride(0), q.stride(1), k.stride(0), k.stride(1),
                                       v.stride(0), v.stride(1), out.stride(0), out.stride(1), C.c_float(D ** -0.5), L.F16, 1, p(dbg),
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream), 2 if PINGPONG else 1)
        assert rc == 0, rc
    torch.cuda.synchronize()
    d = dbg.view(-1, 8).cpu().double()
    nt = d[:, 5].clamp(min=1)
    per = d[:, :len(PH)] / nt[:, None]
    med = per.median(0).values
    if PINGPONG:   # early / late halves separately as well
        w = torch.arange(d.shape[0]) % 8
        print("    early half (waves 0-3):", [round(x) for x in per[w < 4].median(0).values.tolist()], " late half (4-7):", [round(x) for x in per[w >= 4].median(0).values.tolist()])
    tot = float(med.sum())
    print(f"B{B} heads{heads} N{N} ({nw}-wave workgroups, {int(nt[0])} sub-tiles): {tot:.0f} cycles per 64-key sub-tile and wave (stamped build)")
    for name, c in zip(PH, med.tolist()):
        print(f"    {name:32s} {c:8.0f} cycles  {100 * c / tot:5.1f} %")
