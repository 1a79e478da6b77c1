#!/bin/bash
# round-3 runs 21/22 (one box): gemm256 epilogue changes, in-tree vs the previous commit (prev)
R=$(pwd); out=$R/gpurun_out; log=$out/r03_ab22.log; : > $log
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q -x -k "gemm or linear or geglu or deterministic or identity" >> $log 2>&1 || { echo "TESTS FAILED" >> $log; tail -30 $log; exit 1; }
python3 - >> $log 2>&1 <<'PY'
# bit-identity of the two builds on the three activation paths, with and without residual / alpha
import os, sys, subprocess, torch
code = """
import os, sys, torch, hashlib
sys.path.insert(0, os.getcwd())
from rsvld_amd import ops
dev = torch.device('cuda:0'); torch.manual_seed(3)
h = hashlib.sha256()
for (M, K, N, act, res) in [(8192, 1280, 1280, 0, True), (8192, 1280, 3840, 0, False), (8192, 1280, 10240, 2, False), (8192, 640, 1280, 1, False), (5000, 1280, 1000, 0, True)]:
    x = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K) / K ** 0.5
    pc = ops.pack_conv(w, torch.randn(N), torch.float16, dev, geglu=(act == 2))
    r = torch.randn(M, N, device=dev, dtype=torch.float16) if res else None
    y = ops.linear(x, pc, act=act, residual=r)
    h.update(y.cpu().numpy().tobytes())
print(h.hexdigest())
"""
outs = []
for lib in ("", os.path.join(os.getcwd(), "tools/ablate/librsvld_gemm_prev.so")):
    env = dict(os.environ)
    if lib: env["RSVLD_LIB"] = lib
    outs.append(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1])
print("bit-identity of the two builds:", outs[0] == outs[1], outs)
PY
for rep in 1 2; do
  for lib in "" tools/ablate/librsvld_gemm_prev.so; do
    echo "== gemm256, library: ${lib:-in-tree (specialised epilogue)} (pass $rep)" >> $log
    HEADLINE=1 REPS=10 RSVLD_LIB=${lib:+$R/$lib} timeout -k 10 200 python3 tools/bench_linear.py >> $log 2>&1
  done
done
