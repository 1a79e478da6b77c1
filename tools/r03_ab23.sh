#!/bin/bash
# round-3 run 23 (one box): d = 512 attention with a branch-free steady-state copy of the loop body (in-tree) vs one general body (nosteady)
R=$(pwd); out=$R/gpurun_out; log=$out/r03_ab23.log; : > $log
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q -x -k "attention" >> $log 2>&1 || { echo "TESTS FAILED" >> $log; tail -30 $log; exit 1; }
python3 - >> $log 2>&1 <<'PY'
import os, sys, subprocess
code = """
import os, sys, torch, hashlib
sys.path.insert(0, os.getcwd())
from rsvld_amd import ops
dev = torch.device('cuda:0'); torch.manual_seed(3)
h = hashlib.sha256()
for (B, N, shared) in [(1, 4096, True), (2, 1000, False), (1, 16384 + 37, True), (1, 333, False), (1, 96, True)]:
    q = torch.randn(B, N, 512, device=dev, dtype=torch.float16) * 0.3
    k = torch.randn(B, N, 512, device=dev, dtype=torch.float16) * 0.3
    v = k if shared else torch.randn(B, N, 512, device=dev, dtype=torch.float16)
    h.update(ops.attention(q, k, v, heads=1, scale=512 ** -0.5).cpu().numpy().tobytes())
print(h.hexdigest())
"""
outs = []
for lib in ("", os.path.join(os.getcwd(), "tools/ablate/librsvld_d512_nosteady.so")):
    env = dict(os.environ)
    if lib: env["RSVLD_LIB"] = lib
    outs.append(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1])
print("bit-identity of the two builds:", outs[0] == outs[1], outs)
PY
for rep in 1 2; do
for lib in "" d512_nosteady; do
  echo "== d512, library: ${lib:-in-tree (steady-state loop copy)} (pass $rep)" >> $log
  SHARED=1 ONLY512=1 HEADLINE=1 REPS=3 RSVLD_LIB=${lib:+$R/tools/ablate/librsvld_$lib.so} timeout -k 10 200 python3 tools/bench_attn.py >> $log 2>&1
done
done
