"""Audit: the kernels that stream operands by LDS-DMA must not touch scratch memory.  A spill (or a stack table the compiler builds
for a per-lane pointer select) is reloaded with scratch_load, which shares ``vmcnt`` with the LDS-DMA ring: hipcc then waits vmcnt(0)
in front of the reload and drains the prefetch inside the K loop.  Round 4 lost 25 % of conv_igemm that way for a few hours (a runtime
``split`` flag in the source selection) before the per-kernel bench table showed it.  Compiles the units to assembly and reports, per
kernel with matrix instructions, the number of scratch_ instructions.
    python tools/audit_scratch.py            (tests/test_build_audits.py runs it)"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "remote-sensing-vision-language-diffusion-model_amd", "csrc")
UNITS = {"conv_igemm.hip": [], "conv_halo.hip": [], "gemm.hip": [], "split.hip": ["-fno-slp-vectorize"]}
# kernels allowed a few scratch instructions OUTSIDE their loops (none today); name fragment -> max count
ALLOW = {}


def audit(unit):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "unit.s")
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fno-gpu-rdc", *UNITS[unit],
               f"-I{SRC}", f"-I{os.path.join(ROOT, 'include')}", "-S", "--cuda-device-only", "-o", out, os.path.join(SRC, unit)]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        txt = open(out).read()
    parts = re.split(r"^(_Z[^\s:]+):[^\n]*$", txt, flags=re.M)
    res = {}
    for i in range(1, len(parts) - 1, 2):
        name, body = parts[i], parts[i + 1]
        if ".end_amdhsa_kernel" not in body:
            continue
        body = body.split(".end_amdhsa_kernel")[0]
        if "v_mfma" in body:
            res[name] = body.count("scratch_")
    return res


if __name__ == "__main__":
    rc = 0
    for unit in UNITS:
        res = audit(unit)
        bad = {k: v for k, v in res.items() if v > max([m for frag, m in ALLOW.items() if frag in k] or [0])}
        print(f"{unit}: {len(res)} MFMA kernels, {len(bad)} with scratch instructions")
        for k, v in bad.items():
            print(f"  {v:4d} scratch instructions in {k}")
        rc |= 1 if bad else 0
    sys.exit(rc)
