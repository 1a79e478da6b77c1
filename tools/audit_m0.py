"""Audit for the inline-asm LDS-DMA statements that declare M0 clobbered instead of saving / restoring it (-DA5B_M0_CLOBBER=1):
hipcc reserves M0 and warns that clobbering it "may lead to undefined behaviour", so the build is only acceptable while the
COMPILER itself never reads or writes M0 in that translation unit.  Compiles attention.hip to assembly and fails if M0 appears
anywhere outside an ;;#ASMSTART .. ;;#ASMEND block.   python tools/audit_m0.py [extra hipcc flags]"""
import os, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "remote-sensing-vision-language-diffusion-model_amd", "csrc")


def audit(extra=()):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "attention.s")
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fno-gpu-rdc", "-fno-slp-vectorize",
               "-DA5B_M0_CLOBBER=1", f"-I{SRC}", f"-I{os.path.join(ROOT, 'include')}", "-S", "--cuda-device-only", "-o", out,
               os.path.join(SRC, "attention.hip"), *extra]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        bad, inasm, dma = [], False, 0
        for n, line in enumerate(open(out), 1):
            if "#ASMSTART" in line:
                inasm = True
            elif "#ASMEND" in line:
                inasm = False
            else:
                code = line.split(";")[0]
                if inasm and "global_load_lds" in code:
                    dma += 1
                if not inasm and "m0" in code.replace("vm0", "") and not code.strip().startswith("."):
                    bad.append((n, line.strip()))
        return bad, dma


if __name__ == "__main__":
    bad, dma = audit(sys.argv[1:])
    print(f"{dma} LDS-DMA instructions inside asm blocks; compiler uses of m0 outside asm blocks: {len(bad)}")
    for n, l in bad[:20]:
        print(f"  line {n}: {l}")
    sys.exit(1 if bad else 0)
