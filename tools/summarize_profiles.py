"""Turn raw rocprofv3 output under gpurun_out/ into the summaries committed under profiles/.

    python tools/summarize_profiles.py --stats gpurun_out/prof_stats3 --out profiles/r01_c2_kernel_stats.csv
    python tools/summarize_profiles.py --pmc gpurun_out/pmc_fetch3 gpurun_out/pmc_write3 gpurun_out/pmc_mfma3 \
        --out profiles/r01_c2_pmc_traffic.json --command "..."

PMC handling follows MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE come from separate
passes, are in KiB, and on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes -> bytes = (2*FETCH + WRITE) * 1024.
"""
import argparse
import collections
import csv
import glob
import json
import os
import re


SEG_SFX = {"1": "", "2": "_w2", "3": "_split", "4": "_w1", "5": "_q8"}   # the K-segment template parameter of the matrix kernels (pairs / triples / round 6: fp16 + e4m3 cross terms)


def short_name(n):
    m = re.search(r"conv_halo(?:32)?_kernelI(\w+?)Li(\d+)ELi\d+ELi\d+ELi(\d)E", n)   # <T, BN, ., ., SEG>
    if m:
        return "conv_halo_%s%s" % (m.group(2), SEG_SFX.get(m.group(3), ""))
    m = re.search(r"conv_halo(?:32)?_kernelI(\w+?)Li(\d+)E", n)   # conv_halo32 is the Cout > 64 variant: same bench label (rounds 1-4 names)
    if m:
        return "conv_halo_%s" % m.group(2)
    m = re.search(r"conv_igemm_kernelI\w+?Li(\d+)ELi(\d+)ELi\d+ELi\d+ELb[01]ELi\d+ELi\d+ELi(\d)E", n)   # <T, BM, BN, ., ., GLDS, STAGES, KS, SEG>
    if m:
        return "conv_igemm_%sx%s%s" % (m.group(1), m.group(2), SEG_SFX.get(m.group(3), ""))
    m = re.search(r"conv_igemm_kernelI\w+?Li(\d+)ELi(\d+)E", n)
    if m:
        return "conv_igemm_%sx%s" % (m.group(1), m.group(2))
    m = re.search(r"gemm256_kernelI\w+?Li(\d)ELb[01]E", n)   # <T, SEG, PERSIST, PV, WIDE>
    if m:
        return "gemm256" + SEG_SFX.get(m.group(1), "")
    m = re.search(r"attn_d(\d+)b?_kernel", n)
    if m:
        return "attn_d%s" % m.group(1)
    m = re.search(r"(?:N_1\d+|::)([a-z0-9_]+)_kernel", n)
    if m:
        return m.group(1)
    if "rocclr" in n:
        return n.strip('"')
    return "torch:" + re.sub(r"[^A-Za-z_]+", "_", n)[:48]


def find(dirname, suffix):
    hits = glob.glob(os.path.join(dirname, "**", "*" + suffix), recursive=True)
    if not hits:
        raise SystemExit("no *%s under %s" % (suffix, dirname))
    return hits[0]


def stats(dirname, out):
    rows = list(csv.DictReader(open(find(dirname, "_kernel_stats.csv"))))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "percent", "min_us", "max_us", "full_name"])
        for r in rows:
            w.writerow([short_name(r["Name"]), r["Calls"], "%.3f" % (int(r["TotalDurationNs"]) / 1e6),
                        "%.2f" % (float(r["AverageNs"]) / 1e3), r["Percentage"], "%.2f" % (int(r["MinNs"]) / 1e3),
                        "%.2f" % (int(r["MaxNs"]) / 1e3), r["Name"][:160]])
    print("wrote", out, len(rows), "kernels")


def pmc(dirs, out, command):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(int))
    for d in dirs:
        for r in csv.DictReader(open(find(d, "_counter_collection.csv"))):
            k = short_name(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k][r["Counter_Name"]] += 1
    kernels = {}
    for k, c in acc.items():
        if k.startswith("torch:") or "rocclr" in k:
            continue
        e = {}
        n = max(launches[k].values())
        e["launches"] = n
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            f = c["FETCH_SIZE"] / launches[k]["FETCH_SIZE"]
            w = c["WRITE_SIZE"] / launches[k]["WRITE_SIZE"]
            e["fetch_size_kb_per_launch"] = f
            e["write_size_kb_per_launch"] = w
            e["hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0
        rest = {m: v for m, v in c.items() if m not in ("FETCH_SIZE", "WRITE_SIZE")}
        if rest:
            e["pmc"] = rest
            if rest.get("GRBM_GUI_ACTIVE"):
                e["mfma_busy_frac_est"] = rest.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (rest["GRBM_GUI_ACTIVE"] / 8 * 1024)
        kernels[k] = e
    doc = {"command": command,
           "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 128-B requests at 64 B; "
                         "MI355X_MICROARCH.md HBM section); mfma_busy_frac_est = SQ_VALU_MFMA_BUSY_CYCLES / "
                         "(GRBM_GUI_ACTIVE/8 * 1024 SIMDs)",
           "kernels": kernels}
    json.dump(doc, open(out, "w"), indent=1)
    print("wrote", out, sorted(kernels))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats")
    ap.add_argument("--pmc", nargs="*")
    ap.add_argument("--out", required=True)
    ap.add_argument("--command", default="")
    a = ap.parse_args()
    if a.stats:
        stats(a.stats, a.out)
    else:
        pmc(a.pmc, a.out, a.command)
