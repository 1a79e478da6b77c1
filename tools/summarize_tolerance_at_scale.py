"""gpurun_out/r06_tol_scale_*.jsonl (tools/tolerance_at_scale.py) -> the table committed as profiles/r06_tolerance_at_scale.txt
    python tools/summarize_tolerance_at_scale.py gpurun_out/r06_tol_scale_*.jsonl > profiles/r06_tolerance_at_scale.txt"""
import json
import sys

recs, heads = [], []
for path in sys.argv[1:]:
    for line in open(path):
        d = json.loads(line)
        (heads if "tool" in d else recs).append(d)
print("north_star's 1e-3 at the metric's shapes over many steps (tools/tolerance_at_scale.py, MI355X)")
print("tolerance composition (what bench.py's `value` is timed in) vs the fp32-operand kernel family on the device, same seeds, CPU-order noise,")
print("FULL networks; max / mean |delta| per pixel of the final image, and of the sampler state after selected steps.")
if heads:
    print("composition:", json.dumps(heads[-1]["tolerance_composition"]))
print()
s1 = sorted((r for r in recs if r["stage"] == 1), key=lambda r: (r["side"], r["steps"]))
print("Stage 1 (SR3, compute dtype w2 = fp16 tensors x fp16 weight pairs): x_t in pixel space")
print(f"{'side':>6} {'attn tokens':>11} {'steps':>5} | {'final max':>10} {'final mean':>10} | per-step max at steps 1, 5, 10, 25, 50 | fp32 s / w2 s")
for r in s1:
    ps = r["per_step_max"]
    pick = [ps[i - 1] for i in (1, 5, 10, 25, 50) if i <= len(ps)]
    print(f"{r['side']:>6} {(r['side'] // 8) ** 2:>11} {r['steps']:>5} | {r['pixel_max']:10.3e} {r['pixel_mean']:10.3e} | "
          + " ".join(f"{v:.2e}" for v in pick) + f" | {r['seconds_fp32_family']} / {r['seconds_w2']}   inside 1e-3: {r['inside_1e-3']}")
print()
s2 = sorted((r for r in recs if r["stage"] == 2), key=lambda r: (r["latent"], r["img_threshold"], r["steps"]))
print("Stage 2 (ControlNet + UNet, CFG pair, EDM sampler, tiled VAE from latent 256): decoded image per pixel; sampler state = latent z (range ~130)")
print(f"{'latent':>6} {'tokens@lvl0':>11} {'steps':>5} {'cache':>5} | {'pixel max':>10} {'pixel mean':>10} | latent mean |d| at steps 1, 10, 25, 50 | decisions equal (hits) | fp32 s / tol s")
for r in s2:
    ps = r["per_step_latent_mean"]
    pick = [ps[i - 1] for i in (1, 10, 25, 50) if i <= len(ps)]
    print(f"{r['latent']:>6} {r['tokens_level0']:>11} {r['steps']:>5} {r['img_threshold']:>5} | {r['pixel_max']:10.3e} {r['pixel_mean']:10.3e} | "
          + " ".join(f"{v:.2e}" for v in pick) + f" | {r['cache_decisions_equal']} ({r['cache_hits']}/{r['decisions']}) | "
          f"{r['seconds_fp32_family']} / {r['seconds_tolerance']}   inside 1e-3: {r['inside_1e-3']}")
