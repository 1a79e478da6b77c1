"""Run a chosen part of the headline workload N times (for rocprofv3 --kernel-trace --stats: how many __amd_rocclr_copyBuffer
dispatches does ONE Stage-1 iteration / ONE Stage-2 iteration / ONE tiled VAE pass make?).
    PART=s1|s2|vae N=1|2 python tools/count_copies.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from rsvld_amd import measure

part, n = os.environ.get("PART", "s2"), int(os.environ.get("N", 1))
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
torch.manual_seed(0)
if part == "s1":
    net, _ = bench.build_stage1(50)
    net.use_graph = False
    cond = bench.stage1_input([0], 512, 8).to(dev)
    with measure.hooks(net, max_steps=n):
        net.super_resolution(cond, continous=True)
else:
    m = bench.build_stage2(dev, True)
    if part == "vae":
        x = bench.synthetic_image((1, 3, 4096, 4096), seed=1, smooth=4).to(dev)
        for _ in range(n):
            m.encode_first_stage_with_denoise(x, use_sample=False)
    else:
        lq = bench.synthetic_image((1, 3, 4096, 4096), seed=1, smooth=4).to(dev)
        z = torch.randn(1, 4, 512, 512, device=dev)
        c, uc = m.prepare_condition(z, [""], "", "", 1)
        from rsvld_amd.sgm.modules.diffusionmodules.guiders import LinearCFG
        guider = LinearCFG(scale=4.0, scale_min=7.5)
        sigma = torch.tensor([7.3])
        for _ in range(n):
            inp = guider.prepare_inputs(z, sigma, c, uc)
            guider(m.denoiser(m.model, *inp, control_scale=1.0, fbcache_mode="none", partial_info=None), sigma)
torch.cuda.synchronize()
print("done", part, n)
