"""One rank of the data-parallel two-stage pipeline test (launched by tests/test_gpu_dp.py with RANK / WORLD_SIZE /
MASTER_* set; several ranks may share GPU 0 through RSVLD_DEVICE_OVERRIDE + the gloo backend).

Image i (per-image seeds): Stage 1 (SR3, 32 -> 64, 3 steps) -> 8-bit hand-off -> Stage 2 (reduced-depth networks, 64^2,
3 EDM steps, feature cache 0.3, Wavelet) -> uint8.  Image i -> rank i mod N, ONE all-gather (rsvld_amd.parallel.run_sharded).
Rank 0 writes all images, in image order, to argv[2]."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import s2_common as S
from oracle import seeded
from rsvld_amd import parallel
from rsvld_amd.sgm.util import instantiate_from_config
from rsvld_amd.sr3_model.sr3_modules.diffusion import GaussianDiffusion
from rsvld_amd.sr3_model.sr3_modules.unet import UNet

n_images, out_path = int(sys.argv[1]), sys.argv[2]
rank, world, local = parallel.init_from_env()
local = int(os.environ.get("RSVLD_DEVICE_OVERRIDE", local))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)

unet = UNet(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32, channel_mults=[1, 2, 4, 8, 8], attn_res=[28],
            res_blocks=1, dropout=0.2, image_size=224)
net = GaussianDiffusion(unet, image_size=224, channels=3, conditional=True)
seeded.seed_module(net, 1234)
net.to(dev).eval()
net.set_new_noise_schedule(dict(schedule="linear", n_timestep=3, linear_start=1e-6, linear_end=1e-2), dev)
m = instantiate_from_config({"target": "rsvld_amd.models.SR_model.SR_backbone", "params": S.product_params()})
seeded.seed_module(m, S.WEIGHT_SEED)
m.to(dev).eval()


def process(i):
    torch.manual_seed(42 + i)                                    # CPU and device generators: the image's own RNG stream
    lr = seeded.synthetic_image((1, 3, 32, 32), seed=1234 + i, smooth=3)
    cond = torch.nn.functional.interpolate(lr, scale_factor=2, mode="bicubic", align_corners=False).clamp(-1, 1).to(dev)
    sr = net.super_resolution(cond, continous=True)[-1:]
    lq = parallel.to_uint8(sr).float() / 127.5 - 1.0
    out = m.just_sampling(lq, [""], p_p="", n_p="", img_threshold=0.3, dec_img=1.0, num_steps=3, restoration_scale=-1,
                          s_churn=5, s_noise=1.003, cfg_scale=7.5, control_scale=1.0, color_fix_type="Wavelet",
                          use_linear_CFG=True, cfg_scale_start=4.0)
    return parallel.to_uint8(out)[0]


imgs = parallel.run_sharded(process, n_images, rank, world)
if rank == 0:
    np.save(out_path, imgs.cpu().numpy())
if world > 1:
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
print(f"DP_WORKER_OK rank {rank}/{world}", flush=True)
