"""Stage-1 option namespace (reference: configs/sr3.py:1-10)."""
import os


class SR3_Config:
    config = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sr_sr3.json")
    phase = "val"
    gpu_ids = "0"
    debug = False
    enable_wandb = False
    log_infer = False
    log_eval = False
    log_wandb_ckpt = False
