"""SR3 option parser (reference: utils/logger.py:21-93): '//'-commented JSON -> dict.  Unlike the reference
it does NOT export CUDA_VISIBLE_DEVICES (:53): device selection is the launcher's job (one process per GPU)."""
import json
import os
from collections import OrderedDict


def parse(args):
    json_str = ""
    with open(args.config, "r") as f:
        for line in f:
            json_str += line.split("//")[0] + "\n"
    opt = json.loads(json_str, object_pairs_hook=OrderedDict)
    if getattr(args, "debug", False):
        opt["name"] = "debug_{}".format(opt["name"])
    opt["phase"] = args.phase
    if args.gpu_ids is not None:
        opt["gpu_ids"] = [int(i) for i in str(args.gpu_ids).split(",")]
    opt["distributed"] = len(opt.get("gpu_ids") or []) > 1
    if "debug" in opt["name"]:
        for ph in ("train", "val"):
            opt["model"]["beta_schedule"][ph]["n_timestep"] = 10
    opt["enable_wandb"] = getattr(args, "enable_wandb", False)
    rs = opt.get("path", {}).get("resume_state")
    if rs is not None and not os.path.exists(f"{rs}_gen.pth"):
        opt["path"]["resume_state"] = None      # no checkpoint offline: random init
    return opt
