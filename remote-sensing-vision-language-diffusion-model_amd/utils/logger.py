"""SR3 option parser (reference: utils/logger.py:21-93): '//'-commented JSON -> dict.  Unlike the reference
it does NOT export CUDA_VISIBLE_DEVICES (:53): device selection is the launcher's job (one process per GPU)."""
import json
import os
from collections import OrderedDict


def parse(args, allow_random_init=False):
    """``allow_random_init`` (bench / tests): a configured ``resume_state`` whose ``<resume_state>_gen.pth`` does not exist
    becomes None (seeded random weights).  Otherwise it raises FileNotFoundError -- the rule of ``create_SR_model``: a wrong
    checkpoint path must not silently produce noise images (reference: sr3_model/model.py:149-170 loads unconditionally)."""
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
        if not allow_random_init:
            raise FileNotFoundError(f"Stage-1 checkpoint {rs}_gen.pth (path.resume_state of {args.config}) does not exist; "
                                    "pass allow_random_init=True (PipelineConfig.allow_random_init) to run on seeded random weights")
        opt["path"]["resume_state"] = None
    return opt
