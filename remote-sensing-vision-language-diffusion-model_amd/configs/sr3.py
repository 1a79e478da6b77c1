"""Stage-1 option namespace (reference: configs/sr3.py:1-10): the attribute set ``utils.logger.parse`` reads."""
import os

_HERE = os.path.dirname(os.path.abspath(__file__))

# name -> default; logging / wandb switches are all off for inference
_DEFAULTS = dict(config=os.path.join(_HERE, "sr_sr3.json"), phase="val", gpu_ids="0", debug=False,
                 **{flag: False for flag in ("enable_wandb", "log_infer", "log_eval", "log_wandb_ckpt")})

SR3_Config = type("SR3_Config", (), dict(_DEFAULTS, __doc__="class-level defaults, instantiated or used as a namespace"))
