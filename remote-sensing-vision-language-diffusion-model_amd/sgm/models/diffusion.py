"""``DiffusionEngine`` — the construction half of sgm/models/diffusion.py:20-82,104-109: builds network
(+wrapper), denoiser, sampler, conditioner and first stage from ``target:`` configs.  A plain
nn.Module (the reference's LightningModule adds only training hooks, which are out of scope)."""
import torch
from torch import nn

from ..modules import UNCONDITIONAL_CONFIG
from ..modules.diffusionmodules.wrappers import OPENAIUNETWRAPPER
from ..util import default, disabled_train, get_obj_from_str, instantiate_from_config


class DiffusionEngine(nn.Module):
    def __init__(self, network_config, denoiser_config, first_stage_config, conditioner_config=None,
                 sampler_config=None, optimizer_config=None, scheduler_config=None, loss_fn_config=None,
                 network_wrapper=None, ckpt_path=None, use_ema=False, ema_decay_rate=0.9999, scale_factor=1.0,
                 disable_first_stage_autocast=False, input_key="jpg", log_keys=None, no_cond_log=False,
                 compile_model=False):
        super().__init__()
        self.log_keys, self.input_key = log_keys, input_key
        model = instantiate_from_config(network_config)
        self.model = get_obj_from_str(default(network_wrapper, OPENAIUNETWRAPPER))(model, compile_model=compile_model)
        self.denoiser = instantiate_from_config(denoiser_config)
        self.sampler = instantiate_from_config(sampler_config) if sampler_config is not None else None
        self.conditioner = instantiate_from_config(default(conditioner_config, UNCONDITIONAL_CONFIG))
        self._init_first_stage(first_stage_config)
        self.scale_factor = scale_factor
        self.disable_first_stage_autocast = disable_first_stage_autocast
        self.no_cond_log = no_cond_log
        if use_ema or loss_fn_config is not None:
            raise NotImplementedError("training-time members (EMA, loss) are outside the inference hot path")
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path)

    @property
    def device(self):
        return next(self.parameters()).device

    def init_from_ckpt(self, path):
        if path.endswith("ckpt"):
            sd = torch.load(path, map_location="cpu")["state_dict"]
        elif path.endswith("safetensors"):
            from safetensors.torch import load_file
            sd = load_file(path)
        else:
            raise NotImplementedError
        missing, unexpected = self.load_state_dict(sd, strict=False)
        print(f"Restored from {path} with {len(missing)} missing and {len(unexpected)} unexpected keys")

    def _init_first_stage(self, config):
        model = instantiate_from_config(config).eval()
        model.train = disabled_train.__get__(model)
        for p in model.parameters():
            p.requires_grad = False
        self.first_stage_model = model
