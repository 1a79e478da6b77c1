"""``DDPM``: the Stage-1 model wrapper ``infer.py`` talks to.

Reference: models/sr3_model/model.py — ``feed_data`` (:48), ``test(continous)`` (:63-72), ``.SR``,
``set_new_noise_schedule`` (:91-98), ``load_network`` (:149-170, file ``<resume_state>_gen.pth``).
Training members (optimizer, save_network, optimize_parameters) are out of scope.
"""
import logging
from collections import OrderedDict

import torch

from . import networks
from .base_model import BaseModel

logger = logging.getLogger("base")


class DDPM(BaseModel):
    def __init__(self, opt):
        super().__init__(opt)
        self.netG = self.set_device(networks.define_G(opt))
        self.schedule_phase = None
        self.netG.set_loss(self.device)
        self.set_new_noise_schedule(opt["model"]["beta_schedule"]["train"], schedule_phase="train")
        if opt.get("phase") == "train":
            raise NotImplementedError("training is outside the MI355X inference hot path")
        self.load_network()
        self.netG.eval()

    def feed_data(self, data):
        self.data = self.set_device(data)

    @torch.no_grad()
    def test(self, continous=False):
        self.netG.eval()
        self.SR = self.netG.super_resolution(self.data["SR"], continous)

    @torch.no_grad()
    def sample(self, batch_size=1, continous=False):
        self.netG.eval()
        self.SR = self.netG.sample(batch_size, continous)

    def set_new_noise_schedule(self, schedule_opt, schedule_phase="train"):
        if self.schedule_phase is None or self.schedule_phase != schedule_phase:
            self.schedule_phase = schedule_phase
            self.netG.set_new_noise_schedule(schedule_opt, self.device)

    def get_current_visuals(self, need_LR=True, sample=False):
        out = OrderedDict()
        if sample:
            out["SAM"] = self.SR.detach().float().cpu()
        else:
            out["SR"] = self.SR.detach().float().cpu()
            out["INF"] = self.data["SR"].detach().float().cpu()
            if "HR" in self.data:
                out["HR"] = self.data["HR"].detach().float().cpu()
            out["LR"] = self.data["LR"].detach().float().cpu() if need_LR and "LR" in self.data else out["INF"]
        return out

    def load_network(self):
        load_path = self.opt["path"]["resume_state"]
        if load_path is None:
            return
        gen_path = f"{load_path}_gen.pth"
        logger.info("Loading pretrained model for G [%s] ...", gen_path)
        state = torch.load(gen_path, map_location="cpu")
        self.netG.load_state_dict(state, strict=not self.opt["model"]["finetune_norm"])
        self.netG.denoise_fn.invalidate_packed()
