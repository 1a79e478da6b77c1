"""Device bookkeeping shared by the SR3 wrapper (reference: models/sr3_model/base_model.py)."""
import torch


class BaseModel:
    def __init__(self, opt):
        self.opt = opt
        self.device = torch.device("cuda" if opt["gpu_ids"] is not None else "cpu")
        self.begin_step = 0
        self.begin_epoch = 0

    def set_device(self, x):
        if isinstance(x, dict):
            for key, item in x.items():
                if item is not None and hasattr(item, "to"):
                    x[key] = item.to(self.device)
            return x
        if isinstance(x, list):
            return [None if item is None else item.to(self.device) for item in x]
        return x.to(self.device)

    def get_network_description(self, network):
        return str(network), sum(p.numel() for p in network.parameters())
