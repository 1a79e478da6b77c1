"""Config plumbing of the sgm plugin surface (reference: sgm/util.py:150-200): dotted ``target:``
strings are imported and called with ``params``.  Configs may be plain dicts or ``AttrDict``s
(``just_sampling`` assigns ``sampler_config.params.num_steps = ...``, models/SR_model.py:242-252,
so attribute access must work; omegaconf is not required)."""
import importlib
from inspect import isfunction

import torch


class AttrDict(dict):
    """dict with attribute access, recursively (stands in for omegaconf.DictConfig)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        for key, v in list(self.items()):
            self[key] = _wrap(v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = _wrap(v)

    def __deepcopy__(self, memo):
        import copy
        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _wrap(v):
    if isinstance(v, dict) and not isinstance(v, AttrDict):
        return AttrDict(v)
    if isinstance(v, list):
        return [_wrap(x) for x in v]
    return v


def exists(x):
    return x is not None


def default(val, d):
    if exists(val):
        return val
    return d() if isfunction(d) else d


def get_obj_from_str(string, reload=False, invalidate_cache=True):
    module, cls = string.rsplit(".", 1)
    if invalidate_cache:
        importlib.invalidate_caches()
    return getattr(importlib.import_module(module, package=None), cls)


def instantiate_from_config(config):
    if "target" not in config:
        if config in ("__is_first_stage__", "__is_unconditional__"):
            return None
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**config.get("params", dict()))


def append_zero(x):
    return torch.cat([x, x.new_zeros([1])])


def append_dims(x, target_dims):
    dims_to_append = target_dims - x.ndim
    if dims_to_append < 0:
        raise ValueError(f"input has {x.ndim} dims but target_dims is {target_dims}, which is less")
    return x[(...,) + (None,) * dims_to_append]


def disabled_train(self, mode=True):
    return self
