"""Import-only stand-ins for third-party wheels the REFERENCE imports but this container lacks
(authoring-container tooling for the golden generators; never used by the product or on the GPU box).
No reference file is modified; the shims only make `import` succeed, none of them computes anything
the goldens depend on (the xformers stand-in is plain single-head SDPA, needed because
utils/tilevae.py:364 always takes the xformers branch)."""
import importlib.machinery
import sys
import types

sys.dont_write_bytecode = True


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install(xformers=True):
    """``xformers=False``: no xformers stand-in at all -- ``import xformers`` fails as on a box without the wheel and the reference takes its
    own torch-SDPA branch (sgm/modules/attention.py:397-402, SR_modules.py:121); tests/golden/check_s2_no_xformers.py uses it to show that
    the Stage-2 goldens do not depend on the stand-in."""
    import torch
    import torch.nn as nn
    import transformers  # noqa: F401  (must be imported BEFORE the torchvision stub)

    class LightningModule(nn.Module):
        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")

        def log(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass

    _mod("pytorch_lightning", LightningModule=LightningModule)

    class AttrDict(dict):
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

    class ListConfig(list):
        pass

    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            return AttrDict(v)
        return v

    class OmegaConf:
        @staticmethod
        def create(d):
            return AttrDict(d)

        @staticmethod
        def load(path):
            import yaml
            return AttrDict(yaml.safe_load(open(path)))

    oc = _mod("omegaconf", OmegaConf=OmegaConf, ListConfig=ListConfig, DictConfig=AttrDict)
    _mod("omegaconf.listconfig", ListConfig=ListConfig)
    oc.AttrDict = AttrDict
    kd = _mod("k_diffusion")
    kd.sampling = _mod("k_diffusion.sampling", BrownianTreeNoiseSampler=object, get_sigmas_karras=None)
    for name in ("kornia", "open_clip", "cv2", "lpips", "torchmetrics", "peft"):
        _mod(name)
    sys.modules["lpips"].LPIPS = lambda *a, **k: nn.Identity()
    sys.modules["torchmetrics"].functional = _mod("torchmetrics.functional")
    sys.modules["peft"].PeftModel = object
    _mod("llava")
    _mod("llava.mm_utils", tokenizer_image_token=None, process_images=None)
    _mod("llava.model")
    _mod("llava.model.builder", load_pretrained_model=None)
    _mod("llava.constants", DEFAULT_IMAGE_TOKEN="<image>", IMAGE_TOKEN_INDEX=-200)
    _mod("llava.conversation", conv_templates={})
    tv = _mod("torchvision")
    tv.transforms = _mod("torchvision.transforms", ToPILImage=object, ToTensor=object, Compose=object, Normalize=object)
    tv.transforms.functional = _mod("torchvision.transforms.functional")
    tv.models = _mod("torchvision.models")
    tv.utils = _mod("torchvision.utils", make_grid=None)

    def mea(q, k, v, attn_bias=None, op=None):  # [B,N,C] single head
        return torch.nn.functional.scaled_dot_product_attention(q.unsqueeze(1), k.unsqueeze(1), v.unsqueeze(1)).squeeze(1)

    if xformers:
        xf = _mod("xformers")
        xf.ops = _mod("xformers.ops", memory_efficient_attention=mea)
    df = _mod("diffusers")
    df.utils = _mod("diffusers.utils")
    df.utils.import_utils = _mod("diffusers.utils.import_utils", is_xformers_available=lambda: True)
    if "/root/reference" not in sys.path:
        sys.path.insert(1, "/root/reference")
    return AttrDict
