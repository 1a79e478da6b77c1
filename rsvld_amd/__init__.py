"""Import alias for the package directory ``remote-sensing-vision-language-diffusion-model_amd/``.

The directory name required by the repo layout contains hyphens, which Python cannot import;
this package simply points its ``__path__`` there, so ``import rsvld_amd.sr3_model`` resolves to
``remote-sensing-vision-language-diffusion-model_amd/sr3_model``.
"""
import os as _os

_here = _os.path.dirname(_os.path.abspath(__file__))
_real = _os.path.join(_os.path.dirname(_here), "remote-sensing-vision-language-diffusion-model_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _here, _real, _f
