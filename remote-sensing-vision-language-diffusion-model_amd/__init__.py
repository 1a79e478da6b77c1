"""MI355X-native two-stage diffusion super-resolution sampler (hot path only).

Import name: ``rsvld_amd`` (the alias package at the repo root extends its ``__path__`` to this
directory, whose on-disk name carries hyphens and so cannot be imported directly).

Sub-packages mirror the reference's module tree for the hot path:
  rsvld_amd.sr3_model          <- models/sr3_model        (Stage-1 SR3 DDPM)
  rsvld_amd.sgm / .models      <- sgm/**, models/**       (Stage-2 sampler, denoiser, networks)
  rsvld_amd.utils              <- utils/{tilevae,colorfix,tensor2img}.py
Kernels live in ``csrc/`` behind the C ABI of ``include/rsvld_hip.h`` (librsvld_hip.so).
"""
__version__ = "0.1.0"
