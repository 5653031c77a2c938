"""diffroll_amd - MI355X-native (gfx950) sampling engine for DiffRoll.

Scope: the reverse-diffusion sampling hot path of the reference's ``ClassifierFreeDiffRoll``
(mel front-end -> 200 x [network evaluation(s) + posterior update]) as hand-written HIP kernels
behind a C-ABI (include/diffroll_amd.h), exposed through the reference's own Python surface.

    from diffroll_amd import ClassifierFreeDiffRoll
"""
from .model import ClassifierFreeDiffRoll, AttrDict      # noqa: F401
from .engine import Engine, EngineError                    # noqa: F401
from .diffusion import q_sample, extract_x0, linear_beta_schedule   # noqa: F401

__all__ = ["ClassifierFreeDiffRoll", "Engine", "EngineError", "AttrDict", "q_sample", "extract_x0",
           "linear_beta_schedule"]
