"""fly_bproject_amd — MI355X-native drop-in for the hot path of petim0/fly_bProject.

`Fly` (fly.py) and `PPO` / `Net` (ppo.py) mirror the reference's classes of the same names; the
math runs in hand-written gfx950 kernels behind the C ABI declared in include/flyhip.h
(libflyhip.so).  There is no CPU fallback: without the built library, or without a GPU,
constructing `Fly` raises.
"""
from .params import FlyParams, default_params  # noqa: F401

__all__ = ["FlyParams", "default_params", "Fly", "PPO", "Net"]


def __getattr__(name):
    if name == "Fly":
        from .fly import Fly
        return Fly
    if name in ("PPO", "Net"):
        from . import ppo
        return getattr(ppo, name)
    raise AttributeError(name)
