"""gl_gym_amd -- MI355X-native hot path of GreenLight-Gym: the per-timestep GreenLight ODE integration behind
TomatoEnv.step(), as hand-written gfx950 kernels behind a C ABI (include/glgym.h), plus the host-side mirror of
the reference's env interface.  Importing the package does not need a GPU; creating a model/env does."""
from ._lib import GlgymError, GlgymOdeError, INFO_KEYS, METRIC_KEYS, LIB_PATH  # noqa: F401
from .parameters import init_default_params  # noqa: F401
from .greenlight_model import GreenLight  # noqa: F401

__all__ = ["GreenLight", "GlgymError", "GlgymOdeError", "init_default_params", "INFO_KEYS", "METRIC_KEYS", "LIB_PATH"]
