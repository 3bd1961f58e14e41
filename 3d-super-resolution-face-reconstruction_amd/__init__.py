"""MI355X-native SR3 iterative-refinement sampler (hand-written HIP for gfx950 behind the
reference's define_G / GaussianDiffusion / UNet API). See DESIGN.md and INTEGRATION.md."""
from .graph import UNetConfig, param_specs, count_params, flops_per_image  # noqa: F401
from .schedule import make_beta_schedule, schedule_buffers  # noqa: F401
from ._lib import Sr3Error, LIB_PATH  # noqa: F401
from .engine import Engine  # noqa: F401
from .unet import UNet  # noqa: F401
from .diffusion import GaussianDiffusion  # noqa: F401
from .networks import define_G  # noqa: F401

__all__ = ["UNetConfig", "param_specs", "count_params", "flops_per_image", "make_beta_schedule",
           "schedule_buffers", "Sr3Error", "Engine", "UNet", "GaussianDiffusion", "define_G"]
