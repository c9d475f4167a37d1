"""point_sam_amd: MI355X-native (gfx950) inference hot path for Point-SAM.

The product path is HIP-only: importing ``point_sam_amd.ops`` (or anything built on it) raises if the
compiled library ``csrc/libpointsam_hip.so`` is missing.
"""
from .config import CONFIGS, ModelConfig, ViTConfig, get_config  # noqa: F401

__version__ = "0.1.0"
