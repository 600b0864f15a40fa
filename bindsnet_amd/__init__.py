"""bindsnet_amd: MI355X-native implementation of BindsNET's Network.run() hot path.

The compute path is libsnnhip.so (hand-written gfx950 HIP kernels behind the C ABI in
include/snnhip.h); this package is the host side that mirrors BindsNET's Python API for that
path.  There is no CPU or PyTorch fallback.
"""
from pathlib import Path

__version__ = "0.1.0"
ROOT_DIR = Path(__file__).parents[0].parents[0]      # the checkout's root, like bindsnet.ROOT_DIR (bindsnet/__init__.py:18)
