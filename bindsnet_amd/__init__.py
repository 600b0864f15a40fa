"""bindsnet_amd: MI355X-native implementation of BindsNET's Network.run() hot path.

The compute path is libsnnhip.so (hand-written gfx950 HIP kernels behind the C ABI in
include/snnhip.h); this package is the host side that mirrors BindsNET's Python API for that
path.  There is no CPU or PyTorch fallback.
"""
__version__ = "0.1.0"
