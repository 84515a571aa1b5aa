"""sdslam_amd -- MI355X (gfx950) implementation of SD-SLAM's per-frame tracking hot path.

The product is libsdslam_hip.so (hand-written HIP kernels behind the C ABI of
include/sdslam_hip.h).  This package is only the loader plus thin ctypes mirrors of the
reference's class surface, used by tests/ and bench.py.  There is NO CPU fallback: importing
works anywhere (so the symbol table can be checked), but every compute entry point fails
loudly when the library or a GPU is missing.
"""
from .capi import (KP_DTYPE, SdError, lib, lib_path, ORBextractor, Tracker, DeviceBuffer, device_count,  # noqa: F401
                   hamming, plan_info, set_option, get_option, option_names, options)

__all__ = ["KP_DTYPE", "SdError", "lib", "lib_path", "ORBextractor", "Tracker", "DeviceBuffer", "device_count", "hamming", "plan_info",
           "set_option", "get_option", "option_names", "options"]
