"""gbrl_amd -- MI355X-native implementation of NVlabs/gbrl's tree-fit / ensemble-predict hot path.

The product is ``libgbrl_hip.so`` (hand-written HIP for gfx950 behind the C ABI in ``include/gbrl_hip.h``) plus the
Python extension ``gbrl_cpp`` (class ``GBRL``), which mirrors the reference's ``gbrl_cpp.GBRL`` operator interface
(gbrl/src/cpp/binding.cpp:421) for that path.  There is no CPU fallback anywhere in this package: importing it needs the
built extension, and ``GBRL.step`` / ``GBRL.predict`` raise ``RuntimeError`` when no HIP device is usable.
"""
from __future__ import annotations

import importlib.util
import os
import sysconfig

_HERE = os.path.dirname(os.path.abspath(__file__))
_EXT = os.path.join(_HERE, "gbrl_cpp" + sysconfig.get_config_var("EXT_SUFFIX"))
LIB_PATH = os.path.join(_HERE, "libgbrl_hip.so")

if not (os.path.exists(_EXT) and os.path.exists(LIB_PATH)):
    raise ImportError(
        "gbrl_amd: the HIP extension is not built (expected %s and %s). Run `python gbrl_amd/build.py` "
        "(or __graft_entry__.build()). There is no pure-Python / CPU fallback." % (LIB_PATH, _EXT))

# One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64; if this library were loaded first, the system
# copy under /opt/rocm would be bound instead and torch would then fail to see the GPU.  Importing torch first (when it
# is installed) makes both share torch's runtime.  torch is plumbing here (device tensors, DLPack, torch.distributed).
if importlib.util.find_spec("torch") is not None:
    import torch  # noqa: F401

_spec = importlib.util.spec_from_file_location("gbrl_cpp", _EXT)
gbrl_cpp = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gbrl_cpp)

GBRL = gbrl_cpp.GBRL          # the reference exposes this as gbrl.GBRL_CPP (gbrl/__init__.py:115-118)
GBRL_CPP = gbrl_cpp.GBRL
cuda_available = gbrl_cpp.GBRL.cuda_available

__all__ = ["GBRL", "GBRL_CPP", "gbrl_cpp", "cuda_available", "LIB_PATH"]
