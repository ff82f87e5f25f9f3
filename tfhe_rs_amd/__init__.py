"""tfhe_rs_amd — MI355X-native TFHE programmable-bootstrapping backend (host-side package).

The product is libtfhe_hip_backend.so (tfhe_rs_amd/csrc, C ABI in include/tfhe_hip_backend.h);
this package is the thin host side: `ffi` (raw binding) and `core_crypto_gpu` (mirror of
tfhe::core_crypto::gpu), `integer_gpu` (mirror of the wired part of tfhe::integer::gpu).
"""
from . import ffi  # noqa: F401
from . import core_crypto_gpu  # noqa: F401
from . import multi_gpu  # noqa: F401

__all__ = ["ffi", "core_crypto_gpu", "multi_gpu"]
