"""MI355X-native BFV ciphertext arithmetic for the homomorphic image circuits of
wfus/Fully-Homomorphic-Image-Processing.  See DESIGN.md.

The package directory name carries hyphens (it mirrors the reference's name), so import it with
    import importlib; fhe = importlib.import_module("fully-homomorphic-image-processing_amd")
or through the `fhip_amd` shim at the repository root.
"""
from . import _lib, circuits, client, parallel, server
from ._lib import FheError, LIB_PATH, HEADER_PATH, HEADER_PATHS
from .keys import Decryptor, DeviceEncryptor, Encryptor, KeyGenerator
from .evaluator import (PRESETS, SEED, YQT, DctPlan, Evaluator, FractionalEncoder, PreparedPlain, SEALContext,
                        to_device, to_host)

__all__ = ["KeyGenerator", "Encryptor", "DeviceEncryptor", "Decryptor", "FheError", "LIB_PATH", "HEADER_PATH", "HEADER_PATHS", "PRESETS", "SEED", "YQT", "DctPlan", "Evaluator",
           "FractionalEncoder", "PreparedPlain", "SEALContext", "to_device", "to_host", "_lib", "parallel", "circuits", "server", "client"]
