"""Import shim: `import fhip_amd as fhe` == the package in ./fully-homomorphic-image-processing_amd/."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("fully-homomorphic-image-processing_amd")
globals().update({k: getattr(_pkg, k) for k in _pkg.__all__})
package = _pkg
