"""Optional native fast path (taseg_amd/_fast_block.so, built by taseg_amd.csrc.fastpath.build): the conv -> BatchNorm
[-> residual] [-> ReLU] block as a C++ autograd node.  It issues exactly the backend calls of the Python Function
(`functional._ConvBlock`); when the binding has not been built, or TASEG_FAST_BLOCK=0, the Python node serves the
block - both run the same HIP kernels of libtaseg_hip.so."""
import importlib.util
import os

from . import _lib as L

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_fast_block.so")
_mod = None
_tried = False


def module():
    """the loaded binding, or None"""
    global _mod, _tried
    if _tried:
        return _mod
    _tried = True
    if os.environ.get("TASEG_FAST_BLOCK", "1") == "0" or not os.path.exists(_PATH):
        return None
    import torch  # noqa: F401  (libtorch must be loaded before the extension)
    L.load()          # the kernels themselves are not optional: a missing libtaseg_hip.so raises here
    try:
        spec = importlib.util.spec_from_file_location("_fast_block", _PATH)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.load_backend(L.LIB_PATH)
    except Exception as e:  # noqa: BLE001 - e.g. built against another PyTorch: the Python nodes serve the blocks
        import warnings
        warnings.warn(f"taseg_amd: native fast path not usable ({e}); using the Python autograd nodes")
        return None
    _mod = mod
    return _mod
