"""taseg_amd - MI355X-native (gfx950, HIP) implementation of the TASeg / OpenPCSeg hot path:
voxelisation -> sparse-conv MinkUNet backbone -> multi-scan temporal aggregation.

  taseg_amd.backend       torch-facing wrappers of the C ABI (include/taseg_hip.h)
  taseg_amd.torchsparse   torchsparse v1.4.0-compatible operator API on that backend
  taseg_amd.pcseg         pcseg.model-compatible segmentors (MinkUNet, MinkUNetMs, build_network)
  taseg_amd.data          synthetic SemanticKITTI-shaped scans + the device data stage
"""
import sys

__version__ = "0.1.0"


def install_as_dropin():
    """Register the compatible packages under the reference's import names so unmodified
    OpenPCSeg code (`import torchsparse`, `from pcseg.model import build_network`) binds to them."""
    from . import torchsparse as ts
    from . import pcseg as pc
    names = {"torchsparse": ts, "torchsparse.nn": ts.nn, "torchsparse.nn.functional": ts.nn.functional,
             "torchsparse.nn.utils": ts.nn.utils, "torchsparse.utils": ts.utils,
             "torchsparse.utils.quantize": ts.utils.quantize, "torchsparse.utils.collate": ts.utils.collate,
             "torchsparse.backend": ts.backend, "torchsparse.tensor": ts.tensor,
             "pcseg": pc, "pcseg.model": pc.model, "pcseg.loss": pc.loss}
    for name, mod in names.items():
        sys.modules.setdefault(name, mod)
    return names
