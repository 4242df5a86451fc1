"""torchsparse-compatible operator API on the MI355X HIP backend.

Same public names as mit-han-lab/torchsparse v1.4.0 (the library TASeg / OpenPCSeg is
written against, vendored at /root/reference/package/torchsparse.zip):
``SparseTensor``, ``PointTensor``, ``cat``, ``nn.Conv3d/BatchNorm/ReLU``,
``nn.functional.{conv3d, sphash, sphashquery, spcount, spvoxelize, spdevoxelize,
calc_ti_weights, spdownsample}``, ``nn.utils.{get_kernel_offsets, fapply}``,
``utils.{make_ntuple, sparse_quantize, sparse_collate, sparse_collate_fn}``, ``backend``.

``taseg_amd.install_as_dropin()`` registers this package as ``torchsparse`` in
``sys.modules`` so unmodified OpenPCSeg code imports it.
"""
from .tensor import PointTensor, SparseTensor  # noqa: F401
from .operators import cat  # noqa: F401
from . import backend, nn, utils  # noqa: F401

__version__ = "1.4.0+taseg_amd"
