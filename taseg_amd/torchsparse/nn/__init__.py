from . import functional, utils  # noqa: F401
from .modules import BatchNorm, Conv3d, LeakyReLU, PointLinear, ReLU, SyncBatchNorm, bn_act, conv_bn_act  # noqa: F401
