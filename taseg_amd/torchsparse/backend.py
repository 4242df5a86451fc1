"""`torchsparse.backend` lookalike: the reference's ten pybind entry points
(TS/torchsparse/backend/pybind_cuda.cpp:18-39) served by libtaseg_hip.so."""
from ..backend import (convolution_backward_cuda, convolution_forward_cuda, count_cuda,  # noqa: F401
                       devoxelize_backward_cuda, devoxelize_forward_cuda, hash_cuda, hash_query_cuda,
                       kernel_hash_cuda, voxelize_backward_cuda, voxelize_forward_cuda)
