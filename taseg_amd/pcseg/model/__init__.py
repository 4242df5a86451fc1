"""`pcseg.model` entry points the OpenPCSeg trainer calls (reference pcseg/model/__init__.py:10-48)."""
import numpy as np
import torch

from ...torchsparse import SparseTensor
from .segmentor import build_segmentor

__all__ = ["build_network", "load_data_to_gpu"]


def build_network(model_cfgs, num_class):
    """model_cfgs.NAME selects the segmentor class; returns an nn.Module (model/__init__.py:10-15)."""
    return build_segmentor(model_cfgs=model_cfgs, num_class=num_class)


def load_data_to_gpu(batch_dict):
    """Move a collated batch to the current ROCm device in place (model/__init__.py:17-31):
    tensors, SparseTensors, ndarrays and dicts of tensors move; lists pass through."""
    for key, val in batch_dict.items():
        if isinstance(val, (torch.Tensor, SparseTensor)):
            batch_dict[key] = val.cuda()
        elif isinstance(val, np.ndarray):
            batch_dict[key] = torch.from_numpy(val).cuda()
        elif isinstance(val, dict):
            for k, v in val.items():
                val[k] = v.cuda()
        elif isinstance(val, list):
            pass
        else:
            raise ValueError("Invalid type of batch_dict", key, type(val))
