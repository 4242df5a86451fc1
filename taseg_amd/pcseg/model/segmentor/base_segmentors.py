"""BaseSegmentor: checkpoint loading contract of the reference (base_segmentors.py:16-37)."""
import os

import torch
import torch.nn as nn


class BaseSegmentor(nn.Module):
    def __init__(self, model_cfgs, num_class: int):
        super().__init__()
        self.model_cfgs = model_cfgs
        self.num_class = num_class

    def load_params(self, model_state_disk, strict=False):
        """Load every entry whose (DDP-prefix-stripped) name and shape match this model."""
        own = self.state_dict()
        accepted = {}
        for name, value in model_state_disk.items():
            name = name[len("module."):] if name.startswith("module.") else name
            if name in own and own[name].shape == value.shape:
                accepted[name] = value
        return self.load_state_dict(accepted, strict=strict)

    def load_params_from_file(self, filename, logger, to_cpu=False):
        if not os.path.isfile(filename):
            raise FileNotFoundError
        logger.info("==> Loading parameters from checkpoint %s to %s" % (filename, "CPU" if to_cpu else "GPU"))
        state = torch.load(filename, map_location=torch.device("cpu") if to_cpu else None)
        state = state.get("model_state", state)
        msg = self.load_params(state)
        logger.info(f"==> Done {msg}")

    def forward(self, batch_dict):
        raise NotImplementedError
