"""Segmentor registry (reference pcseg/model/segmentor/__init__.py:29-62).  Only the TASeg
hot-path models exist here; the other OpenPCSeg backbones are out of scope."""
from .base_segmentors import BaseSegmentor
from .voxel.minkunet.minkunet import MinkUNet
from .voxel.minkunet.minkunet_ms import MinkUNetMs
from .voxel.minkunet.minkunet_ms_kd import MinkUNetMsKd
from .voxel.minkunet.minkunet_ms_mm import MinkUNetMsMm, MinkUNetMsMmNus

__all__ = {
    "MinkUNet": MinkUNet,
    "MinkUNetMs": MinkUNetMs,
    "MinkUNetMsKd": MinkUNetMsKd,
    "MinkUNetMsMm": MinkUNetMsMm,
    "MinkUNetMsMmNus": MinkUNetMsMmNus,
}

_OUT_OF_SCOPE = ("RangeNet++", "SalsaNext", "FIDNet", "CENet", "Cylinder_TS", "SPVCNN", "RPVNet")


def build_segmentor(model_cfgs, num_class):
    name = model_cfgs.NAME
    if name not in __all__:
        if name in _OUT_OF_SCOPE:
            raise NotImplementedError(f"segmentor '{name}' is outside the TASeg hot path built here")
        raise NameError(f"name '{name}' is not defined")  # what the reference's eval(NAME) raises
    return __all__[name](model_cfgs=model_cfgs, num_class=num_class)
