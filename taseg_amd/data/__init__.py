from .synthetic import (AttrDict, FLEXIBLE_STEPS_KITTI, fill_parameters, make_model_cfg, synth_pose,  # noqa: F401
                        synth_scan)
