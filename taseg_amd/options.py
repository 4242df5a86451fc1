"""The library's options in one typed object.

`taseg_amd.options.options` is the single place where configuration lives.  Set fields programmatically BEFORE the modules that use
them are imported / the first model is built (`from taseg_amd.options import options; options.rccl_direct = "borrow"`), or scope
a change with `options.override(name=value)`.  The environment is an OVERRIDE for diagnostics and A/B measurements only: a variable
`TASEG_<FIELD NAME IN CAPITALS>` (e.g. `TASEG_CLASS_GEMM=0`) replaces the default of that field when this module is imported -
nothing else in `taseg_amd/` reads `os.environ` for configuration.  Unknown `TASEG_*` variables that look like options are
reported once (a typo must not silently measure the default).

Three groups:
  * supported configuration - transports, staging depth, the library path;
  * A/B switches - the older path behind each feature, same results (bit-equal unless the comment says otherwise), kept so that the
    steps of HISTORY.md can be re-measured;
  * thresholds - where one kernel family takes over from another (defaults from the probes under profiles/).
"""
import os
import warnings
from contextlib import contextmanager
from dataclasses import dataclass, fields

__all__ = ["Options", "options"]


@dataclass
class Options:
    # ---- supported configuration
    hip_lib: str = ""                  # another build of libtaseg_hip.so ("" = in-tree)
    rccl_direct: str = "c10d"          # SyncBatchNorm's statistics all-reduces: "c10d" (torch.distributed, default) | "borrow" | "create" (rccl.py)
    wgrad_stream: str = "auto"         # weight gradients of the block backward: "0" caller's stream | "1" second stream | "auto" tuned
    eval_stage_depth: int = 2          # batches staged ahead in the evaluation loops
    image_layout: str = "nhwc"         # UNet2D's memory format and the image gather's form: "nhwc" (channels_last rows) | "nchw" (planes)
    image_fused_bn: bool = True        # UNet2D: LeakyReLU + training BatchNorm2d of a channels-last map as one node on csrc/bn.hip
    image_conv_rows: bool = True       # UNet2D: the 3x3 layers with up to 96 input channels of a channels-last half map on csrc/conv2d_rows.hip
    image_shuffle_cat: bool = True     # UNet2D: UpBlock's PixelShuffle + Dropout2d + concat + Dropout2d as one pass (csrc/shuffle_cat.hip)
    # ---- rehearsals of the N > 1 path on one card (tests, bench.py --force-dist)
    syncbn_single_rank: bool = False   # SyncBatchNorm takes the collective path in a one-rank group
    dist_buckets_on_default_group: bool = False
    # ---- A/B switches (default = the current path)
    stage_program: bool = True         # one autograd node per encoder / decoder stage (False: one per block)
    direct_grads: bool = True          # a stage's parameter gradients delivered to the reducer in one call
    fast_block: bool = True            # C++ binding for the block nodes / index plan (False: Python nodes)
    fused_block: bool = True           # conv3d + BatchNorm + activation as one node
    pointwise_block: bool = True       # 1x1x1 shortcut as one call
    class_gemm: bool = True            # class-sorted implicit GEMM (False: pair GEMM + gather-sum; other summation order, 1e-6-close)
    direct_conv: str = "1"             # one-pass 2x2x2 plans: "1" | "0" | "force" (every fitting product, tests / probes)
    presplit: bool = True              # pre-split weight planes / kept half weights
    kmap_sym: bool = True              # submanifold kernel maps on half the probes
    kd_loss_on_device: bool = True     # MinkUNetMsKd's feature distillation without host reads
    eval_copy_stream: bool = True      # the deferred evaluation tail's device -> host copies on a stream of their own
    fused_eval_tail: bool = True
    eval_tail_in_pass2: bool = True
    fused_loss: bool = True            # CE + Lovasz in csrc/loss.hip (False: tensor ops, 1e-6-close)
    fused_lovasz: bool = True
    devox_cells: bool = True           # stride-16 devoxelize backward as a cell-reduced two-stage sum
    devox_atomic: bool = False         # ... with run-wise float atomics (last-bit noise)
    gather_positions: bool = False     # pass 2 with K position registers per lane instead of LDS lists
    stage_batched: bool = True         # the multi-scan data stage as one launch chain per batch
    # ---- thresholds
    class_min_rows_96: int = 48000
    class_min_rows_128: int = 60000
    class_min_rows_half: int = 16384
    class_finish_rows: int = 0         # 0 = the kernel's own default
    class_finish_rows_half: int = 0
    direct_min_rows: int = 0
    wgrad_wgs: int = 0                 # 0 = the kernel's own target
    # ---- diagnostics
    debug_bn_ablate: str = ""

    # legacy spellings of values the environment may still carry
    _ALIASES = {"rccl_direct": {"0": "c10d", "1": "create"}}
    _CHOICES = {"rccl_direct": ("c10d", "borrow", "create"), "wgrad_stream": ("0", "1", "auto"), "direct_conv": ("0", "1", "force"),
                "image_layout": ("nhwc", "nchw")}

    def _coerce(self, name, value):
        kind = type(self._defaults[name])
        if isinstance(value, str):
            value = self._ALIASES.get(name, {}).get(value, value)
            if kind is bool:
                if value not in ("0", "1"):
                    raise ValueError(f"taseg_amd option {name}: '{value}' is not 0 / 1")
                value = value == "1"
            elif kind is int:
                value = int(value)
        if not isinstance(value, kind):
            raise TypeError(f"taseg_amd option {name}: expected {kind.__name__}, got {type(value).__name__}")
        if name in self._CHOICES and value not in self._CHOICES[name]:
            raise ValueError(f"taseg_amd option {name}: '{value}' is not one of {self._CHOICES[name]}")
        return value

    def __post_init__(self):
        object.__setattr__(self, "_defaults", {f.name: getattr(self, f.name) for f in fields(self)})

    def __setattr__(self, name, value):
        if name.startswith("_"):
            return object.__setattr__(self, name, value)
        if hasattr(self, "_defaults"):
            if name not in self._defaults:
                raise AttributeError(f"taseg_amd has no option '{name}'")
            value = self._coerce(name, value)
        object.__setattr__(self, name, value)

    def load_environment(self, environ=None):
        """TASEG_<FIELD> overrides (diagnostics only); returns the names that were overridden"""
        environ = os.environ if environ is None else environ
        taken = []
        for name in self._defaults:
            raw = environ.get("TASEG_" + name.upper())
            if raw is not None and raw != "":
                setattr(self, name, raw)
                taken.append(name)
        known = {"TASEG_" + n.upper() for n in self._defaults}
        # variables of bench.py / the tests / the tools are theirs, not options of the library
        foreign = ("TASEG_BENCH_", "TASEG_WORKER_", "TASEG_DIST_BACKEND", "TASEG_STAGE_THREAD", "TASEG_STAGE_EARLY", "TASEG_STAGE_DEPTH",
                   "TASEG_REUSE_PLAN", "TASEG_TOOL_")
        for key in environ:
            if key.startswith("TASEG_") and key not in known and not key.startswith(foreign):
                warnings.warn(f"taseg_amd: environment variable {key} is not an option of this library (see taseg_amd/options.py)")
        return taken

    @contextmanager
    def override(self, **kw):
        """scope a change: `with options.override(class_gemm=False): ...` (fields read at import time of their module are not
        affected - see the module docstring)"""
        old = {k: getattr(self, k) for k in kw}
        try:
            for k, v in kw.items():
                setattr(self, k, v)
            yield self
        finally:
            for k, v in old.items():
                setattr(self, k, v)

    def as_dict(self):
        return {name: getattr(self, name) for name in self._defaults}


options = Options()
options.load_environment()
