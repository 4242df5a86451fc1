"""Optional native fast path (taseg_amd/_fast_block.so, built by taseg_amd.csrc.fastpath.build): the conv -> BatchNorm
[-> residual] [-> ReLU] block as a C++ autograd node.  It issues exactly the backend calls of the Python Function
(`functional._ConvBlock`); when the binding has not been built, or TASEG_FAST_BLOCK=0, the Python node serves the
block - both run the same HIP kernels of libtaseg_hip.so."""
import importlib.util
import os

from . import _lib as L
from .options import options

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
    if not options.fast_block or not os.path.exists(_PATH):
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
    if options.wgrad_stream == "1":
        wgrad_stream(True)
    return _mod


_side = None
_single_stream_only = None      # reason (str) while the second stream must stay off


def require_single_stream(reason):
    """From now on (until `reason` is None again) wgrad_stream(True) is refused and the second stream is off.  Set by
    parallel.GradBucketReducer when gradients are averaged over MORE THAN ONE rank: a gradient bucket's all-reduce reads the slots
    in the middle of the backward pass, right behind a join of the second stream, and that combination has failed intermittently in
    every two-rank rehearsal it was given (round 5: a memory-access fault with SyncBatchNorm; round 6: a memory-access fault, a
    crashed rank and gradients that were not the one-stream run's, tests/test_gpu_dist.py) without a cause found.  Until it has
    been, N > 1 keeps every weight gradient on the stream of its backward pass."""
    global _single_stream_only
    _single_stream_only = reason
    if reason is not None and _mod is not None:
        _mod.set_wgrad_stream(0, 0, 0, 0)


def wgrad_stream(on):
    """Weight gradients of the block backward on a second stream (csrc/block.hip, TsConvBlockOpts.wgrad_stream): the node hands
    every block's weight gradient to one side stream and joins it when the backward pass ends.  TASEG_WGRAD_STREAM=1 switches it
    on at import; gradient buckets (taseg_amd.parallel.GradBucketReducer) join it before a bucket's all-reduce."""
    global _side
    mod = module()
    if mod is None:
        return False
    import torch
    if on and _single_stream_only is not None:
        on = False
    if on and torch.cuda.is_available():
        if _side is None:
            _side = torch.cuda.Stream()
        mod.set_wgrad_stream(int(_side.cuda_stream), int(_side.stream_id), int(_side.device_index), int(_side.device_type))
        return True
    mod.set_wgrad_stream(0, 0, 0, 0)
    return False


def join_wgrad_stream():
    """The current stream waits for every weight gradient handed to the second stream so far (a no-op while it is off): what a
    consumer of p.grad in the MIDDLE of a backward pass needs - parallel.GradBucketReducer calls it before a bucket's all-reduce."""
    if _mod is not None and _side is not None:
        _mod.join_wgrad_stream(L.stream())


def tune_wgrad_stream(step, fence, rounds=3, steps=4):
    """Time `step()` (one full training step: forward, backward, optimizer) with the weight gradients on the caller's stream and on
    the second stream - `rounds` alternating rounds of `steps` steps after one unmeasured step each, `fence()` = device (and rank)
    synchronisation - and keep the faster setting.  The second stream is worth 2-4 % of a device-bound step; a host-bound one gains
    nothing from it and, before the launches moved to a thread of their own, lost 10-20 % to the event calls
    (profiles/r04_ab_wgrad_stream.txt) - only a measurement on the actual model, batch and machine tells the cases apart.  TASEG_WGRAD_STREAM=0 / 1 pins the setting.  Returns (chosen, ms_off, ms_on)."""
    import time
    # (several ranks: every rank runs the same number of steps here - they contain collectives - and may end up with its own
    # setting: the second stream is rank-local, the sequence of collectives does not depend on it; the gradient buckets join the
    # second stream before their all-reduce, parallel.GradBucketReducer._launch)
    pinned = options.wgrad_stream
    if _single_stream_only is not None:
        return (wgrad_stream(False) if module() is not None else False), None, None
    if module() is None or pinned in ("0", "1"):
        return (wgrad_stream(pinned == "1") if module() is not None else False), None, None
    best = {False: float("inf"), True: float("inf")}
    wins = 0
    for _ in range(rounds):
        t = {}
        for on in (False, True):
            if wgrad_stream(on) != on:
                return False, None, None
            step()
            fence()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            fence()
            t[on] = (time.perf_counter() - t0) / steps * 1e3
            best[on] = min(best[on], t[on])
        wins += t[True] < 0.99 * t[False]
    # the second stream must win all rounds but one by > 1 % and the best-of-rounds by > 2 %: a host-bound line, whose step time
    # wanders by several per cent between runs, keeps one stream (with "every round" one slow round of four steps cost a
    # device-bound line the 4 % the second stream is worth: 15.85 / 15.15 ms best-of-rounds and still "off")
    choice = wins >= max(rounds - 1, 1) and best[True] < 0.98 * best[False]
    wgrad_stream(choice)
    return choice, best[False], best[True]
