"""Host cost of one conv -> BatchNorm -> ReLU block call (spnn.conv_bn_act) with the device out of the picture: a 3x3x3 block
on ~2 000 voxels (microseconds of kernels), thousands of calls, wall time per call.  eval = no graph (ts_conv_block_eval through the
C++ caller), train = forward of the C++ autograd node, train+bwd = forward + backward of one block.
    python tools/block_host_probe.py [--amp] [--calls 3000]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import taseg_amd.torchsparse.nn as spnn  # noqa: E402
from taseg_amd.torchsparse import SparseTensor  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--amp", action="store_true")
    ap.add_argument("--calls", type=int, default=3000)
    args = ap.parse_args()
    rs = np.random.RandomState(0)
    c = np.unique(rs.randint(0, 16, size=(3000, 3)), axis=0).astype(np.int32)
    coords = torch.from_numpy(np.concatenate([c, np.zeros((len(c), 1), np.int32)], 1)).cuda()
    feats = torch.from_numpy(rs.randn(len(c), 64).astype(np.float32)).cuda()
    conv = spnn.Conv3d(64, 64, kernel_size=3).cuda()
    bn = spnn.BatchNorm(64).cuda()
    x = SparseTensor(feats, coords, 1)
    with torch.no_grad():
        spnn.conv_bn_act(conv, bn, x)          # builds and caches the kernel map

    def timed(fn, n):
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        dt = time.perf_counter() - t0
        torch.cuda.synchronize()
        return 1e6 * dt / n

    def ev():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=args.amp):
            spnn.conv_bn_act(conv, bn, x)

    def tr():
        with torch.autocast("cuda", dtype=torch.float16, enabled=args.amp):
            spnn.conv_bn_act(conv, bn, x)

    def trb():
        conv.kernel.grad = None
        bn.weight.grad = None
        bn.bias.grad = None
        with torch.autocast("cuda", dtype=torch.float16, enabled=args.amp):
            y = spnn.conv_bn_act(conv, bn, x)
        y.F.backward(y.F)

    from taseg_amd import _fast
    fast = _fast.module()
    conv.eval(), bn.eval()
    e = timed(ev, args.calls)
    conv.train(), bn.train()
    t = timed(tr, args.calls)
    if fast is not None:
        fast.host_times()
    b = timed(trb, args.calls // 3)
    if fast is not None:
        nf, tf, taf, nb, tb, tab = fast.host_times()
        print(f"  inside the C++ node: forward {tf / nf / 1e3:.1f} us ({taf / nf / 1e3:.1f} in the backend call), "
              f"backward {tb / nb / 1e3:.1f} us ({tab / nb / 1e3:.1f} in the backend call)")
    print(f"{'amp ' if args.amp else 'fp32'} voxels {len(c)}: eval {e:.1f} us / call, train forward {t:.1f}, train forward + backward {b:.1f}")


if __name__ == "__main__":
    main()
