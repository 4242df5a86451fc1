"""Diagnostic: where a workgroup of the split pair GEMM spends its life.  The PROBE instantiation of pair_gemm_s_kernel
(csrc/conv_pairs_s.hip, ts_debug_phase_stamps) stamps the shader clock at the phase boundaries of every workgroup;
this prints the mean length of each phase, the mean lifetime, the launch time and the mean number of workgroups alive
per CU.

    python tools/phase_probe.py [--layers 1:96:96,2:96:96,4:128:128,8:256:256,16:256:256]
"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from taseg_amd import _lib as L, backend as B
from taseg_amd.torchsparse import SparseTensor
from taseg_amd.torchsparse.nn import functional as spF

ap = argparse.ArgumentParser()
ap.add_argument("--layers", default="1:96:96,2:96:96,4:128:128,8:256:256,16:256:256")
args = ap.parse_args()
lib = L.load()
lib.ts_debug_phase_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int64]
lib.ts_debug_phase_stamps.restype = None
coords, feats, labels, _ = bench.make_scans(0, 2, 120000, "minkunet")
x = SparseTensor(None, coords, 1)
spF.build_pyramid(x, 4)
NAMES = ["tile map (offset table)", "pair indices", "first slice arrives", "split + LDS + barrier", "first MFMA block",
         "remaining slices", "MFMAs retire", "Z stores issue", "Z stores acknowledged"]
cap = 1 << 16
stamps = torch.zeros(cap * 16, dtype=torch.int64, device="cuda")
for spec in args.layers.split(","):
    s, ci, co = (int(v) for v in spec.split(":"))
    km = x.kmaps[((s, s, s), (3, 3, 3), (1, 1, 1), (1, 1, 1))]
    n, P = km.sizes[0], km.total
    xf = torch.randn(n, ci, device="cuda")
    w = torch.randn(27, ci, co, device="cuda") * 0.05
    planes = torch.empty(3 * w.numel(), dtype=torch.int16, device="cuda")
    L.check(lib.ts_conv_split_planes(w.data_ptr(), 27, ci, co, planes.data_ptr(), L.stream()), "split")
    gy = torch.randn(n, co, device="cuda")
    for wt, pre in ((False, 0), (False, 2), (True, 0), (True, 2)):
        src = gy if wt else xf
        plain = lambda: B.conv_pair_gemm(src, w, km.nbmaps_buf, km.nboffs, P, 1 if wt else 0, weight_transposed=wt)
        call = plain
        if pre:
            ref = plain()

            def call():
                lib.ts_conv_planes_hint(w.data_ptr(), planes.data_ptr(), 27, ci, co)
                return plain()
            B.set_conv_impl(11)
            got = call()
            if not torch.equal(ref, got):
                print(f"direct rows: Z differs, max {(ref - got).abs().max().item():.3e}")
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            call()
        e1.record()
        torch.cuda.synchronize()
        plain_us = e0.elapsed_time(e1) * 100
        lib.ts_debug_phase_stamps(stamps.data_ptr(), cap)
        call()
        torch.cuda.synchronize()
        lib.ts_debug_phase_stamps(None, 0)
        B.set_conv_impl(0)
        st = stamps.cpu().numpy().reshape(-1, 16)
        live = st[:, 9] != 0
        st = st[live]
        cyc = (st[:, 9] - st[:, 0]).astype(np.float64)
        rt = (st[:, 10] - st[:, 13]).astype(np.float64)          # 100 MHz ticks
        mhz = cyc.sum() / rt.sum() * 100.0
        span_us = (st[:, 10].max() - st[:, 13].min()) / 100.0
        d = np.diff(st[:, :10].astype(np.float64), axis=1) / mhz
        life = cyc / mhz
        hw = st[:, 11]
        cu_key = ((hw >> 32) & 0xF) * 4096 + (hw & 0xFFFFFFFF & 0xFF00 | ((hw >> 13) & 7) << 16)    # xcc, se, sh, cu
        alive = life.sum() / span_us / len(np.unique(cu_key))
        print(f"s{s} {ci}->{co} {'dgrad' if wt else 'fwd  '}{('', '', ' direct rows')[pre]}: P={P} {len(st)} workgroups on {len(np.unique(cu_key))} CUs, "
              f"launch {plain_us:.1f} us (probe span {span_us:.1f} us), counter {mhz:.0f} MHz, lifetime {life.mean():.2f} us "
              f"(p10 {np.percentile(life, 10):.2f}, p90 {np.percentile(life, 90):.2f}), {alive:.2f} workgroups alive per CU")
        print("     " + " | ".join(f"{nm} {v:.2f}" for nm, v in zip(NAMES, d.mean(0))))
