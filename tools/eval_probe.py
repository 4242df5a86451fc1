"""Diagnostic: where an evaluation pass of the bench workload spends its time - index plan, eval-mode forward to the logits, the
eval branch's un-voxelisation + copies to the host (synchronised after every stage, so the sum is above the pipelined pass).
     python tools/eval_probe.py [--amp]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from taseg_amd.data.synthetic import fill_parameters, make_model_cfg
from taseg_amd.pcseg.model import build_network
from taseg_amd.pcseg.model.segmentor.voxel.minkunet.minkunet import unvoxelise_predictions
from taseg_amd.torchsparse import SparseTensor
from taseg_amd.torchsparse.nn import functional as spF

ap = argparse.ArgumentParser()
ap.add_argument("--amp", action="store_true")
ap.add_argument("--iters", type=int, default=30)
args = ap.parse_args()
model = fill_parameters(build_network(make_model_cfg("MinkUNet", in_dim=4, cr=1.0), 20), seed=1).cuda().eval()
coords, feats, labels, npts = bench.make_scans(0, 2, 120000, "minkunet")
offset = torch.tensor([len(coords)], device="cuda", dtype=torch.int32)
counts = torch.bincount(coords[:, 3].long())
inv = torch.cat([torch.arange(int(c), device="cuda") for c in counts])


def batch():
    return {"lidar": SparseTensor(feats, coords), "targets": SparseTensor(labels, coords), "offset": offset,
            "targets_mapped": SparseTensor(labels, coords), "inverse_map": SparseTensor(inv, coords), "num_points": counts,
            "name": ["a", "b"]}


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


acc = {"plan": 0.0, "forward": 0.0, "tail": 0.0, "whole pass (pipelined)": 0.0}
for it in range(args.iters + 5):
    bd = batch()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=args.amp):
        t0 = sync()
        plan = model.prepare(bd)
        t1 = sync()
        x = bd["lidar"]
        f = spF.spvoxelize(x.F[:, :4], plan["vox_idx"], plan["vox_counts"])
        out = model._unet(f, x.F[:, :4], plan)
        t2 = sync()
        res = unvoxelise_predictions(out, x.C[:, -1], bd["inverse_map"], bd["targets_mapped"], bd["num_points"], False, names=bd["name"])
        t3 = sync()
        model(batch())
        t4 = sync()
    if it >= 5:
        for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            acc[k] += v
print(f"eval pass{' (autocast)' if args.amp else ''}, bs 2, {coords.shape[0]} voxels, ms per batch: " +
      ", ".join(f"{k} {1e3 * v / args.iters:.2f}" for k, v in acc.items()) + f"; logits dtype {out.dtype}")
