"""Host / device time of ONE index plan (model.prepare: voxelisation maps, coordinate pyramid, 9 kernel maps, class / direct plans,
trilinear maps, devoxelize plans, stage geometry) of the default bench batch, alone on the device: wall time per call with a
synchronisation behind it, host time to issue it, and an interpreter profile of the issuing thread."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    from taseg_amd.data.synthetic import fill_parameters, make_model_cfg
    from taseg_amd.pcseg.model import build_network
    from taseg_amd.torchsparse import SparseTensor
    torch.cuda.set_device(0)
    model = fill_parameters(build_network(make_model_cfg("MinkUNet", in_dim=4, cr=1.0), 20), seed=1).cuda().eval()
    coords, feats, labels, npts = bench.make_scans(0, 2, 120000, "minkunet")

    def make():
        return {"lidar": SparseTensor(feats, coords), "targets": SparseTensor(labels, coords)}
    amp = "--amp" in sys.argv
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=amp):
        for _ in range(3):
            bd = make()
            plan = model.prepare(bd)
            x = bd["lidar"]
            from taseg_amd.torchsparse.nn import functional as spF
            model._unet(spF.spvoxelize(x.F, plan["vox_idx"], plan["vox_counts"]), x.F, plan)      # compiles the stage programs
        torch.cuda.synchronize()
        t_issue = t_wall = 0.0
        n = 30
        for _ in range(n):
            bd = make()
            t0 = time.perf_counter()
            model.prepare(bd)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t_issue += t1 - t0
            t_wall += time.perf_counter() - t0
        print(f"prepare(): issue {1e3 * t_issue / n:.2f} ms (includes its host reads), with the device drained {1e3 * t_wall / n:.2f} ms")
        prof = cProfile.Profile()
        prof.enable()
        for _ in range(n):
            model.prepare(make())
        torch.cuda.synchronize()
        prof.disable()
        pstats.Stats(prof).sort_stats("tottime").print_stats(28)


if __name__ == "__main__":
    main()
