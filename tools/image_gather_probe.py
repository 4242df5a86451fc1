"""The TIAF image gather / adjoint (csrc/image.hip) on the FOV points of one bench TIAF batch (bench.make_tiaf_frames + the device
data stage), without the model: a few launches per map for rocprofv3 (--kernel-trace --stats for durations, --pmc FETCH_SIZE /
WRITE_SIZE in separate passes for the HBM bytes; profiles/parse_traffic.py applies the gfx950 corrections).  Prints the
algorithmic bytes per launch next to the event-timed figures of bench.tiaf_gather_roofline.

    python tools/image_gather_probe.py [nhwc|nchw] [f16|f32]          (default: nhwc f16, the autocast TIAF step's form)
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    from taseg_amd.data.synthetic import FLEXIBLE_STEPS_KITTI
    from taseg_amd.data.tiaf import build_tiaf_batch, build_tiaf_sample
    torch.cuda.set_device(0)
    frames, proj, _ = bench.make_tiaf_frames(0, 2, 120000)
    samples = [build_tiaf_sample(fr, FLEXIBLE_STEPS_KITTI, bench.TIAF_MULTISCAN, bench.TIAF_STEP_IMAGE, proj,
                                 (bench.TIAF_HEIGHT, bench.TIAF_WIDTH), 0.05, name=f"0/{b}") for b, fr in enumerate(frames)]
    bd = build_tiaf_batch(samples)
    layout = sys.argv[1] if len(sys.argv) > 1 else "nhwc"
    dtype = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else torch.float16
    out = bench.tiaf_gather_roofline(None, bd, layout, dtype)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
