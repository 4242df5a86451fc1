"""The bench-size backward test (tests/test_gpu_bench_backward.py) on the block-diagonal bs-2 batch the default `bench.py` line times
(2 x 120 000 points -> ~178k voxels), for two seeds: loss, logits and all 191 gradient norms against the oracle on the reference's
own CPU kernels and a float64 evaluation.  ~4 minutes of CPU oracle per case: run once per round, output kept under profiles/.

    python tools/bench_backward_bs2.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import test_gpu_bench_backward as T  # noqa: E402

for rank in (0, 1):
    T.run_bench_scan_case(rank=rank, batch=2)
print("both bs-2 cases within the bars")
