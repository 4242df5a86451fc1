"""CE + Lovasz (pcseg.loss.Losses on csrc/loss.hip) at the size of TIAF's dense image loss: 10 frames of 384 x 1280 pixels x 20 classes,
half logits, forward + backward; event-timed, and the per-kernel table when run under `rocprofv3 --kernel-trace --stats`.

    python tools/loss_probe.py [rows] [classes]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import taseg_amd.pcseg.loss as LS  # noqa: E402


def main():
    torch.manual_seed(0)
    p = int(sys.argv[1]) if len(sys.argv) > 1 else 10 * 384 * 1280
    c = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    logits = (torch.randn(p, c, device="cuda") * 2).half().requires_grad_()
    labels = torch.randint(0, c, (p,), device="cuda")
    crit = LS.Losses(["CELoss", "LovLoss"], [1.0, 1.0], ignore_index=0, label_smoothing=0.0)

    def run():
        loss = crit(logits, labels)
        loss.backward()
        return loss
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        loss = run()
    e1.record()
    e1.synchronize()
    print(f"rows {p} classes {c} loss {float(loss.detach()):.6f}  forward + backward {e0.elapsed_time(e1) / 5:.3f} ms")


if __name__ == "__main__":
    main()
