"""Diagnostic: pair-GEMM throughput vs reduction depth (is the kernel prologue/epilogue bound or per-step bound?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from taseg_amd import backend as B

dev = "cuda"
torch.manual_seed(0)
n, P, K = 100000, 300000, 27
pairs = torch.stack([torch.randint(0, n, (P,), device=dev), torch.arange(P, device=dev) % n], 1).int().contiguous()
offs = torch.tensor([min(P, (P // K + 1) * k) for k in range(K)] + [P], device=dev, dtype=torch.int32)
for cin, cout in ((32, 128), (128, 128), (512, 128), (2048, 128), (128, 96), (1024, 96), (256, 256)):
    x = torch.randn(n, cin, device=dev)
    w = torch.randn(K, cin, cout, device=dev) * 0.05
    for _ in range(3):
        z = B.conv_pair_gemm(x, w, pairs, offs, P, 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        z = B.conv_pair_gemm(x, w, pairs, offs, P, 0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"Cin={cin:5d} Cout={cout:4d}  {ms*1e3:8.1f} us  {2.0*P*cin*cout/ms/1e9:7.1f} TF/s  steps/tile={cin//32}")
