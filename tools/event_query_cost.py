"""Host cost of torch.cuda.Event.query() (hipEventQuery) on a completed and on a pending event, and of creating + recording an event."""
import time
import torch

x = torch.empty(64 << 20, device="cuda")
s = torch.cuda.current_stream()
ev = torch.cuda.Event()
ev.record()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000):
    ev.query()
t_done = (time.perf_counter() - t0) / 2000
# keep the device busy, query an event behind the work
for _ in range(200):
    x.add_(1.0)
ev2 = torch.cuda.Event()
ev2.record()
t0 = time.perf_counter()
n = 0
while n < 2000:
    ev2.query()
    n += 1
t_pend = (time.perf_counter() - t0) / 2000
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000):
    e = torch.cuda.Event()
    e.record()
t_rec = (time.perf_counter() - t0) / 2000
torch.cuda.synchronize()
print(f"query (completed) {1e6 * t_done:.1f} us, query (pending, device busy) {1e6 * t_pend:.1f} us, create + record {1e6 * t_rec:.1f} us")
