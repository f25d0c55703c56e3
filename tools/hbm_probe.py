"""Measure what this MI355X box sustains for pure write / pure read / copy streams (torch ops on 2 GiB), the
ceilings the partials kernels are compared with in DESIGN.md."""
import torch, time
n = 512 * 1024 * 1024
a = torch.empty(n, dtype=torch.float32, device="cuda")
b = torch.empty(n, dtype=torch.float32, device="cuda")
def bench(f, bytes_moved, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return bytes_moved / ms / 1e9
print("write (fill_)   %.2f TB/s" % bench(lambda: a.fill_(1.0), 4 * n))
print("write (zero_)   %.2f TB/s" % bench(lambda: a.zero_(), 4 * n))
print("read  (sum)     %.2f TB/s" % bench(lambda: a.sum(), 4 * n))
print("copy  (r+w)     %.2f TB/s" % bench(lambda: b.copy_(a), 8 * n))
print("axpy  (2r+w)    %.2f TB/s" % bench(lambda: torch.add(a, b, alpha=2.0, out=b), 12 * n))
