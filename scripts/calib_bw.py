"""Calibrate achievable HBM bandwidth on this box with torch fill / copy of bench-sized buffers."""
import torch, time
dev = torch.device("cuda")
n = 1024 * 2160 * 3840
a = torch.empty(n, dtype=torch.uint8, device=dev)
b = torch.empty(n, dtype=torch.uint8, device=dev)
def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
t = timeit(lambda: a.fill_(7)); print(f"fill  {n/1e9:.2f} GB: {t:.3f} ms  {n/t/1e6:.0f} GB/s written")
t = timeit(lambda: b.copy_(a)); print(f"copy  {n/1e9:.2f} GB: {t:.3f} ms  {2*n/t/1e6:.0f} GB/s r+w")
a32 = a.view(torch.int32); 
t = timeit(lambda: a32.sum()); print(f"read  {n/1e9:.2f} GB: {t:.3f} ms  {n/t/1e6:.0f} GB/s read")
c = torch.empty(n // 4, dtype=torch.int32, device=dev)
t = timeit(lambda: torch.add(a32, 1, out=c)); print(f"add   r+w {2*n/1e9:.2f} GB: {t:.3f} ms  {2*n/t/1e6:.0f} GB/s")
