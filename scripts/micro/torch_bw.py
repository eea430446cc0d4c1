import torch, time
n = 4*1024**3
a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty(n, dtype=torch.uint8, device="cuda")
a.fill_(1)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps
ms = t(lambda: b.copy_(a)); print("copy 4 GiB: %.3f ms, %.2f TB/s traffic" % (ms, 2*n/ms/1e9))
ms = t(lambda: b.fill_(3)); print("fill 4 GiB: %.3f ms, %.2f TB/s" % (ms, n/ms/1e9))
a32 = a.view(torch.int32)
ms = t(lambda: a32.sum()); print("sum (read) 4 GiB: %.3f ms, %.2f TB/s" % (ms, n/ms/1e9))
c = torch.empty(n, dtype=torch.uint8, device="cuda")
af, bf, cf = a.view(torch.float32), b.view(torch.float32), c.view(torch.float32)
ms = t(lambda: torch.add(af, bf, out=cf)); print("add 2R+1W: %.3f ms, %.2f TB/s" % (ms, 3*n/ms/1e9))
