import torch, time
dev = torch.device("cuda", 0)
n = 26_220_172
parts = [torch.randint(0, 1 << 30, (n,), dtype=torch.int32, device=dev) for _ in range(4)]
big = [torch.randint(0, 1 << 30, (8 * n,), dtype=torch.int32, device=dev) for _ in range(2)]
def timeit(f, tensors, reps=20):
    for t in tensors: f(t)
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        for t in tensors:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); f(t); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    return ms[len(ms) // 2]
for name, f in (("sum int32", lambda t: t.sum()), ("max int32", lambda t: t.max()), ("view int64 sum", lambda t: t[: (t.numel() // 2) * 2].view(torch.int64).sum())):
    m = timeit(f, parts)
    print("%-16s 105 MB part (4 rotating): %.1f us -> %.2f TB/s" % (name, m * 1e3, 4 * n / (m * 1e-3) / 1e12))
    m = timeit(f, big, 5)
    print("%-16s 839 MB buffer (2 rotating): %.1f us -> %.2f TB/s" % (name, m * 1e3, 32 * n / (m * 1e-3) / 1e12))
# a plain copy of one part's span copy (read 105 MB + write 105 MB), and a read of 105 MB with a 74 MB write (K2's real traffic shape)
outs = [torch.empty_like(p) for p in parts]
import itertools
def copy_all():
    pass
evs = []
for p, o in zip(parts, outs): o.copy_(p)
torch.cuda.synchronize()
for _ in range(20):
    for p, o in zip(parts, outs):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); o.copy_(p); b.record(); evs.append((a, b))
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in evs); m = ms[len(ms) // 2]
print("copy 105 MB -> 105 MB (4 rotating): %.1f us -> %.2f TB/s moved" % (m * 1e3, 8 * n / (m * 1e-3) / 1e12))
k = int(n * 0.7)
evs = []
for _ in range(20):
    for p, o in zip(parts, outs):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); s = p.view(torch.int64).sum(); o[:k].copy_(p[:k]); b.record(); evs.append((a, b))
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in evs); m = ms[len(ms) // 2]
print("read 105 MB (sum) + copy 74 MB (2 kernels): %.1f us" % (m * 1e3))
