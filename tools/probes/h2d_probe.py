import time, torch, numpy as np
n = 2 << 30
a = torch.empty(n, dtype=torch.uint8); a.fill_(1)
d = torch.empty(n, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for k in range(3):
    t = time.time(); d.copy_(a); torch.cuda.synchronize(); print("pageable 2 GiB: %.1f ms = %.1f GB/s" % ((time.time()-t)*1e3, n/ (time.time()-t)/1e9))
t = time.time(); p = torch.empty(256 << 20, dtype=torch.uint8).pin_memory(); print("pin 256 MiB: %.1f ms" % ((time.time()-t)*1e3))
t = time.time(); p2 = torch.empty(32 << 20, dtype=torch.uint8).pin_memory(); print("pin 32 MiB: %.1f ms" % ((time.time()-t)*1e3))
for k in range(3):
    t = time.time(); d[:256 << 20].copy_(p, non_blocking=True); torch.cuda.synchronize(); print("pinned 256 MiB: %.2f ms = %.1f GB/s" % ((time.time()-t)*1e3, (256<<20)/(time.time()-t)/1e9))
# host memcpy rate, one thread
b = torch.empty(256 << 20, dtype=torch.uint8)
for k in range(2):
    t = time.time(); p.copy_(a[:256 << 20]); print("host memcpy 256 MiB (torch, threads=%d): %.1f ms = %.1f GB/s" % (torch.get_num_threads(), (time.time()-t)*1e3, (256<<20)/(time.time()-t)/1e9))
