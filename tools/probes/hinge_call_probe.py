import os, sys, dataclasses, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from hinge_amd import benchsets, capi, synth
from hinge_amd.config import default_filter_params
from hinge_amd.dist import resident_batch
wl = sys.argv[1]
P = default_filter_params()
base = synth.CONFIGS[wl]
rp = benchsets.rank_part(base, 1, 0, 0)
dev = torch.device("cuda", 0)
batch, ctxs = resident_batch([rp], P, dev)
ctx = ctxs[0]
ctx.filter_sweep(P, fetch=True); ctx.filter_hinges(P)
for _ in range(3): batch.step()
torch.cuda.synchronize()
print(wl, "counters (work reads, exact, annotations, hinges):", ctx.counters(), "heavy items:", ctx.heavy_items())
