"""Does a hipGraph of one step (5 kernel launches) beat 5 stream launches?  (one rank, no collectives)"""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from hinge_amd import capi, synth
from hinge_amd.config import default_filter_params
from hinge_amd.dist import BlockTable, Exchange, HipBackend, ShardedFilter
d = synth.generate(synth.CONFIGS["cfg2_ecoli160"])
pile = synth.to_pileups(d)
dev = torch.device("cuda", 0)
P = default_filter_params()
ctx = capi.Context(0)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    be = HipBackend(ctx, P, d.rlen, None, 0, d.n_reads - 1, torch.from_numpy(pile.row_ptr).to(dev), torch.from_numpy(pile.a_span).to(dev),
                    torch.from_numpy(pile.b_span).to(dev), torch.from_numpy(pile.b_flag.view(np.int32)).to(dev))
    job = ShardedFilter(be, Exchange(BlockTable([0, d.n_reads]), dev), mode="merged")
    ctx.filter_stats(P); ctx.filter_median(P, 0, d.n_reads - 1, fetch=True); ctx.filter_mask_annotate(P); ctx.filter_hinges(P)
    for _ in range(3): job.step(fetch_hinges=False)
    torch.cuda.synchronize()
    def timed(fn, n=50):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    print("stream launches: %.4f ms per step" % timed(lambda: job.step(fetch_hinges=False)))
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=s):
            job.step(fetch_hinges=False)
        print("graph replay:    %.4f ms per step" % timed(g.replay))
        ctx.check()
        print("counters after replay", ctx.counters())
    except Exception as e:
        print("capture failed:", repr(e)[:400])
