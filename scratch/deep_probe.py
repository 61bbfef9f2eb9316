import sys, os, tempfile, dataclasses
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import conftest
from hinge_amd import synth, capi, stages
for cov, copies, G in ((450, (3, 3), 50_000), (520, (3, 3), 40_000), (400, (4, 4), 50_000), (330, (5, 5), 60_000)):
    d = synth.generate(dataclasses.replace(synth.CONFIGS["deep"], coverage=cov, repeat_copies=copies, genome_len=G))
    cnt = np.bincount(d.aread, minlength=d.n_reads)
    wd = tempfile.mkdtemp(); synth.write_dataset(d, wd, "G", write_bases=False); conftest.write_ini(os.path.join(wd, "nominal.ini"))
    os.chdir(wd)
    ctx = capi.Context(0)
    rc = stages.run_filter("G", "G.las", "G", "nominal.ini", False, 0, True, False, ctx)
    sys.stderr.write("cov %d copies %s G %d: reads %d novl %d max pile %d, >2048: %d, >4096: %d rc %d\n" % (cov, copies, G, d.n_reads, d.novl, cnt.max(), (cnt > 2048).sum(), (cnt > 4096).sum(), rc))
    ctx.counters()
