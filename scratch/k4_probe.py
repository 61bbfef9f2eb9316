"""rocprofv3 kernel stats of get_maximal_reads on the e2e data set (k_trim_classify = SURVEY 8(d)'s K4)."""
import dataclasses, os, subprocess, sys, tempfile, glob
sys.path.insert(0, "/root/repo")
from hinge_amd import synth
import numpy as np
spec = dataclasses.replace(synth.CONFIGS["cfg2_ecoli160"], genome_len=4_600_000, n_repeat_families=1, repeat_copies=(3, 3), n_blocks=1)
d = synth.generate(spec)
wd = tempfile.mkdtemp(prefix="k4_")
synth.write_dataset(d, wd, "G", write_bases=False)
open(os.path.join(wd, "nominal.ini"), "w").write("[filter]\nlength_threshold = 1000;\naln_threshold = 1000;\nmin_cov = 5;\ncut_off = 300;\ntheta = 300;\n[layout]\nhinge_slack = 1000\nmin_connected_component_size = 8\n")
B = "/root/repo/hinge_amd/bin/"
subprocess.run([B + "Reads_filter", "--db", "G", "--las", "G.las", "-x", "G", "--config", "nominal.ini"], cwd=wd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
out = "/tmp/k4prof"
env = dict(os.environ, TMPDIR="/tmp", HINGE_SLOW_EXIT="1")   # (the fast _exit() would skip the profiler's own exit handler)
subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "-d", out, "-o", "k4", "--output-format", "csv", "--", B + "get_maximal_reads", "--db", "G", "--las", "G.las", "-x", "G", "--config", "nominal.ini"],
               cwd=wd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env)
out2 = "/tmp/k4pmc"
subprocess.run(["rocprofv3", "--pmc", "SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "-d", out2, "-o", "k4", "--output-format", "csv", "--", B + "get_maximal_reads", "--db", "G", "--las", "G.las", "-x", "G", "--config", "nominal.ini"],
               cwd=wd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env)
import csv
for ff in glob.glob(out2 + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(ff)):
        if "trim_classify" in row["Kernel_Name"]:
            print(row["Counter_Name"], row["Counter_Value"])
f = glob.glob(out + "/**/*kernel_stats.csv", recursive=True)[0]
print("\n".join(l[:40] + " ... " + l[-90:] for l in open(f).read().splitlines()[:4]))
tl = np.fromfile(os.path.join(wd, "G.las"), dtype=np.uint8, count=0)
from hinge_amd import formats
recs = formats.read_las(os.path.join(wd, "G.las"))
print("overlaps", recs.novl, "mean tlen (trace values)", float(recs.rec["tlen"].mean()), "tbytes", 1 if recs.tspace <= 125 else 2)
