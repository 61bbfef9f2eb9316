import dataclasses, os, subprocess, sys, tempfile, time
sys.path.insert(0, "/root/repo")
from hinge_amd import synth
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4_600_000
spec = dataclasses.replace(synth.CONFIGS["cfg2_ecoli160"], genome_len=G, n_repeat_families=1, repeat_copies=(3, 3), n_blocks=1)
d = synth.generate(spec)
wd = tempfile.mkdtemp(prefix="probe_")
synth.write_dataset(d, wd, "G", write_bases=False)
open(os.path.join(wd, "nominal.ini"), "w").write("[filter]\nlength_threshold = 1000;\naln_threshold = 1000;\nmin_cov = 5;\ncut_off = 300;\ntheta = 300;\n[layout]\nhinge_slack = 1000\nmin_connected_component_size = 8\n")
tiny = synth.generate(synth.CONFIGS["tiny"])
wt = tempfile.mkdtemp(prefix="probe_tiny_")
synth.write_dataset(tiny, wt, "G")
open(os.path.join(wt, "nominal.ini"), "w").write(open(os.path.join(wd, "nominal.ini")).read())
B = "/root/repo/hinge_amd/bin/"
def run(tag, cwd, argv, env=None, reps=3):
    for r in range(reps):
        t0 = time.perf_counter()
        ru0 = os.times()
        p = subprocess.run(argv, cwd=cwd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, **(env or {})))
        ru1 = os.times()
        dt = time.perf_counter() - t0
        tot = [l for l in p.stderr.decode().splitlines() if "TOTAL" in l and "las.load" not in l]
        print("%-34s wall %6.1f ms  child user %6.1f sys %6.1f   %s" % (tag, dt * 1e3, (ru1.children_user - ru0.children_user) * 1e3, (ru1.children_system - ru0.children_system) * 1e3, tot[-1] if tot else ""), flush=True)
flt = [B + "Reads_filter", "--db", "G", "--las", "G.las", "-x", "G", "--config", "nominal.ini"]
lay = [B + "hinging", "--db", "G", "--las", "G.las", "-x", "G", "-o", "G", "--config", "nominal.ini"]
mx = [B + "get_maximal_reads", "--db", "G", "--las", "G.las", "-x", "G", "--config", "nominal.ini"]
T = {"HINGE_HOST_TIMING": "1"}
for argv in (flt, mx, lay):
    for rep in range(2):
        p = subprocess.run(argv, cwd=wd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, **T)); print("\n".join(l for l in p.stderr.decode().splitlines() if l.startswith("[timing]") and "las.load" not in l))
sys.exit(0)
