#!/usr/bin/env python
"""Wall clock and per-phase host times (HINGE_HOST_TIMING=1) of the three executables on the bench data set, WITHOUT the CPU
oracle (tools/e2e_bench.py and bench.py's e2e block compare with it; this is the quick loop for work on a stage's fixed
costs).  python tools/stage_times.py [--rounds 3] [--dir D] [--env K=V ...]   (through gpurun)"""
import argparse
import dataclasses
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
INI = "[filter]\nlength_threshold = 1000;\naln_threshold = 1000;\nmin_cov = 5;\ncut_off = 300;\ntheta = 300;\n[layout]\nhinge_slack = 1000\nmin_connected_component_size = 8\n"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2_ecoli160")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--dir", default="")
    ap.add_argument("--env", action="append", default=[])
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()
    from hinge_amd import synth
    tmp = args.dir or tempfile.mkdtemp(prefix="hinge_stage_")
    os.makedirs(tmp, exist_ok=True)
    if not os.path.exists(os.path.join(tmp, "G.las")):
        d = synth.generate(dataclasses.replace(synth.CONFIGS[args.workload], n_blocks=1))
        synth.write_dataset(d, tmp, "G", write_bases=False)
        open(os.path.join(tmp, "nominal.ini"), "w").write(INI)
    env = dict(os.environ, HINGE_HOST_TIMING="1")
    for kv in args.env:
        k, v = kv.split("=", 1)
        env[k] = v
    hinge = os.path.join(ROOT, "hinge_amd", "bin", "hinge")
    res = {"filter": [], "maximal": [], "layout": []}
    for r in range(args.rounds):
        for sub, extra in (("filter", []), ("maximal", []), ("layout", ["-o", "H"])):
            t0 = time.perf_counter()
            p = subprocess.run([hinge, sub, "--db", "G", "--las", "G.las", "-x", "H", "--config", "nominal.ini"] + extra, cwd=tmp, env=env,
                               stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            wall = time.perf_counter() - t0
            assert p.returncode == 0, p.stderr.decode()[-2000:]
            phases = {}
            for line in p.stderr.decode().splitlines():
                if line.startswith("[timing]"):
                    tok = line.split()
                    phases[" ".join(tok[2:-2])] = float(tok[-2])
            res[sub].append({"wall_s": round(wall, 3), "inside_ms": phases.get("TOTAL"), "phases": phases})
            if args.verbose or r == args.rounds - 1:
                print(sub, "wall %.3f s" % wall, " | ".join("%s %.0f" % kv for kv in phases.items()), flush=True)
    summary = {k: {"wall_s": [x["wall_s"] for x in v], "inside_ms": [x["inside_ms"] for x in v]} for k, v in res.items()}
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
