#!/usr/bin/env python
"""End-to-end wall clock of `hinge filter` + `hinge maximal` + `hinge layout` (the C++ programs over the HIP
library, .las ingest and text output included) next to the single-thread CPU oracle on the same files.
Secondary measurement; bench.py stays the headline kernel-throughput bench."""
import argparse
import dataclasses
import filecmp
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=int, default=1_000_000)
    ap.add_argument("--workload", default="cfg2_ecoli160")
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--dir", default="", help="work in this directory and keep it (e.g. to profile the executables on the data set afterwards)")
    ap.add_argument("--exact-config", action="store_true", help="at the configuration's own genome size: its own repeat model too (the data set bench.py uses)")
    args = ap.parse_args()
    import oracle
    from hinge_amd import synth
    spec = dataclasses.replace(synth.CONFIGS[args.workload], genome_len=args.genome, n_repeat_families=1, repeat_copies=(3, 3), n_blocks=1)
    if args.exact_config and args.genome == synth.CONFIGS[args.workload].genome_len:
        spec = dataclasses.replace(synth.CONFIGS[args.workload], n_blocks=1)
    d = synth.generate(spec)
    tmp = args.dir or tempfile.mkdtemp(prefix="hinge_e2e_")
    if args.dir:
        os.makedirs(tmp, exist_ok=True)
        args.keep = True
    out = {"workload": "%s at G=%d" % (args.workload, args.genome), "reads": d.n_reads, "overlaps": d.novl}
    try:
        dirs = {}
        for side in ("oracle", "hip"):
            wd = os.path.join(tmp, side)
            if side == "oracle":
                synth.write_dataset(d, wd, "G", write_bases=False)
            else:
                os.makedirs(wd)
                for f in os.listdir(dirs["oracle"]):
                    os.link(os.path.join(dirs["oracle"], f), os.path.join(wd, f))
            with open(os.path.join(wd, "nominal.ini"), "w") as f:
                f.write("[filter]\nlength_threshold = 1000;\naln_threshold = 1000;\nmin_cov = 5;\ncut_off = 300;\ntheta = 300;\n[layout]\nhinge_slack = 1000\nmin_connected_component_size = 8\n")
            dirs[side] = wd
        out["las_bytes"] = os.path.getsize(os.path.join(dirs["oracle"], "G.las"))
        lib = oracle.oracle_lib()
        cwd = os.getcwd()
        os.chdir(dirs["oracle"])
        t = {}
        t0 = time.perf_counter(); rc = lib.oracle_filter(b"G", b"G.las", 0, b"G", b"nominal.ini", b""); t["filter"] = time.perf_counter() - t0; assert rc == 0
        t0 = time.perf_counter(); rc = lib.oracle_maximal(b"G", b"G.las", 0, b"G", b"nominal.ini"); t["maximal"] = time.perf_counter() - t0; assert rc == 0
        t0 = time.perf_counter(); rc = lib.oracle_layout(b"G", b"G.las", 0, b"G", b"G", b"nominal.ini"); t["layout"] = time.perf_counter() - t0; assert rc == 0
        os.chdir(cwd)
        out["cpu_oracle_s"] = t
        hinge = os.path.join(ROOT, "hinge_amd", "bin", "hinge")
        g = {}
        for sub, extra in (("filter", []), ("maximal", []), ("layout", ["-o", "G"])):
            t0 = time.perf_counter()
            r = subprocess.run([hinge, sub, "--db", "G", "--las", "G.las", "-x", "G", "--config", "nominal.ini"] + extra, cwd=dirs["hip"],
                               stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, HINGE_HOST_TIMING="1"))
            g[sub] = time.perf_counter() - t0
            assert r.returncode == 0, r.stderr.decode()[-1000:]
            sys.stderr.write("".join(l + "\n" for l in r.stderr.decode().splitlines() if l.startswith("[timing]")))
        out["gpu_cli_s"] = g
        same = all(filecmp.cmp(os.path.join(dirs["oracle"], "G" + s), os.path.join(dirs["hip"], "G" + s), shallow=False)
                   for s in (".mas", ".repeat.txt", ".hinges.txt", ".max", ".edges.hinges", ".hinge.list", ".deadends.txt", ".coverage.txt"))
        out["byte_identical"] = same
        out["speedup_filter_layout"] = (t["filter"] + t["layout"]) / (g["filter"] + g["layout"])
        out["speedup_all_three"] = sum(t.values()) / sum(g.values())
    finally:
        if not args.keep:
            shutil.rmtree(tmp, ignore_errors=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
