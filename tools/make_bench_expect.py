#!/usr/bin/env python
"""Hinge counts of the CPU oracle on the read sets bench.py rotates through (tests/golden/bench_expect.json).
bench.py asserts the GPU pass against them.  Run here (CPU only): python tools/make_bench_expect.py [--parts 4]"""
import argparse
import dataclasses
import json
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2_ecoli160")
    ap.add_argument("--parts", type=int, default=4)
    args = ap.parse_args()
    import oracle
    from hinge_amd import synth
    from bench import INI, count_pairs
    lib = oracle.oracle_lib()
    path = os.path.join(ROOT, "tests", "golden", "bench_expect.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    base = synth.CONFIGS[args.workload]
    for p in range(args.parts):
        seed = base.seed + 17 * p
        d = synth.generate(dataclasses.replace(base, n_blocks=1, seed=seed))
        tmp = tempfile.mkdtemp(prefix="hinge_expect_")
        try:
            synth.write_dataset(d, tmp, "G", write_bases=False)
            open(os.path.join(tmp, "nominal.ini"), "w").write(INI)
            cwd = os.getcwd()
            os.chdir(tmp)
            try:
                rc = lib.oracle_filter(b"G", b"G.las", 0, b"G", b"nominal.ini", b"")
            finally:
                os.chdir(cwd)
            assert rc == 0, rc
            out.setdefault(args.workload, {})[str(seed)] = count_pairs(os.path.join(tmp, "G.hinges.txt"))
            print(seed, d.n_reads, d.novl, out[args.workload][str(seed)], flush=True)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        json.dump(out, open(path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
