#!/usr/bin/env python
"""Regression fixtures for the oracle's STAGE-level outputs (filter -> maximal -> layout) on the seeded
synthetic data sets: sha256 of every output file -> tests/golden/stage_hashes.json.

These are NOT a parity pin: the reference's three main() programs cannot be built in this image
(spdlog / Boost.Graph absent), so nothing here came from the reference.  They only freeze the oracle's
current behaviour so that an accidental change to the restatement is noticed."""
import hashlib
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from conftest import write_ini  # noqa: E402
from hinge_amd import synth  # noqa: E402

CASES = [("tiny", False, ""), ("tiny_qv", False, ""), ("tiny_mlas", True, ""), ("tiny_mlas", False, ""), ("ties", False, ""),
         ("chimera", False, ""), ("long_repeat", False, ""), ("tiny", False, "min_connected_component_size = 2\n")]
FILES = [".mas", ".cmas", ".repeat.txt", ".hinges.txt", ".coverage.txt", ".max", ".contained.txt", ".edges.hinges", ".edges.hinges2",
         ".hinge.list", ".deadends.txt", ".hgraph", ".killed.hinges", ".garbage.txt", ".edges.greedy", ".edges.1", ".edges.2", ".edges.skipped"]


def run_case(lib, name, mlas, extra_layout, wd):
    d = synth.generate(synth.CONFIGS[name])
    synth.write_dataset(d, wd, "G")
    write_ini(os.path.join(wd, "nominal.ini"), extra_layout=extra_layout)
    cwd = os.getcwd()
    os.chdir(wd)
    try:
        las = b"G" if mlas else b"G.las"
        rc = [lib.oracle_filter(b"G", las, int(mlas), b"G", b"nominal.ini", b""),
              lib.oracle_maximal(b"G", las, int(mlas), b"G", b"nominal.ini"),
              lib.oracle_layout(b"G", las, int(mlas), b"G", b"G", b"nominal.ini")]
    finally:
        os.chdir(cwd)
    h = {}
    for f in FILES:
        p = os.path.join(wd, "G" + f)
        h[f] = hashlib.sha256(open(p, "rb").read()).hexdigest() if os.path.exists(p) else None
    return rc, h


def main():
    lib = oracle.oracle_lib()
    out = {}
    for name, mlas, extra in CASES:
        with tempfile.TemporaryDirectory() as wd:
            rc, h = run_case(lib, name, mlas, extra, wd)
        out["%s|mlas=%d|%s" % (name, int(mlas), extra.strip())] = {"rc": rc, "sha256": h}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stage_hashes.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    main()
