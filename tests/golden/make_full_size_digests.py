#!/usr/bin/env python
"""sha256 digests of the CPU oracle's stage files on BASELINE configs 2 - 4 AT THEIR OWN SIZE -> tests/golden/full_size_digests.json.

Run in the build container (minutes of single-thread oracle per configuration); the GPU tests (tests/test_full_size_gpu.py)
regenerate the same seeded data sets, run the three executables and compare their files with these digests - the full-size
comparison without the oracle's minutes on the GPU box.  The inputs are digested too (lengths, records): a generator that
drifts makes the test say so instead of comparing apples with pears.

Like tests/golden/stage_hashes.json these freeze the ORACLE's files, and the oracle's three main() bodies are parity
unpinned: the digests carry the full-size comparison to the driver's run, they do not pin anything new.

    python tests/golden/make_full_size_digests.py [cfg3_nctc cfg4_yeast cfg2_ecoli160]
"""
import hashlib
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "full_size_digests.json")


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def input_digest(d):
    """What identifies the generated data set: read lengths and every overlap's coordinates (not the 3 GB of .las bytes)."""
    import numpy as np
    h = hashlib.sha256()
    for a in (d.rlen, d.aread, d.bread, d.comp, d.ab, d.ae, d.bb, d.be):
        h.update(np.ascontiguousarray(a).astype(np.int64).tobytes())
    return h.hexdigest()


def run(name):
    import conftest
    import fuzz_pipeline
    import oracle
    from hinge_amd import synth
    lib = oracle.oracle_lib()
    spec = synth.CONFIGS[name]
    d = synth.generate(spec)
    mlas = spec.n_blocks > 1
    tmp = tempfile.mkdtemp(prefix="hinge_digest_")
    try:
        synth.write_dataset(d, tmp, "G", write_bases=False)
        conftest.write_ini(os.path.join(tmp, "v.ini"))
        las = b"G" if mlas else b"G.las"
        t0 = time.time()
        rcs = [conftest.run_in(tmp, lib.oracle_filter, b"G", las, int(mlas), b"G", b"v.ini", b""),
               conftest.run_in(tmp, lib.oracle_maximal, b"G", las, int(mlas), b"G", b"v.ini"),
               conftest.run_in(tmp, lib.oracle_layout, b"G", las, int(mlas), b"G", b"G", b"v.ini")]
        assert rcs == [0, 0, 0], rcs
        hinges = sum((len(l.split()) - 1) // 2 for l in open(os.path.join(tmp, "G.hinges.txt")))
        return {"reads": int(d.n_reads), "records": int(d.novl), "blocks": int(spec.n_blocks), "hinges": hinges, "input_sha256": input_digest(d),
                "oracle_s": round(time.time() - t0, 1), "sha256": {f: sha(os.path.join(tmp, f)) for f in fuzz_pipeline.FILES}}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    names = sys.argv[1:] or ["cfg3_nctc", "cfg4_yeast", "cfg2_ecoli160"]
    out = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for name in names:
        out[name] = run(name)
        print(name, {k: v for k, v in out[name].items() if k != "sha256"}, flush=True)
        with open(OUT, "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
