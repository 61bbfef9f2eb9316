#!/usr/bin/env python
"""BASELINE config 5, one rank's share AT SIZE (52 Mb at 100x: ~7.4e5 reads, >= 1.25e8 overlaps) through the CPU oracle's
`hinge filter`, once, in the build container -> tests/golden/cfg5_share_digest.json: sha256 of the mask / cmask rows, the
repeat annotations and the hinge rows, plus a digest of the input.  tests/test_cfg5_gpu.py regenerates the same pile-ups
(hinge_amd/synth_device.py on the CPU generator, whose stream does not depend on the machine), runs the HIP pass on them and
compares - the oracle comparison at size without the oracle's minutes (and its ~16 GB .las) on the GPU box.

    python tests/golden/make_cfg5_digest.py          (~20 min, ~40 GB of RAM, 17 GB of scratch disk)
"""
import dataclasses
import hashlib
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.join(HERE, "cfg5_share_digest.json")
GENOME = 52_000_000


def sha_arr(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def input_digest(p):
    return sha_arr(p.rlen.astype(np.int32), p.row_ptr.cpu().numpy(), p.a_span.cpu().numpy(), p.b_span.cpu().numpy(), p.b_flag.cpu().numpy())


def result_digests(n, mask, cmask, off, pos, typ, ish):
    """mask / cmask: int32 [n, 2]; annotations of read i: pos / typ[off[i]:off[i + 1]]; ish: which of them are hinges.  The hinge
    rows stop before the part's last read (filter.cpp:1091)."""
    cnt = np.diff(off).astype(np.int32)
    owner = np.repeat(np.arange(n, dtype=np.int32), cnt)
    keep = (np.asarray(ish) != 0) & (owner < n - 1)
    return {"mask": sha_arr(np.asarray(mask, np.int32)), "cmask": sha_arr(np.asarray(cmask, np.int32)),
            "repeat": sha_arr(cnt, np.asarray(pos, np.int32), np.asarray(typ, np.int32)),
            "hinges": sha_arr(owner[keep], np.asarray(pos, np.int32)[keep], np.asarray(typ, np.int32)[keep]),
            "n_annotations": int(cnt.sum()), "n_hinges": int(keep.sum())}


def parse_rows(path, n):
    """`read v v v ...` lines -> (off, flat values)"""
    off = np.zeros(n + 1, np.int64)
    vals = []
    for line in open(path):
        t = line.split()
        if not t:
            continue
        i = int(t[0])
        v = [int(x) for x in t[1:]]
        off[i + 1] = len(v)
        vals.extend(v)
    return np.cumsum(off), np.asarray(vals, np.int64)


def main():
    import conftest
    import oracle
    import torch
    from hinge_amd import capi, synth, synth_device
    lib = oracle.oracle_lib()
    spec = dataclasses.replace(synth.CONFIGS["cfg5_share"], genome_len=GENOME)
    t0 = time.time()
    p = synth_device.generate_pileups(spec, "cpu", span16_pad=0)
    n, m = p.n_reads, p.n_ovl
    print("generated", n, m, round(time.time() - t0, 1), flush=True)
    dig = {"genome": GENOME, "reads": int(n), "overlaps": int(m), "input_sha256": input_digest(p), "torch": torch.__version__}
    d = synth_device.extract_block(p, 0, n)
    del p
    tmp = tempfile.mkdtemp(prefix="hinge_cfg5_", dir=os.environ.get("HINGE_SCRATCH", "/tmp"))
    try:
        synth.write_dataset(d, tmp, "G", write_bases=False)
        del d
        conftest.write_ini(os.path.join(tmp, "nominal.ini"))
        print("written", round(time.time() - t0, 1), flush=True)
        rc = conftest.run_in(tmp, lib.oracle_filter, b"G", b"G.las", 0, b"O", b"nominal.ini", b"")
        assert rc == 0, rc
        print("oracle done", round(time.time() - t0, 1), flush=True)
        mas = np.loadtxt(os.path.join(tmp, "O.mas"), dtype=np.int64)
        cmas = np.loadtxt(os.path.join(tmp, "O.cmas"), dtype=np.int64)
        assert mas.shape == (n, 3) and np.array_equal(mas[:, 0], np.arange(n))
        off, flat = parse_rows(os.path.join(tmp, "O.repeat.txt"), n)
        pos, typ = flat[0::2], flat[1::2]
        hoff, hflat = parse_rows(os.path.join(tmp, "O.hinges.txt"), n)
        # is-hinge flags from the hinge rows: the hinges of a read are a subsequence of its annotations
        ish = np.zeros(len(pos), np.uint8)
        hp, ht = hflat[0::2], hflat[1::2]
        hoff2 = hoff // 2
        off2 = off // 2
        for i in np.nonzero(np.diff(hoff2))[0]:
            a0, a1 = off2[i], off2[i + 1]
            k = a0
            for q in range(hoff2[i], hoff2[i + 1]):
                while not (pos[k] == hp[q] and typ[k] == ht[q]):
                    k += 1
                ish[k] = 1
                k += 1
            assert k <= a1
        dig.update(result_digests(n, mas[:, 1:], cmas[:, 1:], off2, pos, typ, ish))
        dig["oracle_s"] = round(time.time() - t0, 1)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    with open(OUT, "w") as f:
        json.dump(dig, f, indent=1, sort_keys=True)
    print(dig)


if __name__ == "__main__":
    main()
