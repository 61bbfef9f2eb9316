#!/usr/bin/env python
"""Generates tests/golden/ref_vectors.npz from the REFERENCE's own compiled library code
(oracle/_ref/libhinge_ref.so, built from /root/reference/src/lib by oracle/Makefile).

Run in the build container (where /root/reference exists):  python tests/golden/make_golden.py
The vectors are data only: seeded random inputs and what the reference functions returned for them.
  profileCoverage                      src/lib/LAInterface.cpp:4298-4320
  trim_overlap + AddTypesAsymmetric    src/lib/LAInterface.cpp:4552-4683, 4721-4806 (through the
                                       ProcessAlignment body of src/maximal/maximal.cpp:65-134)
  GetMatchingPosition                  src/lib/LAInterface.cpp:4498-4546
  std::sort + compare_overlap / pairAscend / pairDescend / compare_overlap_weight
                                       src/lib/LAInterface.cpp:4875-4923
  getOverlap (record parse + strand flip) on a small synthetic .las
                                       src/lib/LAInterface.cpp:1519-1634
  INIReader on nominal.ini-style text  src/lib/INIReader.cpp, src/lib/ini.c
"""
import ctypes
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from hinge_amd import synth  # noqa: E402

ip = ctypes.POINTER(ctypes.c_int)
u16p = ctypes.POINTER(ctypes.c_uint16)
P = lambda a: a.ctypes.data_as(ip)  # noqa: E731

INI_TEXT = """[filter]
length_threshold = 1000;
quality_threshold = 0.23;
n_iter = 3; // filter iteration
aln_threshold = 1000;
min_cov = 5;
cut_off = 300;
theta = 300;
use_qv = true;
coverage = false
hexv = 0x10
octv = 010
cmt = 7 ; trailing comment
  continued
[running]
n_proc = 12;
[layout]
hinge_slack = 1000
del_telomere = 1
"""
INI_QUERIES = [("filter", "length_threshold", -1), ("filter", "n_iter", -1), ("filter", "min_cov", -1), ("filter", "hexv", -1),
               ("filter", "octv", -1), ("filter", "cmt", -1), ("filter", "missing", 42), ("FILTER", "CUT_OFF", 0),
               ("running", "n_proc", 4), ("layout", "hinge_slack", 7), ("layout", "del_telomere", 0), ("layout", "del_telomeres", 0)]
INI_BOOLS = [("filter", "use_qv", 0), ("filter", "use_qv", 1), ("filter", "coverage", 1), ("filter", "nokey", 1)]


def random_overlap(rng):
    """One overlap with a consistent trace (tspace 100) plus read masks."""
    alen, blen = int(rng.integers(2000, 12000)), int(rng.integers(2000, 12000))
    L = int(rng.integers(300, min(alen, blen)))
    ab = int(rng.integers(0, alen - L + 1))
    ae = ab + L
    d = int(rng.integers(0, 7))
    bL = L - d
    bb = int(rng.integers(0, blen - bL + 1))
    be = bb + bL
    comp = int(rng.integers(0, 2))
    nseg = (ae + 99) // 100 - ab // 100
    adv = np.full(nseg, 100, np.int64)
    base = (ab // 100) * 100
    adv[0] = (ae - ab) if nseg == 1 else base + 100 - ab
    if nseg > 1:
        adv[-1] = ae - (base + (nseg - 1) * 100)
    for j in range(min(d, nseg)):
        adv[j] -= 1
    if d > nseg:
        adv[-1] -= d - nseg
    adv = np.maximum(adv, 0)
    trace = np.zeros(2 * nseg, np.uint16)
    trace[0::2] = rng.integers(0, 20, size=nseg)
    trace[1::2] = adv
    a_es, a_ee = int(rng.integers(0, 900)), alen - int(rng.integers(0, 900))
    b_es, b_ee = int(rng.integers(0, 900)), blen - int(rng.integers(0, 900))
    if rng.random() < 0.15:
        a_es = int(rng.integers(0, alen))
    if rng.random() < 0.15:
        b_ee = int(rng.integers(0, blen))
    return np.array([ab, ae, bb, be, comp, a_es, a_ee, b_es, b_ee], np.int32), trace


def main():
    ref = oracle.ref_lib()
    assert ref is not None, "build oracle/_ref first (make -C oracle)"
    rng = np.random.default_rng(20260928)
    out = {}

    # profileCoverage
    cov_in, cov_out = [], []
    for case in range(60):
        n = int(rng.choice([0, 1, 2, 5, 40, 300]))
        ab = rng.integers(0, 9000, size=n).astype(np.int32)
        ae = (ab + rng.integers(50, 3000, size=n)).astype(np.int32)
        cutoff = int(rng.choice([0, 300, 700]))
        buf = np.zeros(4096, np.int32)
        K = ref.ref_profile_coverage(n, P(ab), P(ae), 40, cutoff, P(buf), 4096)
        cov_in.append(np.concatenate([[n, cutoff], ab, ae]).astype(np.int32))
        cov_out.append(np.concatenate([[K], buf[:K]]).astype(np.int32))
    out["cov_in"] = np.array(cov_in, dtype=object)
    out["cov_out"] = np.array(cov_out, dtype=object)

    # ProcessAlignment + GetMatchingPosition
    pa_in, pa_tr, pa_out, mp = [], [], [], []
    for case in range(400):
        hdr, trace = random_overlap(rng)
        res = np.zeros(10, np.int32)
        aln_thr = int(rng.choice([1000, 2500]))
        theta, theta2 = int(rng.choice([300, 50])), int(rng.choice([0, 100]))
        ref.ref_process_alignment(P(hdr), trace.ctypes.data_as(u16p), len(trace), aln_thr, theta, theta2, P(res))
        pa_in.append(np.concatenate([hdr, [aln_thr, theta, theta2]]).astype(np.int32))
        pa_tr.append(trace)
        pa_out.append(res)
        pos = rng.integers(hdr[0] - 50, hdr[1] + 50, size=6).astype(np.int32)
        got = [ref.ref_matching_position(int(hdr[0]), int(hdr[1]), int(hdr[2]), int(hdr[3]), int(hdr[4]),
                                         trace.ctypes.data_as(u16p), len(trace), int(x)) for x in pos]
        mp.append(np.concatenate([pos, got]).astype(np.int32))
    out["pa_in"] = np.array(pa_in)
    out["pa_trace"] = np.array(pa_tr, dtype=object)
    out["pa_out"] = np.array(pa_out)
    out["mp"] = np.array(mp)

    # std::sort permutations (mode: 0 compare_overlap, 1 pairAscend, 2 pairDescend, 3 compare_overlap_weight)
    sk, sp = [], []
    for case in range(80):
        n = int(rng.choice([0, 1, 15, 16, 17, 18, 40, 100, 257, 1000]))
        key = [rng.integers(0, 4, size=n), rng.integers(0, max(1, n // 6) + 1, size=n), np.sort(rng.integers(0, 30, size=n)),
               rng.integers(0, 1 << 20, size=n)][case % 4].astype(np.int32)
        for mode in range(4):
            perm = np.zeros(max(n, 1), np.int32)
            ref.ref_sort_perm(n, P(key), mode, P(perm))
            sk.append(np.concatenate([[mode], key]).astype(np.int32))
            sp.append(perm[:n].copy())
    out["sort_in"] = np.array(sk, dtype=object)
    out["sort_out"] = np.array(sp, dtype=object)

    # getOverlap on a small synthetic DB/.las (inputs = the dataset the seeded generator writes)
    with tempfile.TemporaryDirectory() as tmp:
        d = synth.generate(synth.SynthSpec(genome_len=40_000, coverage=12, seed=99))
        db = synth.write_dataset(d, tmp, "G")
        buf = np.zeros((d.novl, 8), np.int32)
        n = ref.ref_load_las(db.encode(), (db + ".las").encode(), P(buf), d.novl)
        assert n == d.novl
        out["las_spec"] = np.array([40_000, 12, 99], np.int32)
        out["las_records"] = buf
        ini = os.path.join(tmp, "q.ini")
        with open(ini, "w") as f:
            f.write(INI_TEXT)
        out["ini_int"] = np.array([ref.ref_ini_int(ini.encode(), s.encode(), k.encode(), dflt) for s, k, dflt in INI_QUERIES], np.int64)
        out["ini_bool"] = np.array([ref.ref_ini_bool(ini.encode(), s.encode(), k.encode(), dflt) for s, k, dflt in INI_BOOLS], np.int32)
        out["ini_real"] = np.array([ref.ref_ini_real(ini.encode(), b"filter", b"quality_threshold", 0.0)])
        out["ini_error"] = np.array([ref.ref_ini_error(ini.encode()), ref.ref_ini_error(b"/nonexistent.ini")], np.int32)

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
