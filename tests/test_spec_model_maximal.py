"""tests/spec_model_maximal.py (a second reading of maximal.cpp's main body) against the oracle's .max / .contained.txt."""
import ctypes
import os

import numpy as np
import pytest

from conftest import clone_dataset, run_in

import spec_model_maximal


def _primitives(lib):
    ip, u16p = ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint16)

    def sort_perm(keys):
        k = np.ascontiguousarray(keys, dtype=np.int32)
        out = np.zeros(max(len(k), 1), np.int32)
        lib.oracle_sort_perm(len(k), k.ctypes.data_as(ip), 0, out.ctypes.data_as(ip))
        return out[:len(k)].tolist()

    lib.oracle_umap_order.argtypes = [ctypes.c_int, ip, ip]
    lib.oracle_umap_order.restype = ctypes.c_int

    def umap_order(keys):
        k = np.ascontiguousarray(keys, dtype=np.int32)
        out = np.zeros(max(len(k), 1), np.int32)
        n = lib.oracle_umap_order(len(k), k.ctypes.data_as(ip), out.ctypes.data_as(ip))
        return out[:n].tolist()

    def process_alignment(ab, ae, bb, be, comp, eff_a, eff_b, trace, aln_threshold, theta, theta2, trim):
        assert trim
        hdr = np.array([ab, ae, bb, be, comp, eff_a[0], eff_a[1], eff_b[0], eff_b[1]], np.int32)
        tr = np.ascontiguousarray(trace, dtype=np.uint16)
        out = np.zeros(10, np.int32)
        lib.oracle_process_alignment(hdr.ctypes.data_as(ip), tr.ctypes.data_as(u16p), len(tr), aln_threshold, theta, theta2, out.ctypes.data_as(ip))
        return {"type": int(out[4]), "active": bool(out[5])}

    return process_alignment, sort_perm, umap_order


def _part(recs, rlen):
    from hinge_amd import formats
    r = recs.rec
    comp = (r["flags"] & 1).astype(np.int32)
    blen = rlen[r["bread"]]
    bb = np.where(comp == 1, blen - r["bepos"], r["bbpos"])            # getOverlap's flip to B's forward strand (LAInterface.cpp:1619-1626)
    be = np.where(comp == 1, blen - r["bbpos"], r["bepos"])
    tb = 1 if recs.tspace <= formats.TRACE_XOVR else 2
    trace = []
    for k in range(recs.novl):
        raw = recs.trace[recs.trace_off[k]:recs.trace_off[k + 1]]
        trace.append(raw.astype(np.uint16) if tb == 1 else raw.view("<u2").astype(np.uint16))
    return {"aread": r["aread"], "bread": r["bread"], "comp": comp, "ab": r["abpos"], "ae": r["aepos"], "bb": bb, "be": be, "trace": trace}


@pytest.mark.parametrize("name,mlas", [("tiny", False), ("tiny_mlas", True), ("ties", False), ("chimera", False), ("edges", False), ("tspace200", False)])
def test_maximal_model_agrees_with_the_oracle(datasets, oracle_lib, tmp_path, name, mlas):
    from hinge_amd import formats
    src, d = datasets(name)
    wd = clone_dataset(src, str(tmp_path / "o"))
    las = b"G" if mlas else b"G.las"
    assert run_in(wd, oracle_lib.oracle_filter, b"G", las, int(mlas), b"G", b"nominal.ini", b"") == 0
    assert run_in(wd, oracle_lib.oracle_maximal, b"G", las, int(mlas), b"G", b"nominal.ini") == 0
    eff = np.zeros((d.n_reads, 2), np.int64)
    for line in open(os.path.join(wd, "G.mas")):
        i, s, e = (int(t) for t in line.split())
        eff[i] = (s, e)
    names = [os.path.join(wd, "G.%d.las" % (k + 1)) for k in range(d.spec.n_blocks)] if mlas else [os.path.join(wd, "G.las")]
    parts = [_part(formats.read_las(p), d.rlen) for p in names]
    got_max, got_contained = spec_model_maximal.maximal(d.rlen, eff.tolist(), parts, 1000, 1000, 300, 0, True, True, *_primitives(oracle_lib))
    want_max = open(os.path.join(wd, "G.max")).read().split("\n")[:-1]
    want_contained = open(os.path.join(wd, "G.contained.txt")).read().split("\n")[:-1]
    assert len(want_contained) > 10 and 0 < len(want_max) < d.n_reads
    assert got_max == want_max
    assert got_contained == want_contained
