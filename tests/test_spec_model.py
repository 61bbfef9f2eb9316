"""Three-way agreement for the stage loops the reference's own code cannot pin here: tests/spec_model.py (written from the prose
of SURVEY.md Appendix A) against the oracle's output files, on the data sets the GPU tests compare the kernels with the oracle on."""
import ctypes
import os

import numpy as np
import pytest

from conftest import clone_dataset, run_in, write_ini

import spec_model

ip = ctypes.POINTER(ctypes.c_int)


def _pairs(path):
    out = {}
    for line in open(path):
        tok = line.split()
        out[int(tok[0])] = [(int(tok[j]), int(tok[j + 1])) for j in range(1, len(tok) - 1, 2)]
    return out


@pytest.mark.parametrize("name,mlas,extra", [("tiny", False, ""), ("tiny_qv", False, ""), ("tiny_mlas", True, ""), ("ties", False, ""),
                                             ("chimera", False, ""), ("edges", False, ""),
                                             ("tiny_qv", False, "ec = 60\nhinge_min_support = 3\nhinge_unbridged = 2\nhinge_min_pileup = 3\nno_hinge_region = 200\n")])
def test_spec_model_agrees_with_the_oracle(datasets, oracle_lib, tmp_path, name, mlas, extra):
    from hinge_amd import formats
    from hinge_amd.config import IniFile, filter_params
    from hinge_amd.stages import las_list, qv_masks
    src, d = datasets(name)
    wd = clone_dataset(src, str(tmp_path / "oracle"))
    write_ini(os.path.join(wd, "v.ini"), extra_filter=extra)
    assert run_in(wd, oracle_lib.oracle_filter, b"G", b"G" if mlas else b"G.las", int(mlas), b"G", b"v.ini", b"") == 0

    def sort_perm(keys, descending):
        keys = np.ascontiguousarray(keys, np.int32)
        perm = np.zeros(max(len(keys), 1), np.int32)
        oracle_lib.oracle_sort_perm(len(keys), keys.ctypes.data_as(ip), 0 if descending else 1, perm.ctypes.data_as(ip))
        return perm[:len(keys)].tolist()

    qv = formats.read_qual_track(os.path.join(wd, "G"))
    fp = filter_params(IniFile(os.path.join(wd, "v.ini")), qv is not None)
    P = dict(CUT=fp.cut_off, EC=fp.est_cov, TH=fp.theta, CF=fp.coverage_fraction, MINRA=fp.min_repeat_annotation, MAXRA=fp.max_repeat_annotation,
             GAP=fp.repeat_annotation_gap, NHR=fp.no_hinge_region, SUP=fp.hinge_min_support, PIL=fp.hinge_bin_pileup, UNB=fp.hinge_unbridged,
             TOL=fp.hinge_tolerance, BIN=2 * fp.hinge_tolerance, USE_QV=fp.use_qv_mask, USE_COV=fp.use_coverage_mask)
    names = las_list(os.path.join(wd, "G") if mlas else os.path.join(wd, "G.las"), mlas)
    first = formats.read_las(names[0])
    qvm = qv_masks(qv, first.tspace) if qv is not None else None
    maskvec = {i: (0, 0) for i in range(d.n_reads)}
    min_cov = fp.min_cov
    want_mask = {int(a): (int(b), int(c)) for a, b, c in np.loadtxt(os.path.join(wd, "G.mas"), dtype=np.int64).reshape(-1, 3)}
    want_cmask = {int(a): (int(b), int(c)) for a, b, c in np.loadtxt(os.path.join(wd, "G.cmas"), dtype=np.int64).reshape(-1, 3)}
    want_rep, want_hg = _pairs(os.path.join(wd, "G.repeat.txt")), _pairs(os.path.join(wd, "G.hinges.txt"))
    want_cov = {}
    for line in open(os.path.join(wd, "G.coverage.txt")):
        tok = line.split()
        want_cov[int(tok[1])] = [int(t.split(",")[1]) for t in tok[2:]]
    n_hinges = 0
    for part, path in enumerate(names):
        recs = first if part == 0 else formats.read_las(path)
        pile = formats.pileups_from_las(recs, d.rlen)
        r_begin, r_end = int(recs.rec["aread"][0]), int(recs.rec["aread"][-1])
        min_cov, cmask, annos, hinges, cov0 = spec_model.filter_part(d.rlen, qvm, pile.row_ptr, pile.a_span, pile.b_span, pile.b_flag, r_begin, r_end,
                                                                     maskvec, min_cov, P, sort_perm)
        for i in range(r_begin, r_end + 1):
            assert maskvec[i] == want_mask[i], ("mask", i)
            assert cmask[i] == want_cmask[i], ("cmask", i)
            assert cov0[i].tolist() == want_cov[i], ("coverage", i)
            if part == 0:                       # .repeat.txt is closed after the first part (filter.cpp:1086)
                assert annos[i] == want_rep[i], ("annotations", i)
            if i < r_end:                       # .hinges.txt stops before the part's last read (filter.cpp:1091)
                assert hinges[i] == want_hg[i], ("hinges", i, hinges[i], want_hg[i])
                n_hinges += len(hinges[i])
    assert n_hinges > 0
