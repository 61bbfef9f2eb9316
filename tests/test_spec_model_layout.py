"""tests/spec_model_layout.py (a second reading of hinging.cpp's hinge bookkeeping) against the oracle's files."""
import ctypes
import os

import numpy as np
import pytest

from conftest import clone_dataset, run_in, write_ini

import spec_model_layout
from test_spec_model_maximal import _part, _primitives


def _layout_primitives(lib):
    ip, u16p = ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint16)
    _, sort_perm, umap_order = _primitives(lib)

    def process_alignment(ab, ae, bb, be, comp, eff_a, eff_b, trace, aln_threshold, theta, theta2, trim):
        hdr = np.array([ab, ae, bb, be, comp, eff_a[0], eff_a[1], eff_b[0], eff_b[1]], np.int32)
        tr = np.ascontiguousarray(trace, dtype=np.uint16)
        out = np.zeros(10, np.int32)
        lib.oracle_process_alignment(hdr.ctypes.data_as(ip), tr.ctypes.data_as(u16p), len(tr), aln_threshold, theta, theta2, out.ctypes.data_as(ip))
        return {"eff_ab": int(out[0]), "eff_ae": int(out[1]), "eff_bb": int(out[2]), "eff_be": int(out[3]), "type": int(out[4]), "active": bool(out[5]),
                "weight": int(out[6]), "length": int(out[7])}

    def matching_position(raw, comp, trace, pos):
        tr = np.ascontiguousarray(trace, dtype=np.uint16)
        return int(lib.oracle_matching_position(raw[0], raw[1], raw[2], raw[3], comp, tr.ctypes.data_as(u16p), len(tr), pos))

    return process_alignment, matching_position, sort_perm, umap_order


@pytest.mark.parametrize("name,mlas,layout_ini", [("tiny", False, ""), ("tiny_mlas", True, ""), ("ties", False, ""), ("chimera", False, ""),
                                                   ("long_repeat", False, "min_connected_component_size = 3\nmatching_hinge_slack = 400\nhinge_slack = 200\nhinge_tolerance = 300\n"),
                                                   ("tspace200", False, "kill_hinge_overlap = 100\nkill_hinge_internal = 10\nuse_two_matches = 0\n")])
def test_layout_hinge_bookkeeping_agrees_with_the_oracle(datasets, oracle_lib, tmp_path, name, mlas, layout_ini):
    from hinge_amd import formats
    from hinge_amd.config import IniFile
    src, d = datasets(name)
    wd = clone_dataset(src, str(tmp_path / "o"))
    write_ini(os.path.join(wd, "v.ini"), extra_layout=layout_ini)
    las = b"G" if mlas else b"G.las"
    assert run_in(wd, oracle_lib.oracle_filter, b"G", las, int(mlas), b"G", b"v.ini", b"") == 0
    assert run_in(wd, oracle_lib.oracle_maximal, b"G", las, int(mlas), b"G", b"v.ini") == 0
    assert run_in(wd, oracle_lib.oracle_layout, b"G", las, int(mlas), b"G", b"G", b"v.ini") == 0
    n = d.n_reads
    eff = [(0, 0)] * n
    for line in open(os.path.join(wd, "G.mas")):
        i, s, e = (int(t) for t in line.split())
        eff[i] = (s, e)
    maximal = [False] * n
    for line in open(os.path.join(wd, "G.max")):
        maximal[int(line)] = True
    repeats = spec_model_layout.parse_pairs(os.path.join(wd, "G.repeat.txt"), n)
    hinges = spec_model_layout.parse_pairs(os.path.join(wd, "G.hinges.txt"), n)
    ini = IniFile(os.path.join(wd, "v.ini"))
    P = {"length_threshold": ini.get_int("filter", "length_threshold", -1), "aln_threshold": ini.get_int("filter", "aln_threshold", -1),
         "theta": ini.get_int("filter", "theta", -1), "theta2": ini.get_int("filter", "theta2", 0),
         "kill_hinge_overlap": ini.get_int("layout", "kill_hinge_overlap", 300), "kill_hinge_internal": ini.get_int("layout", "kill_hinge_internal", 40),
         "matching_hinge_slack": ini.get_int("layout", "matching_hinge_slack", 200),
         "min_connected_component_size": ini.get_int("layout", "min_connected_component_size", 8),
         "use_two_matches": bool(ini.get_int("layout", "use_two_matches", 1)),
         "hinge_tolerance": ini.get_int("layout", "hinge_tolerance", 150), "hinge_slack": ini.get_int("layout", "hinge_slack", 1000)}
    names = [os.path.join(wd, "G.%d.las" % (k + 1)) for k in range(d.spec.n_blocks)] if mlas else [os.path.join(wd, "G.las")]
    parts = [_part(formats.read_las(p), d.rlen) for p in names]
    got = spec_model_layout.layout_hinges(n, eff, maximal, repeats, hinges, parts, P, *_layout_primitives(oracle_lib))
    nonempty = 0
    for suffix, lines in got.items():
        want = open(os.path.join(wd, "G" + suffix)).read().split("\n")[:-1]
        assert lines == want, "%s: %d lines, the oracle has %d; first difference at line %s" % (
            suffix, len(lines), len(want), next((k for k, (x, y) in enumerate(zip(lines, want)) if x != y), min(len(lines), len(want))))
        nonempty += len(want) > 0
    assert len(got[".hgraph"]) > 0 and len(got[".edges.hinges"]) > 0 and nonempty >= 5
