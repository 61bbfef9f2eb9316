"""Stage-level behaviour of the CPU oracle (filter -> maximal -> layout): frozen regression hashes and
the reference quirks that are part of "bit-exact" (SURVEY.md 7-3)."""
import json
import os
import sys

import pytest

from conftest import clone_dataset, run_in, write_ini

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_stage_golden as msg  # noqa: E402

HASHES = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stage_hashes.json")))


@pytest.mark.parametrize("name,mlas,extra", msg.CASES)
def test_stage_outputs_are_stable(oracle_lib, tmp_path, name, mlas, extra):
    rc, h = msg.run_case(oracle_lib, name, mlas, extra, str(tmp_path))
    want = HASHES["%s|mlas=%d|%s" % (name, int(mlas), extra.strip())]
    assert rc == want["rc"] == [0, 0, 0]
    assert h == want["sha256"]


def test_mlas_quirks(oracle_lib, datasets, tmp_path):
    """--mlas: .repeat.txt holds only part 1 (closed inside the part loop, filter.cpp:1086), .hinges.txt
    drops the last read of every part (filter.cpp:1091), .mas covers every read."""
    src, d = datasets("tiny_mlas")
    wd = clone_dataset(src, str(tmp_path / "w"))
    assert run_in(wd, oracle_lib.oracle_filter, b"G", b"G", 1, b"G", b"nominal.ini", b"") == 0
    first = d.block_first
    mas = [int(l.split()[0]) for l in open(os.path.join(wd, "G.mas"))]
    assert mas == list(range(d.n_reads))
    rep = [int(l.split()[0]) for l in open(os.path.join(wd, "G.repeat.txt"))]
    assert rep == list(range(first[0], first[1]))
    hg = [int(l.split()[0]) for l in open(os.path.join(wd, "G.hinges.txt"))]
    want = [i for k in range(len(first) - 1) for i in range(first[k], first[k + 1] - 1)]
    assert hg == want


def test_layout_uses_hinges(oracle_lib, datasets, tmp_path):
    """Repeats longer than the reads leave unbridged hinges: layout keeps them and emits hinged edges."""
    src, _ = datasets("long_repeat")
    wd = clone_dataset(src, str(tmp_path / "w"))
    assert run_in(wd, oracle_lib.oracle_filter, b"G", b"G.las", 0, b"G", b"nominal.ini", b"") == 0
    assert run_in(wd, oracle_lib.oracle_maximal, b"G", b"G.las", 0, b"G", b"nominal.ini") == 0
    assert run_in(wd, oracle_lib.oracle_layout, b"G", b"G.las", 0, b"G", b"G", b"nominal.ini") == 0
    assert os.path.getsize(os.path.join(wd, "G.hinge.list")) > 0
    edges = [l.split() for l in open(os.path.join(wd, "G.edges.hinges"))]
    assert len(edges) > 50 and all(len(e) == 18 for e in edges)
    assert any(e[5] == "1" for e in edges), "expected at least one hinged (internal) edge"


def test_undefined_inputs_are_reported(oracle_lib, tmp_path):
    """No read >= 5000 bp: the reference indexes an empty vector / divides by zero (filter.cpp:660-666)."""
    import numpy as np
    from hinge_amd import synth
    d = synth.generate(synth.SynthSpec(genome_len=30_000, coverage=25, len_min=1500, len_max=4000, seed=3))
    wd = str(tmp_path)
    synth.write_dataset(d, wd, "G")
    write_ini(os.path.join(wd, "nominal.ini"))
    assert int(np.max(d.rlen)) < 5000
    assert run_in(wd, oracle_lib.oracle_filter, b"G", b"G.las", 0, b"G", b"nominal.ini", b"") == -3
