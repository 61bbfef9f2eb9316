"""The Python mirrors of `hinge maximal` / `hinge layout` (hinge_amd/stages.py run_maximal, run_layout: the single-rank case of
the sharded drivers in hinge_amd/dist.py) against the oracle's files, after the Python mirror of `hinge filter`."""
import filecmp
import os

import pytest

from conftest import clone_dataset, run_in

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,mlas", [("tiny", False), ("tiny_mlas", True), ("edges", False), ("tspace200", False), ("long_repeat", False)])
def test_python_stages_match_oracle(datasets, oracle_lib, tmp_path, name, mlas):
    from hinge_amd import stages
    src, d = datasets(name)
    wd_o = clone_dataset(src, str(tmp_path / "oracle"))
    wd_h = clone_dataset(src, str(tmp_path / "hip"))
    las = b"G" if mlas else b"G.las"
    assert run_in(wd_o, oracle_lib.oracle_filter, b"G", las, int(mlas), b"G", b"nominal.ini", b"") == 0
    assert run_in(wd_o, oracle_lib.oracle_maximal, b"G", las, int(mlas), b"G", b"nominal.ini") == 0
    assert run_in(wd_h, stages.run_filter, "G", "G" if mlas else "G.las", "G", "nominal.ini", mlas) == 0
    assert run_in(wd_h, stages.run_maximal, "G", "G" if mlas else "G.las", "G", "nominal.ini", mlas) == 0
    files = [".mas", ".hinges.txt", ".repeat.txt", ".max", ".contained.txt"]
    if not mlas:        # run_layout takes one merged .las (ShardedLayout over several blocks: tests/test_dist_gpu.py)
        assert run_in(wd_o, oracle_lib.oracle_layout, b"G", las, 0, b"G", b"O", b"nominal.ini") == 0
        assert run_in(wd_h, stages.run_layout, "G", "G.las", "G", "O", "nominal.ini") == 0
        files += ["G.garbage.txt", "G.killed.hinges", "O.hgraph", "O.hinge.list", "O.edges.hinges", "O.edges.hinges2", "O.edges.skipped", "O.deadends.txt"]
    for f in files:
        fn = f if f[0] != "." else "G" + f
        assert filecmp.cmp(os.path.join(wd_o, fn), os.path.join(wd_h, fn), shallow=False), fn
    assert os.path.getsize(os.path.join(wd_o, "G.contained.txt")) > 0
