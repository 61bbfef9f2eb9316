"""`hinge maximal` through the sharded driver (hinge_amd/dist.py: ShardedMaximal + HipMaximalBackend) on one GPU:
the block's best overlaps are trimmed and classified by k_trim_classify, the candidate rows resolved by
hinge_resolve_containment; the mask must equal the oracle's .max."""
import os

import numpy as np
import pytest
import torch

from conftest import clone_dataset, run_in

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["tiny", "chimera", "edges", "tspace200"])
def test_sharded_maximal_matches_oracle(datasets, oracle_lib, tmp_path, name):
    from hinge_amd import capi, formats
    from hinge_amd.dist import BlockTable, Exchange, HipMaximalBackend, ShardedMaximal
    src, d = datasets(name)
    wd = clone_dataset(src, str(tmp_path / "w"))
    assert run_in(wd, oracle_lib.oracle_filter, b"G", b"G.las", 0, b"G", b"nominal.ini", b"") == 0
    assert run_in(wd, oracle_lib.oracle_maximal, b"G", b"G.las", 0, b"G", b"nominal.ini") == 0
    want = np.zeros(d.n_reads, np.uint8)
    want[np.loadtxt(os.path.join(wd, "G.max"), dtype=np.int64)] = 1
    eff = np.loadtxt(os.path.join(wd, "G.mas"), dtype=np.int64)[:, 1:].astype(np.int32)
    recs = formats.read_las(os.path.join(wd, "G.las"))
    pile = formats.pileups_from_las(recs, d.rlen)
    toff = recs.trace_off[:-1][pile.las_index]
    tlen = recs.rec["tlen"][pile.las_index]
    tbytes = 1 if recs.tspace <= formats.TRACE_XOVR else 2
    ctx = capi.Context(0)
    be = HipMaximalBackend(ctx, d.rlen, eff, 0, d.n_reads - 1, pile.row_ptr, pile.a_span, pile.b_span, pile.b_flag, recs.trace, toff, tlen,
                           tbytes, 1000, 1000, 300, 0, True, torch.device("cuda:0"))
    active = ShardedMaximal(be, Exchange(BlockTable([0, d.n_reads]), torch.device("cuda:0"))).step()
    assert np.array_equal(active, want)
    assert 0 < int(active.sum()) < d.n_reads
    ctx.close()
