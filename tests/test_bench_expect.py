"""tools/make_bench_expect.py derives bench.py's expected results for N > 2 ranks from per-team oracle runs (the 8-block world
does not fit the oracle).  Here the derivation is checked against the oracle run on the WHOLE 4-block world of a small
workload - including a part where a team's own MIN_COV differs from the world's (the `ec` rerun) - and the data-set
construction of hinge_amd/benchsets.py against the merged world's records."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

INI = ("[filter]\nlength_threshold = 1000;\naln_threshold = 1000;\nmin_cov = 5;\ncut_off = 300;\ntheta = 300;\n"
       "[layout]\nhinge_slack = 1000\nmin_connected_component_size = 8\n")


def test_derived_world_equals_oracle_on_merged_world(oracle_lib):
    import make_bench_expect as mbe
    from hinge_amd import benchsets, synth
    base = synth.CONFIGS["chimera"]
    quiet = lambda *a: None
    reran = 0
    for p in (0, 1):
        team = mbe.team_runs(oracle_lib, base, 2, p, INI, quiet)
        for N in (2, 4):
            got = mbe.world_entries(oracle_lib, team, N, INI, 5, quiet)
            reran += sum(1 for t in team for k in t if k.startswith("rows_ec_"))
            d = mbe.merged_world(team, N)
            rows, means, est = mbe.run_oracle(oracle_lib, d, INI)
            assert est == got["cov_est"]
            want = [mbe.entry(rows, d.block_first[r], d.block_first[r + 1]) for r in range(N)]
            assert want == got["ranks"], (p, N)
            assert sum(e["hinges"] for e in want) > 100
    assert reran > 0, "no case exercised the `ec` rerun"


def test_rank_part_matches_the_merged_world():
    """What a rank builds for itself (benchsets.rank_part) names the same overlaps as the merged world's records of its block."""
    import make_bench_expect as mbe
    from hinge_amd import benchsets, synth
    base = synth.CONFIGS["tiny"]
    N, p = 4, 1
    team = [{"d": synth.generate(benchsets.part_spec(base, N, 2 * q, p)[0])} for q in range(N // 2)]
    d = mbe.merged_world(team, N)
    for rank in range(N):
        rp = benchsets.rank_part(base, N, rank, p)
        lo, hi = d.block_first[rank], d.block_first[rank + 1]
        sel = (d.aread >= lo) & (d.aread < hi)
        assert rp.n_records == int(sel.sum()) and np.array_equal(rp.rlen, d.rlen[lo:hi])
        keep = sel & (d.aread != d.bread)
        assert rp.n_ovl == int(keep.sum())
        a_of = np.repeat(np.arange(rp.n_reads), np.diff(rp.row_ptr))
        assert np.array_equal(a_of + lo, d.aread[keep])
        first = np.asarray(d.block_first)
        assert np.array_equal(first[rp.b_owner] + rp.b_local, d.bread[keep])
        assert np.array_equal(rp.a_span[:, 0], d.ab[keep]) and np.array_equal(rp.b_span[:, 1], d.be[keep]) and np.array_equal(rp.comp, d.comp[keep])
        assert rp.last_a + lo == d.aread[sel][-1]
        assert len(np.unique(rp.b_owner)) == 2 and set(np.unique(rp.b_owner)) == {rank // 2 * 2, rank // 2 * 2 + 1}
