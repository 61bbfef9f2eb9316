"""GPU parity of the one-sweep pass (include/hinge_hip.h, hinge_filter_sweep*): K2 runs first with a PREDICTED MIN_COV and
yields the coverage sums; the exact median verifies; the guard-band reads run again with the exact value.  Whatever the
prediction says - right, off by one inside the band, far outside it - the stage's files are the CPU oracle's, byte for byte."""
import filecmp
import os

import numpy as np
import pytest

from conftest import clone_dataset, run_in, write_ini

pytestmark = pytest.mark.gpu

FILTER_FILES = [".mas", ".cmas", ".repeat.txt", ".hinges.txt", ".coverage.txt", ".cov.flag", ".self.flag"]


def _oracle_filter(lib, wd, mlas, ini="nominal.ini"):
    las = b"G" if mlas else b"G.las"
    return run_in(wd, lib.oracle_filter, b"G", las, 1 if mlas else 0, b"G", ini.encode(), b"")


def _hip_filter(wd, mlas, ctx, ini="nominal.ini"):
    from hinge_amd import stages
    return run_in(wd, stages.run_filter, "G", "G" if mlas else "G.las", "G", ini, mlas, 0, True, False, ctx, True)


def _compare(wd_o, wd_h):
    bad = [s for s in FILTER_FILES if not filecmp.cmp(os.path.join(wd_o, "G" + s), os.path.join(wd_h, "G" + s), shallow=False)]
    assert not bad, "differs from the oracle: %s" % bad


# (band, bias): the prediction is shifted by `bias`; |bias| <= band keeps the exact value inside the band (guard-band reads only),
# beyond it the verification orders the whole part again
KNOBS = [(1, 0), (1, 1), (1, -1), (0, 0), (0, 1), (2, -2), (1, 4), (1, -3), (3, 2)]


@pytest.mark.parametrize("name,mlas,general", [("tiny", False, 0), ("tiny_qv", False, 0), ("tiny_mlas", True, 0), ("chimera", False, 0), ("deep", False, 0),
                                               ("long_reads", False, 0), ("edges", False, 0), ("long_repeat", False, 1)])
def test_one_sweep_under_forced_mispredictions(datasets, oracle_lib, tmp_path, name, mlas, general):
    from hinge_amd import capi
    src, _ = datasets(name)
    wd_o = clone_dataset(src, str(tmp_path / "oracle"))
    assert _oracle_filter(oracle_lib, wd_o, mlas) == 0
    seen_off, seen_out, seen_guard = 0, 0, 0
    for band, bias in KNOBS:
        wd_h = clone_dataset(src, str(tmp_path / ("hip_%d_%d" % (band, bias + 10))))
        ctx = capi.Context(0)
        ctx.force_general_mask(general)
        ctx.debug_spec(band=band, sample=4096, bias=bias)
        assert _hip_filter(wd_h, mlas, ctx) == 0
        verified, off, outside, guard, pred, exact = ctx.spec_stats()
        assert verified >= 1, "the route of the executables must be the one-sweep pass"
        if abs(bias) > band:
            assert outside >= 1, "band %d, bias %d: the verification must have ordered the part again" % (band, bias)
        if bias != 0 and abs(bias) <= band:
            assert off >= 1
        seen_off += off
        seen_out += outside
        seen_guard += max(guard, 0)
        _compare(wd_o, wd_h)
        ctx.close()
    assert seen_off > 0 and seen_out > 0


@pytest.mark.parametrize("name", ["chimera", "long_reads"])
def test_one_sweep_equals_two_sweeps_table_by_table(datasets, name):
    """The library calls themselves: sweep (one-sweep pass) against stats + median + mask_annotate on the same pile-ups -
    estimate, totals (est.total_cov / est.num_slot), means, masks, coverage bins, annotations, work list size.  `long_reads` has
    well-formed reads too long for the fast kernel's LDS slots: the general kernel takes them inside the one-sweep pass and the
    median must count each of them once (round 4 counted them twice, from a sum nobody had stored)."""
    from hinge_amd import capi, formats
    from hinge_amd.config import default_filter_params
    src, d = datasets(name)
    idx = formats.read_db_index(os.path.join(src, "G"))
    rlen = idx["rlen"]
    recs = formats.read_las(os.path.join(src, "G.las"))
    pile = formats.pileups_from_las(recs, rlen)
    r0, r1 = int(recs.rec["aread"][0]), int(recs.rec["aread"][-1])
    P = default_filter_params()
    span16, max_pile, in_range = capi.pack_spans(pile.row_ptr, pile.a_span, rlen)
    res = []
    for one in (False, True):
        ctx = capi.Context(0)
        ctx.set_reads(rlen, None)
        ctx.set_min_cov(P.min_cov)
        ctx.set_pileups_packed(r0, r1, pile.row_ptr, pile.a_span, pile.b_span, pile.b_flag, span16, max_pile, in_range)
        ctx.coverage_out(True)
        if one:
            est = ctx.filter_sweep(P, fetch=True)
            assert ctx.spec_stats()[0] == 1
            if name == "long_reads":
                assert ctx.fallback_reads() > 0, "needs reads the fast kernel hands back"
        else:
            est = ctx.filter_stats_median(P, fetch=True)
            ctx.filter_mask_annotate(P)
        ctx.filter_hinges(P)
        mask, cmask, flags = ctx.get_masks()
        off, pos, typ, ish = ctx.get_annotations()
        nb, cov = ctx.get_coverage()
        res.append(dict(est=(est.cov_est, est.n_long, est.total_cov, est.num_slot), min_cov=ctx.get_min_cov(), mask=mask, cmask=cmask, flags=flags,
                        off=off, pos=pos, typ=typ, ish=ish, nb=nb, cov=cov, counters=ctx.counters()))
        ctx.close()
    a, b = res
    assert a["est"] == b["est"] and a["min_cov"] == b["min_cov"]
    for k in ("mask", "cmask", "flags", "off", "pos", "typ", "ish", "nb", "cov"):
        assert np.array_equal(a[k], b[k]), k
    assert tuple(a["counters"]) == tuple(b["counters"])
    assert len(a["pos"]) > 0 and int(np.sum(a["ish"])) > 0


def test_one_sweep_guard_band_is_small_and_not_empty(datasets):
    """On a chimera-rich data set some reads do have a bin inside the band: the guard-band list is exercised without any knob."""
    from hinge_amd import capi, formats
    from hinge_amd.config import default_filter_params
    src, d = datasets("chimera")
    rlen = formats.read_db_index(os.path.join(src, "G"))["rlen"]
    recs = formats.read_las(os.path.join(src, "G.las"))
    pile = formats.pileups_from_las(recs, rlen)
    r0, r1 = int(recs.rec["aread"][0]), int(recs.rec["aread"][-1])
    P = default_filter_params()
    ctx = capi.Context(0)
    ctx.set_reads(rlen, None)
    ctx.set_min_cov(P.min_cov)
    ctx.set_pileups(r0, r1, pile.row_ptr, pile.a_span, pile.b_span, pile.b_flag)
    guards = []
    for band in (0, 1, 3):
        ctx.debug_spec(band=band)
        ctx.set_min_cov(P.min_cov)
        ctx.filter_sweep(P)
        guards.append(ctx.spec_stats()[3])
    assert guards[0] == 0, "band 0: no read can be inside the band"
    assert 0 < guards[1] <= guards[2] < (r1 - r0 + 1) // 2, guards
    ctx.close()


def test_two_sweep_pass_behind_the_same_calls_with_delete_telomere(datasets, oracle_lib, tmp_path):
    """delete_telomere sums max(cov, MIN_COV) (filter.cpp:731-760): no band - the same two calls run the two-sweep pass."""
    from hinge_amd import capi
    src, _ = datasets("tiny")
    wd_o = clone_dataset(src, str(tmp_path / "oracle"))
    wd_h = clone_dataset(src, str(tmp_path / "hip"))
    for wd in (wd_o, wd_h):
        write_ini(os.path.join(wd, "nominal.ini"), extra_layout="del_telomere = 1")
    assert _oracle_filter(oracle_lib, wd_o, False) == 0
    ctx = capi.Context(0)
    assert _hip_filter(wd_h, False, ctx) == 0
    assert ctx.spec_stats()[3] == -1
    _compare(wd_o, wd_h)
    ctx.close()


def _ragged_parts(datasets):
    """The chimera data set cut into four parts of very different sizes (the last one a single read) + the tiny data set's
    reads behind them: what a batched sweep has to cope with."""
    from hinge_amd import capi, formats
    src, d = datasets("chimera")
    rlen = formats.read_db_index(os.path.join(src, "G"))["rlen"]
    recs = formats.read_las(os.path.join(src, "G.las"))
    pile = formats.pileups_from_las(recs, rlen)
    n = len(rlen)
    cuts = [0, n // 2, n // 2 + n // 3, n - 1, n]
    parts = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        s, e = int(pile.row_ptr[a]), int(pile.row_ptr[b])
        rp = np.ascontiguousarray(np.clip(pile.row_ptr, s, e) - s)       # (n_reads + 1 entries: empty rows outside the part)
        parts.append((a, b - 1, rp, pile.a_span[s:e].copy(), pile.b_span[s:e].copy(), pile.b_flag[s:e].copy()))
    return rlen, parts


@pytest.mark.parametrize("env", [{"HINGE_K2_BATCH": "0"}, {"HINGE_K2_BATCH": "1", "HINGE_K2_STEAL": "0"}, {"HINGE_K2_BATCH": "1", "HINGE_K2_STEAL": "1"},
                                 {"HINGE_K2_BATCH": "1", "HINGE_K2_STEAL": "2"}, {}, {"MIXED": "1"}])
def test_batched_sweep_is_the_per_part_sweep(datasets, monkeypatch, env):
    """hinge_filter_sweep_batch_async over ragged parts - one k_mask_annotate_q20_batch launch, its workgroups moving from part to
    part - against hinge_filter_sweep of every part on its own: estimate, MIN_COV, masks, bins, annotations, hinges; twice, the
    second time on the advanced item counters."""
    from hinge_amd import capi
    from hinge_amd.config import default_filter_params
    rlen, parts = _ragged_parts(datasets)
    P = default_filter_params()

    def fetch(ctx):
        mask, cmask, flags = ctx.get_masks()
        off, pos, typ, ish = ctx.get_annotations()
        nb, cov = ctx.get_coverage()
        return dict(min_cov=ctx.get_min_cov(), mask=mask, cmask=cmask, flags=flags, off=off, pos=pos, typ=typ, ish=ish, nb=nb, cov=cov, counters=tuple(ctx.counters()))

    mixed = env.pop("MIXED", None) is not None     # every other part with the ingest's 16|16 span copy: two kernel variants, so a launch per part

    def make(part, packed=False):
        r0, r1, rp, a, b, f = part
        ctx = capi.Context(0)
        ctx.set_reads(rlen, None)
        ctx.set_min_cov(P.min_cov)
        if packed:
            span16, max_pile, in_range = capi.pack_spans(rp, a, rlen)
            ctx.set_pileups_packed(r0, r1, rp, a, b, f, span16, max_pile, in_range)
        else:
            ctx.set_pileups(r0, r1, rp, a, b, f)
        ctx.coverage_out(True)
        return ctx

    want = []
    monkeypatch.setenv("HINGE_K2_BATCH", "0")
    for part in parts:
        ctx = make(part)
        ctx.filter_sweep(P)
        ctx.filter_hinges(P)
        want.append(fetch(ctx))
        ctx.close()
    monkeypatch.delenv("HINGE_K2_BATCH")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ctxs = [make(part, packed=mixed and k % 2 == 1) for k, part in enumerate(parts)]
    for _ in range(2):
        for c in ctxs:
            c.set_min_cov(P.min_cov)
        capi.sweep_batch_async(ctxs, P)
        capi.finish_batch_async(ctxs, P)
        capi.hinges_batch_async(ctxs, P)
        for c, w in zip(ctxs, want):
            c.check()
            got = fetch(c)
            assert got["min_cov"] == w["min_cov"] and got["counters"] == w["counters"]
            for k in ("mask", "cmask", "flags", "off", "pos", "typ", "ish", "nb", "cov"):
                assert np.array_equal(got[k], w[k]), k
    assert sum(len(w["pos"]) for w in want) > 0
    for c in ctxs:
        c.close()
