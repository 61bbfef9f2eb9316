"""GPU parity of the trim / classify kernels (ProcessAlignment, GetMatchingPosition) against the CPU
oracle (which is itself pinned to the reference's compiled LOverlap methods), overlap by overlap."""
import ctypes
import os

import numpy as np
import pytest

from conftest import clone_dataset, run_in

pytestmark = pytest.mark.gpu
ip = ctypes.POINTER(ctypes.c_int)
u16p = ctypes.POINTER(ctypes.c_uint16)


def _setup(datasets, oracle_lib, tmp_path, name):
    from hinge_amd import capi, formats
    src, d = datasets(name)
    wd = clone_dataset(src, str(tmp_path / "w"))
    assert run_in(wd, oracle_lib.oracle_filter, b"G", b"G.las", 0, b"G", b"nominal.ini", b"") == 0
    eff = np.loadtxt(os.path.join(wd, "G.mas"), dtype=np.int64)[:, 1:].astype(np.int32)
    recs = formats.read_las(os.path.join(wd, "G.las"))
    pile = formats.pileups_from_las(recs, d.rlen)
    ctx = capi.Context(0)
    ctx.set_reads(d.rlen, None)
    ctx.set_pileups(0, d.n_reads - 1, pile.row_ptr, pile.a_span, pile.b_span, pile.b_flag)
    toff = recs.trace_off[:-1][pile.las_index]
    tlen = recs.rec["tlen"][pile.las_index]
    ctx.set_traces(recs.trace, toff, tlen, 1 if recs.tspace <= 125 else 2)
    ctx.set_eff_reads(eff)
    a_of = np.repeat(np.arange(d.n_reads, dtype=np.int32), np.diff(pile.row_ptr).astype(np.int64))
    return ctx, recs, pile, eff, a_of, toff, tlen


@pytest.mark.parametrize("name", ["tiny", "chimera"])
@pytest.mark.parametrize("thr", [(1000, 300, 0), (2500, 50, 100)])
def test_trim_classify_matches_oracle(datasets, oracle_lib, tmp_path, name, thr):
    ctx, recs, pile, eff, a_of, toff, tlen = _setup(datasets, oracle_lib, tmp_path, name)
    n = pile.n_ovl
    sel = np.arange(n, dtype=np.int64)
    got = ctx.trim_classify(sel, a_of, *thr)
    rng = np.random.default_rng(0)
    check = np.concatenate([np.arange(min(n, 3000)), rng.integers(0, n, size=6000)])
    types = set()
    for k in check:
        b = int(pile.b_flag[k] & 0x7FFFFFFF)
        comp = int(pile.b_flag[k] >> 31)
        a = int(a_of[k])
        hdr = np.array([pile.a_span[k, 0], pile.a_span[k, 1], pile.b_span[k, 0], pile.b_span[k, 1], comp,
                        eff[a, 0], eff[a, 1], eff[b, 0], eff[b, 1]], np.int32)
        tr = recs.trace[toff[k]:toff[k] + tlen[k]].astype(np.uint16)
        want = np.zeros(10, np.int32)
        oracle_lib.oracle_process_alignment(hdr.ctypes.data_as(ip), tr.ctypes.data_as(u16p), len(tr), thr[0], thr[1], thr[2], want.ctypes.data_as(ip))
        assert np.array_equal(got[k], want), (k, got[k], want)
        types.add(int(want[4]))
    assert len(types) >= 5, types
    ctx.close()


@pytest.mark.parametrize("name", ["tiny", "chimera", "tspace200"])
def test_trim_classify_part_equals_the_list_form(datasets, oracle_lib, tmp_path, name):
    """hinge_trim_classify_part (every overlap of the part, one wavefront per A read, coalesced) == hinge_trim_classify /
    hinge_trim_classify_types over the list of all overlaps (which the test above holds against the oracle)."""
    ctx, recs, pile, eff, a_of, toff, tlen = _setup(datasets, oracle_lib, tmp_path, name)
    n = pile.n_ovl
    sel = np.arange(n, dtype=np.int64)
    thr = (1000, 300, 0)
    full = ctx.trim_classify(sel, a_of, *thr)
    types = ctx.trim_classify_types(sel, a_of, *thr)
    part = ctx.trim_classify_part(n, *thr)                 # the streaming kernel (one lane per overlap over an LDS-staged .las)
    part_full = ctx.trim_classify_part_full(n, *thr)
    os.environ["HINGE_K4_ROWS"] = "1"                       # the eight-lanes-per-overlap kernel
    try:
        part_rows = ctx.trim_classify_part(n, *thr)
    finally:
        del os.environ["HINGE_K4_ROWS"]
    assert np.array_equal(types, full[:, 4].astype(np.uint8))
    assert np.array_equal(part, types)
    assert np.array_equal(part_rows, types)
    assert np.array_equal(part_full, full), np.nonzero((part_full != full).any(axis=1))[0][:10]
    assert len(set(part.tolist())) >= 4
    ctx.close()


def _image_ctx(datasets, oracle_lib, tmp_path, name, eff_override=None):
    """The same part set up through hinge_set_las_image: no trace_off / tlen columns, the kernel reads the file's bytes."""
    from hinge_amd import capi, formats
    ctx0, recs, pile, eff, a_of, toff, tlen = _setup(datasets, oracle_lib, tmp_path, name)
    src, d = datasets(name)
    image = np.fromfile(os.path.join(str(tmp_path / "w"), "G.las"), dtype=np.uint8)
    row_base, rec_rel = formats.las_image_table(recs, pile)
    ctx = capi.Context(0)
    ctx.set_reads(d.rlen, None)
    ctx.set_pileups(0, d.n_reads - 1, pile.row_ptr, pile.a_span, pile.b_span, pile.b_flag)
    ctx.set_las_image(image, row_base, rec_rel, 1 if recs.tspace <= 125 else 2)
    ctx.set_eff_reads(eff if eff_override is None else eff_override)
    return ctx0, ctx, pile, eff, a_of, tlen


@pytest.mark.parametrize("name", ["tiny", "chimera", "tspace200", "edges", "long_reads"])
def test_trim_classify_image_form_equals_the_list_form(datasets, oracle_lib, tmp_path, name):
    """k_trim_classify_image (hinge_set_las_image: tlen, spans, strand and B read from the staged .las records, strand flip in the
    kernel) == the list form over all overlaps (held against the oracle and the reference above), all ten fields; the data sets
    have self-overlap records between kept ones (stepped over), both strands, two-byte traces and traces that overflow the stage."""
    ctx0, ctx, pile, eff, a_of, tlen = _image_ctx(datasets, oracle_lib, tmp_path, name)
    n = pile.n_ovl
    sel = np.arange(n, dtype=np.int64)
    for thr in ((1000, 300, 0), (2500, 50, 100)):
        full = ctx0.trim_classify(sel, a_of, *thr)
        img_full = ctx.trim_classify_part_full(n, *thr)
        img_types = ctx.trim_classify_part(n, *thr)
        bad = np.nonzero((img_full != full).any(axis=1))[0]
        assert len(bad) == 0, (name, thr, len(bad), bad[:5], img_full[bad[:3]], full[bad[:3]])
        assert np.array_equal(img_types, full[:, 4].astype(np.uint8))
    assert (pile.b_flag >> 31).any() and not (pile.b_flag >> 31).all()
    # the list forms need hinge_set_traces: with an image only they refuse
    from hinge_amd import capi
    with pytest.raises(capi.HingeError):
        ctx.trim_classify(sel[:1], a_of[:1], 1000, 300, 0)
    ctx0.close(); ctx.close()


def test_trim_classify_image_form_survives_a_bad_offset_table(datasets, oracle_lib, tmp_path):
    """Entries of rec_rel that point outside the image (a caller's bug): nothing is read there - those overlaps come back inactive,
    every other overlap keeps its result (the overlap in front of a bad entry loses its end bound and walks global memory)."""
    from hinge_amd import capi, formats
    ctx0, recs, pile, eff, a_of, toff, tlen = _setup(datasets, oracle_lib, tmp_path, "chimera")
    src, d = datasets("chimera")
    image = np.fromfile(os.path.join(str(tmp_path / "w"), "G.las"), dtype=np.uint8)
    win_base, rec_rel = formats.las_image_table(recs, pile)
    n = pile.n_ovl
    rng = np.random.default_rng(3)
    bad = np.unique(rng.integers(0, n, size=40))
    bad = bad[bad % 64 != 0]                                   # (a window's first entry is win_base itself)
    rr = rec_rel.copy()
    rr[bad] = np.uint32(0xFFFFFF00)
    ctx = capi.Context(0)
    ctx.set_reads(d.rlen, None)
    ctx.set_pileups(0, d.n_reads - 1, pile.row_ptr, pile.a_span, pile.b_span, pile.b_flag)
    ctx.set_las_image(image, win_base, rr, 1)
    ctx.set_eff_reads(eff)
    full = ctx0.trim_classify(np.arange(n, dtype=np.int64), a_of, 1000, 300, 0)
    got = ctx.trim_classify_part_full(n, 1000, 300, 0)
    good = np.ones(n, bool)
    good[bad] = False
    assert np.array_equal(got[good], full[good]), np.nonzero((got != full).any(axis=1) & good)[0][:10]
    assert (got[bad, 4] == 6).all() and (got[bad, 5] == 0).all()      # MT_NOT_ACTIVE, inactive
    # a win_base that is not ascending is refused by the host check
    wb = win_base.copy()
    wb[1] = wb[0] - 2
    with pytest.raises(capi.HingeError):
        ctx.set_las_image(image, wb, rec_rel, 1)
    ctx0.close(); ctx.close()


def test_trim_classify_image_form_with_masks_that_cut_deep(datasets, oracle_lib, tmp_path):
    ctx0, ctx, pile, eff, a_of, tlen = _image_ctx(datasets, oracle_lib, tmp_path, "long_reads")
    n = pile.n_ovl
    rng = np.random.default_rng(11)
    eff2 = eff.copy()
    cut = rng.random(len(eff2)) < 0.5
    eff2[cut, 0] = (eff2[cut, 0] + rng.integers(0, 6000, size=int(cut.sum()))).astype(np.int32)
    eff2[cut, 1] = np.maximum(eff2[cut, 0], eff2[cut, 1] - rng.integers(0, 6000, size=int(cut.sum()))).astype(np.int32)
    ctx0.set_eff_reads(eff2); ctx.set_eff_reads(eff2)
    sel = np.arange(n, dtype=np.int64)
    full = ctx0.trim_classify(sel, a_of, 1000, 300, 0)
    img_full = ctx.trim_classify_part_full(n, 1000, 300, 0)
    assert np.array_equal(img_full, full), np.nonzero((img_full != full).any(axis=1))[0][:10]
    assert ((tlen.astype(np.int64) + 40) * 64 > 10240).any() and len(set(full[:, 4].tolist())) >= 5
    ctx0.close(); ctx.close()


def test_trim_classify_stream_with_masks_that_cut_deep(datasets, oracle_lib, tmp_path):
    """Masks that end in the middle of the reads (the walks of the streaming kernel then run far into the traces) and traces longer
    than the wavefront's stage buffer can hold 64 of: the stream form against the list form, all ten fields."""
    ctx, recs, pile, eff, a_of, toff, tlen = _setup(datasets, oracle_lib, tmp_path, "long_reads")
    n = pile.n_ovl
    rng = np.random.default_rng(7)
    rl = eff[:, 1].max()
    eff2 = eff.copy()
    cut = rng.random(len(eff2)) < 0.5
    eff2[cut, 0] = (eff2[cut, 0] + rng.integers(0, 6000, size=int(cut.sum()))).astype(np.int32)
    eff2[cut, 1] = np.maximum(eff2[cut, 0], eff2[cut, 1] - rng.integers(0, 6000, size=int(cut.sum()))).astype(np.int32)
    ctx.set_eff_reads(eff2)
    sel = np.arange(n, dtype=np.int64)
    thr = (1000, 300, 0)
    full = ctx.trim_classify(sel, a_of, *thr)
    part_full = ctx.trim_classify_part_full(n, *thr)
    assert np.array_equal(part_full, full), np.nonzero((part_full != full).any(axis=1))[0][:10]
    assert (tlen.astype(np.int64) * 64 > 10240).any()       # some steps cannot stage all 64 overlaps at once
    assert rl > 0 and len(set(full[:, 4].tolist())) >= 5
    ctx.close()


def test_matching_position_matches_oracle(datasets, oracle_lib, tmp_path):
    ctx, recs, pile, eff, a_of, toff, tlen = _setup(datasets, oracle_lib, tmp_path, "tiny")
    rng = np.random.default_rng(1)
    q = rng.integers(0, pile.n_ovl, size=4000).astype(np.int64)
    pos = (pile.a_span[q, 0] + rng.integers(-60, 60, size=len(q)) + (rng.random(len(q)) * (pile.a_span[q, 1] - pile.a_span[q, 0])).astype(np.int64)).astype(np.int32)
    got = ctx.matching_position(q, pos)
    for j, k in enumerate(q):
        tr = recs.trace[toff[k]:toff[k] + tlen[k]].astype(np.uint16)
        want = oracle_lib.oracle_matching_position(int(pile.a_span[k, 0]), int(pile.a_span[k, 1]), int(pile.b_span[k, 0]), int(pile.b_span[k, 1]),
                                                   int(pile.b_flag[k] >> 31), tr.ctypes.data_as(u16p), len(tr), int(pos[j]))
        assert got[j] == want, (k, pos[j], got[j], want)
    assert (got == -1).any() and (got >= 0).any()
    ctx.close()
