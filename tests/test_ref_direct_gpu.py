"""HIP kernels against what the REFERENCE ITSELF computed - no oracle in between.

Two forms:
 * golden: tests/golden/ref_vectors.npz (outputs of the reference's own profileCoverage, trim_overlap + AddTypesAsymmetric,
   GetMatchingPosition and std::sort comparators, captured from oracle/_ref by tests/golden/make_golden.py) fed straight to
   the C ABI: hinge_filter_coverage_bins, K2's stored bins, hinge_trim_classify (list form), hinge_trim_classify_part[_full]
   (stream form and, HINGE_K4_ROWS=1, the rows form), hinge_matching_position, hinge_sort_order_desc, hinge_debug_pileup_order;
 * live: where oracle/_ref/libhinge_ref.so exists (it ships to the GPU box) >= 10^5 seeded random cases per primitive -
   traces whose per-segment B advance is uniform in [tspace - 15 %, tspace + 15 %] (real PacBio traces vary like that), one-
   and two-byte traces, masks that cut at every trace-point phase, both strands - HIP vs ref_process_alignment /
   ref_matching_position / ref_profile_coverage directly.
Reference lines: LAInterface.cpp:4298-4320 (profileCoverage), :4498-4546 (GetMatchingPosition), :4552-4683 (trim_overlap),
:4721-4806 (AddTypesAsymmetric), :4875-4923 (comparators); ProcessAlignment maximal.cpp:65-134."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ip = ctypes.POINTER(ctypes.c_int)
lp = ctypes.POINTER(ctypes.c_long)
u16p = ctypes.POINTER(ctypes.c_uint16)
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_vectors.npz"), allow_pickle=True)


# ---- one part out of a list of independent overlaps ------------------------------------------------------------------------
class Part:
    """n overlaps with their own masks laid out as a part the library can take: every overlap gets its own B read; `group`
    consecutive overlaps share an A read only when their A masks are equal (group_of gives the A read per overlap)."""

    def __init__(self, hdr, traces, tbytes, a_group=None):
        from hinge_amd import capi
        hdr = np.ascontiguousarray(hdr, np.int32)
        n = len(hdr)
        if a_group is None:
            a_group = np.arange(n)
        na = int(a_group.max()) + 1 if n else 0
        self.n, self.na = n, na
        self.a_of = a_group.astype(np.int32)
        n_reads = na + n
        rlen = np.full(n_reads, 1, np.int32)
        eff = np.zeros((n_reads, 2), np.int32)
        np.maximum.at(rlen, self.a_of, hdr[:, 1] + 1)
        rlen[na:] = hdr[:, 3] + 1
        eff[self.a_of] = hdr[:, 5:7]
        eff[na:] = hdr[:, 7:9]
        rlen = np.maximum(rlen, 1)
        counts = np.bincount(self.a_of, minlength=n_reads).astype(np.int64)
        row_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        b_flag = (np.arange(na, na + n, dtype=np.uint32) | (hdr[:, 4].astype(np.uint32) << np.uint32(31))).astype(np.uint32)
        tl = np.array([len(t) for t in traces], np.int64)
        toff_vals = np.concatenate([[0], np.cumsum(tl)]).astype(np.int64)
        flat = np.concatenate(traces).astype(np.uint16) if n else np.zeros(0, np.uint16)
        self.trace16, self.toff_vals = np.ascontiguousarray(flat), toff_vals
        if tbytes == 1:
            assert flat.max(initial=0) <= 255
            tr = flat.astype(np.uint8)
        else:
            tr = flat.astype("<u2").view(np.uint8)
        self.ctx = capi.Context(0)
        self.ctx.set_reads(rlen, None)
        self.ctx.set_pileups(0, n_reads - 1, row_ptr, np.ascontiguousarray(hdr[:, 0:2]), np.ascontiguousarray(hdr[:, 2:4]), b_flag)
        self.ctx.set_traces(tr, toff_vals[:-1] * tbytes, tl.astype(np.int32), tbytes)
        self.ctx.set_eff_reads(eff)
        self.hdr = hdr
        # the same overlaps as a .las image (records as DALIGNER writes them: B coordinates of complemented overlaps in the
        # complemented frame) behind a second context: k_trim_classify_image reads nothing but these bytes and one offset each
        comp = hdr[:, 4].astype(np.int64)
        blen = rlen[na:].astype(np.int64)
        rec = np.zeros((n, 10), np.int32)
        rec[:, 0] = tl
        rec[:, 2] = hdr[:, 0]; rec[:, 4] = hdr[:, 1]
        rec[:, 3] = np.where(comp == 1, blen - hdr[:, 3], hdr[:, 2]); rec[:, 5] = np.where(comp == 1, blen - hdr[:, 2], hdr[:, 3])
        rec[:, 6] = comp; rec[:, 7] = self.a_of; rec[:, 8] = np.arange(na, na + n)
        size = 40 + tl * tbytes
        starts = 12 + np.concatenate([[0], np.cumsum(size)]).astype(np.int64)
        image = np.zeros(int(starts[-1]), np.uint8)
        if n:
            image[(starts[:-1, None] + np.arange(40)[None, :]).reshape(-1)] = rec.view(np.uint8).reshape(-1)
            owner = np.repeat(np.arange(n), tl * tbytes)
            within = np.arange(len(tr)) - np.repeat(toff_vals[:-1] * tbytes, tl * tbytes)
            image[starts[owner] + 40 + within] = tr
        from hinge_amd import formats
        win_base, rec_rel = formats.image_windows(starts[:-1], size)
        self.ctx_img = capi.Context(0)
        self.ctx_img.set_reads(rlen, None)
        self.ctx_img.set_pileups(0, n_reads - 1, row_ptr, np.ascontiguousarray(hdr[:, 0:2]), np.ascontiguousarray(hdr[:, 2:4]), b_flag)
        self.ctx_img.set_las_image(image, win_base, rec_rel, tbytes)
        self.ctx_img.set_eff_reads(eff)

    def all_forms(self, thr):
        """(list form, stream form, the types of the stream and the rows form): every kernel that computes ProcessAlignment."""
        sel = np.arange(self.n, dtype=np.int64)
        full = self.ctx.trim_classify(sel, self.a_of, *thr)
        types_list = self.ctx.trim_classify_types(sel, self.a_of, *thr)
        stream = self.ctx.trim_classify_part_full(self.n, *thr)
        types_stream = self.ctx.trim_classify_part(self.n, *thr)
        os.environ["HINGE_K4_ROWS"] = "1"
        try:
            types_rows = self.ctx.trim_classify_part(self.n, *thr)
        finally:
            del os.environ["HINGE_K4_ROWS"]
        self.image_full = self.ctx_img.trim_classify_part_full(self.n, *thr)
        self.image_types = self.ctx_img.trim_classify_part(self.n, *thr)
        return full, stream, types_list, types_stream, types_rows

    def close(self):
        self.ctx.close()
        self.ctx_img.close()


def _check_classify(part, want, thr, what):
    full, stream, t_list, t_stream, t_rows = part.all_forms(thr)
    for name, got in (("k_trim_classify", full), ("k_trim_classify_stream", stream), ("k_trim_classify_image", part.image_full)):
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert len(bad) == 0, (what, name, len(bad), bad[:5], got[bad[:3]], want[bad[:3]], part.hdr[bad[:3]])
    for name, got in (("types list", t_list), ("types stream", t_stream), ("k_trim_classify_rows", t_rows), ("types image", part.image_types)):
        bad = np.nonzero(got != want[:, 4].astype(np.uint8))[0]
        assert len(bad) == 0, (what, name, len(bad), bad[:5])


# ---- golden vectors -> the library ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("tbytes", [1, 2])
def test_golden_process_alignment_through_every_kernel(tbytes):
    """pa_in / pa_trace / pa_out: the reference's trim_overlap + AddTypesAsymmetric, case by case, through the list, stream
    and rows kernels (the vectors carry their own thresholds, so the cases are grouped by them)."""
    pa_in, pa_tr, pa_out = GOLD["pa_in"], GOLD["pa_trace"], GOLD["pa_out"]
    seen = set()
    for thr in sorted(set(map(tuple, pa_in[:, 9:12].tolist()))):
        m = np.nonzero((pa_in[:, 9:12] == np.array(thr)).all(axis=1))[0]
        part = Part(pa_in[m, :9], [pa_tr[i] for i in m], tbytes)
        _check_classify(part, pa_out[m], thr, "golden thr=%s" % (thr,))
        seen |= set(pa_out[m, 4].tolist())
        part.close()
    assert len(seen) >= 6, seen


def test_golden_matching_position():
    pa_in, pa_tr, mp = GOLD["pa_in"], GOLD["pa_trace"], GOLD["mp"]
    part = Part(pa_in[:, :9], list(pa_tr), 1)
    q = np.repeat(np.arange(len(pa_in), dtype=np.int64), 6)
    got = part.ctx.matching_position(q, mp[:, :6].reshape(-1))
    assert np.array_equal(got, mp[:, 6:].reshape(-1)), np.nonzero(got != mp[:, 6:].reshape(-1))[0][:10]
    assert (got == -1).any() and (got >= 0).any()
    part.close()


def _coverage_part(rows_ab, rows_ae, min_rlen=0):
    """One A read per pile-up (its own B reads behind them); returns the context and n."""
    from hinge_amd import capi
    n = len(rows_ab)
    cnt = np.array([len(x) for x in rows_ab], np.int64)
    tot = int(cnt.sum())
    ab = np.concatenate(rows_ab).astype(np.int32) if tot else np.zeros(0, np.int32)
    ae = np.concatenate(rows_ae).astype(np.int32) if tot else np.zeros(0, np.int32)
    n_reads = n + 1                                   # one shared B read
    rlen = np.full(n_reads, max(min_rlen, 1), np.int32)
    a_of = np.repeat(np.arange(n), cnt)
    if tot:
        np.maximum.at(rlen, a_of, np.maximum(ab, ae) + 1)
    row_ptr = np.concatenate([[0], np.cumsum(cnt), [tot]]).astype(np.int64)
    b_span = np.zeros((tot, 2), np.int32)
    b_span[:, 1] = 1
    b_flag = np.full(tot, n, np.uint32)
    ctx = capi.Context(0)
    ctx.set_reads(rlen, None)
    ctx.set_pileups(0, n_reads - 1, row_ptr, np.ascontiguousarray(np.stack([ab, ae], axis=1)), b_span, b_flag)
    return ctx, n, rlen


def test_golden_profile_coverage_bins():
    """cov_in / cov_out -> hinge_filter_coverage_bins (k_coverage_bins) at each case's own cut-off."""
    cin, cout = GOLD["cov_in"], GOLD["cov_out"]
    rows_ab = [c[2:2 + int(c[0])] for c in cin]
    rows_ae = [c[2 + int(c[0]):2 + 2 * int(c[0])] for c in cin]
    ctx, n, _ = _coverage_part(rows_ab, rows_ae)
    for i in range(n):
        nb, cov = ctx.coverage_bins(i, i, 40, int(cin[i][1]))
        assert nb[0] == cout[i][0] and np.array_equal(cov, cout[i][1:]), (i, int(cin[i][0]), int(cin[i][1]))
    ctx.close()


def test_golden_profile_coverage_stored_by_the_sweep():
    """The cut-off 0 cases of cov_in -> the bins K2 itself stores during the filter pass (hinge_filter_coverage_out: the
    .coverage.txt payload of the executables), one-sweep and two-sweep pass."""
    from hinge_amd import config
    cin, cout = GOLD["cov_in"], GOLD["cov_out"]
    keep = [i for i in range(len(cin)) if int(cin[i][1]) == 0]
    rows_ab = [cin[i][2:2 + int(cin[i][0])] for i in keep]
    rows_ae = [cin[i][2 + int(cin[i][0]):2 + 2 * int(cin[i][0])] for i in keep]
    assert len(keep) >= 10
    for one_sweep in ("1", "0"):
        os.environ["HINGE_ONE_SWEEP"] = one_sweep
        try:
            ctx, n, _ = _coverage_part(rows_ab, rows_ae, min_rlen=6000)   # the median needs reads >= 5000 (filter.cpp:646)
            p = config.default_filter_params()
            ctx.coverage_out(True)
            ctx.filter_sweep(p)
            nb, cov = ctx.get_coverage()
        finally:
            del os.environ["HINGE_ONE_SWEEP"]
        off = np.concatenate([[0], np.cumsum(nb.astype(np.int64))])
        for k, i in enumerate(keep):
            assert nb[k] == cout[i][0] and np.array_equal(cov[off[k]:off[k + 1]], cout[i][1:]), (one_sweep, k, i)
        ctx.close()


def test_golden_std_sort_orders():
    """sort_in / sort_out: where libstdc++'s std::sort leaves equal keys under the reference's comparators.  compare_overlap and
    compare_overlap_weight (modes 0, 3: descending) -> hinge_sort_order_desc (host) and the wavefront-parallel replay in LDS
    (hinge_debug_pileup_order, what k_hinge_call sorts pile-ups with); pairDescend (mode 2) is the same order on keys."""
    from hinge_amd import capi
    ctx = capi.Context(0)
    n_dev = 0
    for sin, sout in zip(GOLD["sort_in"], GOLD["sort_out"]):
        mode, key = int(sin[0]), np.ascontiguousarray(sin[1:], np.int32)
        if mode == 1:
            continue                                   # pairAscend: ascending - no descending entry point to hold against
        got = capi.sort_order_desc(key.astype(np.int64), 1)
        assert np.array_equal(got, sout), (mode, len(key))
        if len(key) <= 4096:
            pos = ctx.debug_pileup_order(key)
            perm = np.zeros(len(key), np.int32)
            perm[pos] = np.arange(len(key), dtype=np.int32)
            assert np.array_equal(perm, sout), (mode, len(key), "device replay")
            n_dev += 1
    assert n_dev >= 100
    ctx.close()


# ---- live reference library ------------------------------------------------------------------------------------------------
def _bind_batch(ref):
    c = ctypes
    ref.ref_process_alignment_batch.argtypes = [c.c_long, ip, u16p, lp, c.c_int, c.c_int, c.c_int, ip]
    ref.ref_process_alignment_batch.restype = None
    ref.ref_matching_position_batch.argtypes = [c.c_long, lp, ip, ip, u16p, lp, ip]
    ref.ref_matching_position_batch.restype = None
    ref.ref_profile_coverage_batch.argtypes = [c.c_long, lp, ip, ip, c.c_int, c.c_int, ip, ip, c.c_long]
    ref.ref_profile_coverage_batch.restype = c.c_long
    return ref


def random_cases(rng, n_a, tspace, jitter_pct=15):
    """Random overlaps grouped by A read (1-40 per read, so the stream kernel's 64-overlap steps cross reads), realistic traces:
    the B advance of every full segment uniform in tspace * (1 +- jitter_pct %), partial first / last segments in proportion;
    four cases in five have consistent B coordinates (bepos = bbpos + the sum of the advances), the rest a B end that disagrees
    with the trace (the reference then pins the last trace point to it: a non-monotone last step).  Masks cut anywhere:
    at, before and behind trace points, inside and outside the match, sometimes empty or inverted."""
    per = rng.integers(1, 41, size=n_a)
    n = int(per.sum())
    a_group = np.repeat(np.arange(n_a), per)
    alen = rng.integers(1500, 40000, size=n_a)
    # A masks per A read
    a_es = rng.integers(0, 1200, size=n_a)
    a_ee = alen - rng.integers(0, 1200, size=n_a)
    deep = rng.random(n_a) < 0.3
    a_es = np.where(deep, rng.integers(0, alen), a_es)
    a_ee = np.where(rng.random(n_a) < 0.3, rng.integers(0, alen + 1), a_ee)
    AL = alen[a_group]
    L = np.minimum(rng.integers(200, 30000, size=n), AL - 1)
    L = np.maximum(L, 1)
    ab = (rng.random(n) * (AL - L + 1)).astype(np.int64)
    ae = ab + L
    comp = rng.integers(0, 2, size=n)
    nseg = (ae + tspace - 1) // tspace - ab // tspace
    off = np.concatenate([[0], np.cumsum(nseg)])
    tot = int(off[-1])
    J = tspace * jitter_pct // 100
    adv = tspace + rng.integers(-J, J + 1, size=tot)
    base = (ab // tspace) * tspace
    first_len = np.where(nseg == 1, ae - ab, base + tspace - ab)
    last_len = np.where(nseg == 1, ae - ab, ae - (base + (nseg - 1) * tspace))
    adv[off[:-1]] = np.maximum(0, first_len + (first_len * rng.integers(-J, J + 1, size=n)) // tspace)
    adv[off[1:] - 1] = np.where(nseg == 1, adv[off[:-1]], np.maximum(0, last_len + (last_len * rng.integers(-J, J + 1, size=n)) // tspace))
    bsum = np.add.reduceat(adv, off[:-1])
    bb = rng.integers(0, 3000, size=n)
    be = bb + bsum
    odd = rng.random(n) < 0.2
    be = np.where(odd, np.maximum(bb + 1, be + rng.integers(-300, 300, size=n)), be)
    blen = be + rng.integers(0, 3000, size=n)
    b_es = rng.integers(0, 1200, size=n)
    b_ee = blen - rng.integers(0, 1200, size=n)
    b_es = np.where(rng.random(n) < 0.3, (rng.random(n) * blen).astype(np.int64), b_es)
    b_ee = np.where(rng.random(n) < 0.3, (rng.random(n) * (blen + 1)).astype(np.int64), b_ee)
    # a share of the masks snapped exactly onto trace-point coordinates of A (>= / <= edges of the first / last point search)
    snap = rng.random(n_a) < 0.25
    a_es = np.where(snap, (a_es // 100) * 100, a_es)
    a_ee = np.where(rng.random(n_a) < 0.25, (a_ee // 100) * 100, a_ee)
    hdr = np.stack([ab, ae, bb, be, comp, a_es[a_group], a_ee[a_group], b_es, b_ee], axis=1).astype(np.int32)
    dif = rng.integers(0, min(tspace // 4, 60), size=tot)
    tr = np.empty(2 * tot, np.uint16)
    tr[0::2] = dif
    tr[1::2] = adv
    traces = [tr[2 * off[i]:2 * off[i + 1]] for i in range(n)]
    return hdr, traces, a_group


@pytest.mark.parametrize("tspace,tbytes,seed", [(100, 1, 1), (100, 2, 2), (200, 2, 3), (50, 1, 4)])
def test_live_process_alignment_and_matching_position(ref_lib, tspace, tbytes, seed):
    """>= 10^5 overlaps per parameter set (4 x 10^5+ in all), every ProcessAlignment kernel and k_matching_position against the
    reference's own functions called on the same arrays."""
    ref = _bind_batch(ref_lib)
    rng = np.random.default_rng(seed)
    hdr, traces, a_group = random_cases(rng, 5200, tspace)
    n = len(hdr)
    assert n >= 100_000
    part = Part(hdr, traces, tbytes, a_group)
    toff = np.ascontiguousarray(part.toff_vals, np.int64)
    types = set()
    for thr in ((1000, 300, 0), (2500, 50, 100)):
        want = np.zeros((n, 10), np.int32)
        ref.ref_process_alignment_batch(n, hdr.ctypes.data_as(ip), part.trace16.ctypes.data_as(u16p), toff.ctypes.data_as(lp), thr[0], thr[1], thr[2],
                                        want.ctypes.data_as(ip))
        _check_classify(part, want, thr, "live tspace=%d tbytes=%d" % (tspace, tbytes))
        types |= set(want[:, 4].tolist())
        assert (want[:, 5] == 0).any() and (want[:, 5] == 1).any()
        assert (want[:, 8] > 0).mean() > 0.2, "masks should cut into most traces"
    assert len(types) >= 7, types
    # GetMatchingPosition: positions around both ends, at trace points and in between
    nq = 150_000
    q = rng.integers(0, n, size=nq).astype(np.int64)
    span = (hdr[q, 1] - hdr[q, 0]).astype(np.int64)
    pos = hdr[q, 0] + (rng.random(nq) * (span + 120)).astype(np.int64) - 60
    at_tp = rng.random(nq) < 0.3
    pos = np.where(at_tp, (pos // 100) * 100, pos).astype(np.int32)
    want = np.zeros(nq, np.int32)
    ref.ref_matching_position_batch(nq, q.ctypes.data_as(lp), pos.ctypes.data_as(ip), hdr.ctypes.data_as(ip), part.trace16.ctypes.data_as(u16p),
                                    toff.ctypes.data_as(lp), want.ctypes.data_as(ip))
    got = part.ctx.matching_position(q, pos)
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, (len(bad), bad[:5], got[bad[:5]], want[bad[:5]], hdr[q[bad[:5]]], pos[bad[:5]])
    assert (want == -1).any() and (want >= 0).any()
    part.close()


@pytest.mark.parametrize("name", ["chimera", "tspace200", "edges", "long_reads"])
def test_live_las_file_reference_reader_against_the_image_kernel(ref_lib, datasets, tmp_path, name):
    """The same .las FILE on both sides, no array of ours in between: the reference opens the DB and the file, parses every record
    and its trace points with its own LAInterface::getOverlap (strand flip included, LAInterface.cpp:1519-1634) and runs its own
    trim_overlap + AddTypesAsymmetric (ref_process_las: the body of maximal.cpp:65-134); k_trim_classify_image gets the file's bytes
    and one offset per overlap (hinge_set_las_image) and decodes the records on the device.  All ten fields of every non-self
    overlap, jittered traces, one- and two-byte traces, both strands, self-overlap records in between, reads of 100 kb."""
    from conftest import clone_dataset
    from hinge_amd import capi, formats
    src, d = datasets(name)
    wd = clone_dataset(src, str(tmp_path / "w"))
    recs = formats.read_las(os.path.join(wd, "G.las"))
    pile = formats.pileups_from_las(recs, d.rlen)
    rng = np.random.default_rng(5)
    # effective read bounds as `hinge filter` would write them: masks that cut a few hundred bases, some reads cut deep
    eff = np.stack([rng.integers(0, 600, size=d.n_reads), d.rlen - rng.integers(0, 600, size=d.n_reads)], axis=1).astype(np.int32)
    deep = rng.random(d.n_reads) < 0.2
    eff[deep, 0] += rng.integers(0, 3000, size=int(deep.sum())).astype(np.int32)
    eff[:, 1] = np.maximum(eff[:, 1], eff[:, 0])
    eff = np.ascontiguousarray(eff)
    ref_lib.ref_process_las.restype = ctypes.c_long
    ref_lib.ref_process_las.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ip, ctypes.c_int, ctypes.c_int, ctypes.c_int, ip, ctypes.c_long]
    image = np.fromfile(os.path.join(wd, "G.las"), dtype=np.uint8)
    win_base, rec_rel = formats.las_image_table(recs, pile)
    ctx = capi.Context(0)
    ctx.set_reads(d.rlen, None)
    ctx.set_pileups(0, d.n_reads - 1, pile.row_ptr, pile.a_span, pile.b_span, pile.b_flag)
    ctx.set_las_image(image, win_base, rec_rel, 1 if recs.tspace <= 125 else 2)
    ctx.set_eff_reads(eff)
    types = set()
    for thr in ((1000, 300, 0), (2500, 50, 100)):
        want = np.zeros((recs.novl, 12), np.int32)
        n = ref_lib.ref_process_las(os.path.join(wd, "G").encode(), os.path.join(wd, "G.las").encode(), eff.ctypes.data_as(ip), thr[0], thr[1], thr[2],
                                    want.ctypes.data_as(ip), recs.novl)
        assert n == recs.novl
        want = want[pile.las_index]                                   # the kept (non-self) records, file order = storage order
        assert np.array_equal(want[:, 1], (pile.b_flag & 0x7FFFFFFF).astype(np.int32))
        got = ctx.trim_classify_part_full(pile.n_ovl, *thr)
        bad = np.nonzero((got != want[:, 2:]).any(axis=1))[0]
        assert len(bad) == 0, (name, thr, len(bad), bad[:5], got[bad[:3]], want[bad[:3]])
        types |= set(want[:, 6].tolist())
    assert len(types) >= 5, types
    assert (pile.b_flag >> 31).any() and not (pile.b_flag >> 31).all()
    ctx.close()


@pytest.mark.parametrize("name", ["tiny", "edges", "long_reads", "tspace200"])
def test_live_coverage_txt_of_the_executable_against_the_reference_reader(ref_lib, datasets, tmp_path, name):
    """`.coverage.txt` as the GPU executable `hinge filter` writes it (the bins k_mask_annotate_q20 stores from its scan registers;
    reads handed to the general kernel: its own) against the same text made by the reference's own LAInterface::getOverlap +
    profileCoverage over the same .las (ref_coverage_txt_las: the loop of filter.cpp:529-548, 599-602) - byte for byte."""
    import subprocess
    from conftest import clone_dataset
    src, d = datasets(name)
    wd = clone_dataset(src, str(tmp_path / "w"))
    hinge = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hinge_amd", "bin", "hinge")
    r = subprocess.run([hinge, "filter", "--db", "G", "--las", "G.las", "-x", "G", "--config", "nominal.ini"], cwd=wd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-1500:]
    ref_lib.ref_coverage_txt_las.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p]
    want_path = os.path.join(wd, "ref.coverage.txt")
    assert ref_lib.ref_coverage_txt_las(os.path.join(wd, "G").encode(), os.path.join(wd, "G.las").encode(), 40, want_path.encode()) == 0
    got, want = open(os.path.join(wd, "G.coverage.txt"), "rb").read(), open(want_path, "rb").read()
    assert len(want) > 10000 and got == want, (len(got), len(want))


@pytest.mark.parametrize("seed", [11, 12])
def test_live_profile_coverage(ref_lib, seed):
    """>= 10^5 pile-ups' worth of bins: k_coverage_bins at cut-offs 0, 300 and one that is no multiple of 20, and the bins K2
    stores, against LAInterface::profileCoverage on the same pile-ups (empty pile-ups, single overlaps, spans shorter than
    twice the cut-off - where the profile goes negative - included)."""
    from hinge_amd import config
    ref = _bind_batch(ref_lib)
    rng = np.random.default_rng(seed)
    n = 52_000
    cnt = rng.choice([0, 1, 2, 3, 8, 30, 120, 400], size=n, p=[0.05, 0.1, 0.1, 0.15, 0.3, 0.2, 0.08, 0.02])
    rl = rng.integers(5000, 30000, size=n)
    rows_ab, rows_ae = [], []
    rl_rep = np.repeat(rl, cnt)
    tot = int(cnt.sum())
    span = np.minimum(rng.integers(30, 12000, size=tot), rl_rep)
    ab = (rng.random(tot) * (rl_rep - span + 1)).astype(np.int64)
    ae = ab + span
    off = np.concatenate([[0], np.cumsum(cnt)])
    rows_ab = [ab[off[i]:off[i + 1]] for i in range(n)]
    rows_ae = [ae[off[i]:off[i + 1]] for i in range(n)]
    ctx, _, _ = _coverage_part(rows_ab, rows_ae, min_rlen=30001)
    row_ptr = np.ascontiguousarray(off, np.int64)
    ab32, ae32 = np.ascontiguousarray(ab, np.int32), np.ascontiguousarray(ae, np.int32)

    def reference(cutoff):
        nb = np.zeros(n, np.int32)
        t = ref.ref_profile_coverage_batch(n, row_ptr.ctypes.data_as(lp), ab32.ctypes.data_as(ip), ae32.ctypes.data_as(ip), 40, cutoff, nb.ctypes.data_as(ip), None, 0)
        cov = np.zeros(max(t, 1), np.int32)
        ref.ref_profile_coverage_batch(n, row_ptr.ctypes.data_as(lp), ab32.ctypes.data_as(ip), ae32.ctypes.data_as(ip), 40, cutoff, nb.ctypes.data_as(ip), cov.ctypes.data_as(ip), t)
        return nb, cov[:t]

    for cutoff in (0, 300, 310):
        want_nb, want_cov = reference(cutoff)
        nb, cov = ctx.coverage_bins(0, n - 1, 40, cutoff)
        assert np.array_equal(nb, want_nb), (cutoff, np.nonzero(nb != want_nb)[0][:5])
        assert np.array_equal(cov, want_cov), (cutoff, np.nonzero(cov != want_cov)[0][:5])
        if cutoff:
            assert (want_cov < 0).any(), "short spans should drive the cut-off profile negative somewhere"
    want_nb, want_cov = reference(0)
    p = config.default_filter_params()
    ctx.coverage_out(True)
    ctx.filter_sweep(p)
    nb, cov = ctx.get_coverage()
    assert np.array_equal(nb[:n], want_nb) and np.array_equal(cov, want_cov)
    ctx.close()


def test_live_std_sort_orders(ref_lib):
    """The reference's comparators under this libstdc++ (LAInterface.cpp:4875-4923) on fresh keys: the host entry point and the
    device replay."""
    from hinge_amd import capi
    ctx = capi.Context(0)
    rng = np.random.default_rng(21)
    for case in range(400):
        n = int(rng.choice([0, 1, 15, 16, 17, 33, 100, 257, 1000, 2048, 4096]) if case % 2 else rng.integers(0, 4097))
        key = rng.integers(0, max(2, n // int(rng.integers(1, 40))), size=n).astype(np.int32)
        for mode in (0, 3):
            want = np.zeros(max(n, 1), np.int32)
            ref_lib.ref_sort_perm(n, key.ctypes.data_as(ip), mode, want.ctypes.data_as(ip))
            assert np.array_equal(capi.sort_order_desc(key.astype(np.int64), 1), want[:n]), (n, mode)
        pos = ctx.debug_pileup_order(key)
        perm = np.zeros(n, np.int32)
        perm[pos] = np.arange(n, dtype=np.int32)
        ref_lib.ref_sort_perm(n, key.ctypes.data_as(ip), 0, want.ctypes.data_as(ip))
        assert np.array_equal(perm, want[:n]), (n, "device replay")
    ctx.close()
