"""GPU parity: `hinge filter` through the C ABI / HIP kernels vs the CPU oracle, byte for byte."""
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


def _hip_filter(wd, mlas, ini="nominal.ini", **kw):
    from hinge_amd import stages
    return run_in(wd, stages.run_filter, "G", "G" if mlas else "G.las", "G", ini, mlas, 0, True, kw.get("force_exact", False),
                  kw.get("ctx"), kw.get("packed", False))


def _compare(wd_o, wd_h):
    bad = [s for s in FILTER_FILES if not filecmp.cmp(os.path.join(wd_o, "G" + s), os.path.join(wd_h, "G" + s), shallow=False)]
    assert not bad, "differs from the oracle: %s" % bad


@pytest.mark.parametrize("name,mlas", [("tiny", False), ("tiny_qv", False), ("tiny_mlas", True), ("tiny_mlas", False),
                                       ("ties", False), ("chimera", False), ("long_repeat", False), ("tspace200", False), ("edges", False)])
@pytest.mark.parametrize("exact", [0, 1, 2, 3, 4, 5, 6])
def test_filter_matches_oracle(datasets, oracle_lib, tmp_path, monkeypatch, name, mlas, exact):
    """exact: 0 the shipped route (k_hinge_count, then k_hinge_call_light - the order-independent evaluation on sorted supporters -,
    then k_hinge_call<CAP> for what is left), 1 the serial exact kernel, 2 the exact replay in
    LDS, 3 without the light kernel (k_hinge_call<CAP>'s own binned evaluation, rounds 1-3), 4 with the quarter-size instance
    k_hinge_call<1024> in front of the second tier (HINGE_CALL_MINI=1), 5 the second tier drawing ITEMS as in round 5 instead of reads with
    their item chains (HINGE_CALL_GROUP=0; the shipped route, 0, draws reads: one sort replay per read)."""
    if exact == 5:
        monkeypatch.setenv("HINGE_CALL_GROUP", "0")
        exact = 0
    if exact == 6:      # the full-size replay instance (76 KiB of LDS) behind the light kernel instead of k_hinge_call<2048, LEAN>
        monkeypatch.setenv("HINGE_CALL_LEAN", "0")
        exact = 0
    if exact == 3:
        monkeypatch.setenv("HINGE_CALL_LIGHT", "0")
        exact = 0
    if exact == 4:
        monkeypatch.setenv("HINGE_CALL_MINI", "1")
        exact = 0
    src, _ = datasets(name)
    wd_o = clone_dataset(src, str(tmp_path / "oracle"))
    wd_h = clone_dataset(src, str(tmp_path / "hip"))
    assert _oracle_filter(oracle_lib, wd_o, mlas) == 0
    assert _hip_filter(wd_h, mlas, force_exact=exact) == 0
    _compare(wd_o, wd_h)
    # the comparison must not be vacuous
    nh = sum((len(l.split()) - 1) // 2 for l in open(os.path.join(wd_o, "G.hinges.txt")))
    na = sum((len(l.split()) - 1) // 2 for l in open(os.path.join(wd_o, "G.repeat.txt")))
    assert na > 0 and nh > 0


@pytest.mark.parametrize("name,mlas,general", [("tiny_qv", False, 0), ("tiny_mlas", True, 0), ("ties", False, 0), ("edges", False, 0), ("tspace200", False, 0),
                                               ("long_repeat", False, 1), ("long_reads", False, 0), ("deep", False, 0)])
def test_filter_packed_route(datasets, oracle_lib, tmp_path, name, mlas, general):
    """The route of the executables: hinge_set_pileups_packed (span copy and facts from the ingest, no k_pileup_facts sweep) and
    the .coverage.txt bins stored by K2 itself (fast kernel, general kernel and the hand-back between them)."""
    from hinge_amd import capi
    src, _ = datasets(name)
    wd_o = clone_dataset(src, str(tmp_path / "oracle"))
    wd_h = clone_dataset(src, str(tmp_path / "hip"))
    assert _oracle_filter(oracle_lib, wd_o, mlas) == 0
    ctx = capi.Context(0)
    ctx.force_general_mask(general)
    assert _hip_filter(wd_h, mlas, ctx=ctx, packed=True) == 0
    _compare(wd_o, wd_h)
    ctx.close()


@pytest.mark.parametrize("mode", ["huge", "past_end"])
def test_filter_packed_route_fallback_reads(datasets, oracle_lib, tmp_path, mode):
    from hinge_amd import capi
    src, _ = _bloated_dataset(datasets, tmp_path, mode)
    wd_o = clone_dataset(src, str(tmp_path / "oracle"))
    wd_h = clone_dataset(src, str(tmp_path / "hip"))
    assert _oracle_filter(oracle_lib, wd_o, False) == 0
    ctx = capi.Context(0)
    assert _hip_filter(wd_h, False, ctx=ctx, packed=True) == 0
    # past_end: the ingest sees a coordinate outside its read, so no span copy is handed over and every read of the part
    # takes the int32 spans; the one bad pile-up still leaves the fast kernel through the fallback list
    assert ctx.fallback_reads() == 1
    _compare(wd_o, wd_h)


def test_context_reuse_with_more_reads(datasets, oracle_lib, tmp_path):
    """One context, two data sets, the second with more reads: the library-owned per-read tables are reallocated and every
    kernel must follow them (hinge_set_reads re-points mask / mean_cov unless a caller table is attached)."""
    from hinge_amd import capi
    ctx = capi.Context(0)
    sizes = []
    for name in ("tiny", "long_repeat", "tiny"):
        src, d = datasets(name)
        sizes.append(d.n_reads)
        wd_o = clone_dataset(src, str(tmp_path / (name + "_oracle%d" % len(sizes))))
        wd_h = clone_dataset(src, str(tmp_path / (name + "_hip%d" % len(sizes))))
        assert _oracle_filter(oracle_lib, wd_o, False) == 0
        assert _hip_filter(wd_h, False, ctx=ctx, packed=len(sizes) % 2 == 0) == 0
        _compare(wd_o, wd_h)
    assert sizes[1] > sizes[0]
    ctx.close()


def test_context_reuse_across_read_sets_with_coverage_out(oracle_lib, tmp_path):
    """One context, K2 storing the coverage bins, read sets with the SAME number of reads and the same part range but different
    lengths (and pile-ups): every run must lay its bin output out for ITS lengths (the layout used to be cached under
    (r_begin, r_end, reso, cut_off) only and hinge_set_reads did not invalidate it: short -> long wrote past the slots)."""
    import dataclasses
    from hinge_amd import capi, synth
    base = synth.CONFIGS["tiny"]
    da = synth.generate(dataclasses.replace(base, len_min=3000, len_max=6000, seed=71))
    n = da.n_reads
    db = None
    for cov in range(80, 200, 10):      # longer reads: the same genome needs fewer of them, so raise the coverage until there are n
        cand = synth.generate(dataclasses.replace(base, len_min=9000, len_max=14000, coverage=cov, seed=72))
        if cand.n_reads >= n:
            db = cand
            break
    assert db is not None
    if db.n_reads > n:      # drop the surplus reads (and their overlaps) from the end of the id range
        keep = (db.aread < n) & (db.bread < n)
        db = dataclasses.replace(db, rlen=db.rlen[:n], aread=db.aread[keep], bread=db.bread[keep], comp=db.comp[keep], ab=db.ab[keep], ae=db.ae[keep],
                                 bb=db.bb[keep], be=db.be[keep], block_first=[0, n])
    assert db.n_reads == n and int(db.rlen.sum()) > 1.5 * int(da.rlen.sum())
    ctx = capi.Context(0)
    for tag, d in (("a", da), ("b", db), ("a2", da)):
        wd_o, wd_h = str(tmp_path / (tag + "_o")), str(tmp_path / (tag + "_h"))
        for wd in (wd_o, wd_h):
            synth.write_dataset(d, wd, "G")
            write_ini(os.path.join(wd, "nominal.ini"))
        assert _oracle_filter(oracle_lib, wd_o, False) == 0
        assert _hip_filter(wd_h, False, ctx=ctx, packed=True) == 0
        first = open(os.path.join(wd_o, "G.mas")).readline().split()[0]
        last = open(os.path.join(wd_o, "G.mas")).read().splitlines()[-1].split()[0]
        assert (int(first), int(last)) == (0, n - 1), "the read sets must span the same part range"
        _compare(wd_o, wd_h)
    ctx.close()


@pytest.mark.parametrize("name,wgs,deal,heavy", [("chimera", 8, 1, 2), ("tiny_mlas", 16, 1, 2), ("long_reads", 16, 1, 2), ("edges", 24, 1, 2), ("chimera", 16, 0, 2),
                                                 ("deep", 8, 1, 2), ("deep", 16, 1, 1), ("deep", 8, 1, 0), ("long_repeat", 8, 1, 2)])
def test_filter_k2_read_deal(datasets, oracle_lib, tmp_path, monkeypatch, name, wgs, deal, heavy):
    """k_mask_annotate_q20's drawn reads dealt XCD-contiguously in storage order (on when the persistent workgroups are a
    multiple of the 8 XCDs: forced here on small data through HINGE_K2_WGS) and in round 2's longest-first order; the deep
    pile-ups of an XCD's sequence spread over its first 60 % (2, the default), first (1) or left where they are (0): same files."""
    from hinge_amd import capi
    monkeypatch.setenv("HINGE_K2_WGS", str(wgs))
    monkeypatch.setenv("HINGE_K2_DEAL", str(deal))
    monkeypatch.setenv("HINGE_K2_HEAVY", str(heavy))
    src, _ = datasets(name)
    wd_o = clone_dataset(src, str(tmp_path / "oracle"))
    wd_h = clone_dataset(src, str(tmp_path / "hip"))
    mlas = name == "tiny_mlas"
    assert _oracle_filter(oracle_lib, wd_o, mlas) == 0
    ctx = capi.Context(0)
    assert _hip_filter(wd_h, mlas, ctx=ctx, packed=True) == 0
    _compare(wd_o, wd_h)
    ctx.close()


@pytest.mark.parametrize("extra", ["ec = 60\n", "coverage = false\n", "hinge_min_support = 3\nhinge_unbridged = 2\nhinge_min_pileup = 3\n",
                                   "no_hinge_region = 200\nrepeat_annotation_gap_threshold = 100\n"])
def test_filter_ini_variants(datasets, oracle_lib, tmp_path, extra):
    src, _ = datasets("tiny_qv")
    wd_o = clone_dataset(src, str(tmp_path / "oracle"))
    wd_h = clone_dataset(src, str(tmp_path / "hip"))
    for wd in (wd_o, wd_h):
        write_ini(os.path.join(wd, "v.ini"), extra_filter=extra, extra_layout="del_telomere = 1\n")
    assert _oracle_filter(oracle_lib, wd_o, False, "v.ini") == 0
    assert _hip_filter(wd_h, False, "v.ini") == 0
    _compare(wd_o, wd_h)


@pytest.mark.parametrize("exact", [0, 2])
def test_filter_deep_pileups(datasets, oracle_lib, tmp_path, exact):
    """Pile-ups of 2049-4096 overlaps: their undecided annotations go through the full-size instance of k_hinge_call (the
    half-size one, two workgroups per CU, takes pile-ups up to 2048)."""
    from hinge_amd import capi, stages
    src, d = datasets("deep")
    counts = np.bincount(d.aread[d.aread != d.bread], minlength=d.n_reads)
    assert ((counts > 2048) & (counts <= 4096)).any()
    wd_o = clone_dataset(src, str(tmp_path / "oracle"))
    wd_h = clone_dataset(src, str(tmp_path / "hip"))
    assert _oracle_filter(oracle_lib, wd_o, False) == 0
    ctx = capi.Context(0)
    assert run_in(wd_h, stages.run_filter, "G", "G.las", "G", "nominal.ini", False, 0, True, exact, ctx) == 0
    small, big = ctx.heavy_items()
    assert big > 0, (small, big)      # (the half-size instance is what every other data set of this file runs)
    _compare(wd_o, wd_h)
    assert sum((len(l.split()) - 1) // 2 for l in open(os.path.join(wd_o, "G.hinges.txt"))) > 0


def test_pileup_order_replays_std_sort(oracle_lib):
    """wave_pileup_order (parallel Hoare partitions in LDS) == std::sort(compare_overlap) of libstdc++."""
    import ctypes
    from hinge_amd import capi
    ctx = capi.Context(0)
    rng = np.random.default_rng(5)
    ip = ctypes.POINTER(ctypes.c_int)
    sizes = [0, 1, 2, 15, 16, 17, 18, 33, 64, 65, 100, 257, 1000, 2048, 4095, 4096] + [int(x) for x in rng.integers(17, 4096, size=40)]
    for n in sizes:
        for kind in range(8):      # kinds 0-5, 7: the packed replay (key | element in one LDS word); 6: keys that span 2^20+ (wide form)
            if kind == 0:
                key = rng.integers(0, 4, size=n)
            elif kind == 6:
                key = rng.integers(-(1 << 30), 1 << 30, size=n) // (1 if n % 2 else 1 << 12) * (1 if n % 2 else 1 << 12)
            elif kind == 7:
                key = rng.integers(-70000, -69000 + n // 8, size=n)
            elif kind == 1:
                key = rng.integers(0, max(1, n // 8) + 1, size=n)
            elif kind == 2:
                key = np.sort(rng.integers(0, 50, size=n))
            elif kind == 3:
                key = np.sort(rng.integers(0, 50, size=n))[::-1]
            elif kind == 4:
                key = rng.integers(0, 1 << 20, size=n)
            else:
                key = np.concatenate([np.arange(n // 2), np.arange(n - n // 2)[::-1]])
            key = np.ascontiguousarray(key, dtype=np.int32)
            perm = np.zeros(max(n, 1), np.int32)
            oracle_lib.oracle_sort_perm(n, key.ctypes.data_as(ip), 0, perm.ctypes.data_as(ip))   # descending, like compare_overlap
            pos = ctx.debug_pileup_order(key)
            got = np.zeros(n, np.int32)
            got[pos] = np.arange(n, dtype=np.int32)
            assert np.array_equal(got, perm[:n]), (n, kind)
    ctx.close()


# ---- K2 kernel selection: 20-bp fast kernel, general kernel, and the hand-back between them -----------------
@pytest.mark.parametrize("name", ["tiny_qv", "long_repeat"])
def test_filter_general_mask_kernel_matches_oracle(datasets, oracle_lib, tmp_path, name):
    """The general two-histogram kernel (any reso / cut_off) forced where the fast kernel would run."""
    from hinge_amd import capi
    src, _ = datasets(name)
    wd_o = clone_dataset(src, str(tmp_path / "oracle"))
    wd_h = clone_dataset(src, str(tmp_path / "hip"))
    assert _oracle_filter(oracle_lib, wd_o, False) == 0
    ctx = capi.Context(0)
    ctx.force_general_mask(1)
    from hinge_amd import stages
    assert run_in(wd_h, stages.run_filter, "G", "G.las", "G", "nominal.ini", False, 0, True, False, ctx) == 0
    _compare(wd_o, wd_h)


@pytest.mark.parametrize("cut_off", [0, 20, 100, 290, 305, 400, 1000])
def test_filter_cut_off_values(datasets, oracle_lib, tmp_path, cut_off):
    """cut_off % 20 == 0 runs the 20-bp kernel with another shift, anything else the general kernel."""
    src, _ = datasets("long_repeat")
    wd_o = clone_dataset(src, str(tmp_path / "oracle"))
    wd_h = clone_dataset(src, str(tmp_path / "hip"))
    for wd in (wd_o, wd_h):
        txt = open(os.path.join(wd, "nominal.ini")).read().replace("cut_off = 300;", "cut_off = %d;" % cut_off)
        assert "cut_off = %d;" % cut_off in txt
        open(os.path.join(wd, "c.ini"), "w").write(txt)
    assert _oracle_filter(oracle_lib, wd_o, False, "c.ini") == 0
    assert _hip_filter(wd_h, False, "c.ini") == 0
    _compare(wd_o, wd_h)


def _bloated_dataset(datasets, tmp_path, mode):
    """`tiny` with one read's pile-up blown up to 66 000 overlaps (mode 'huge') or with a few alignments that run
    past the end of the A read (mode 'past_end'): both must leave the fast kernel through the fallback list."""
    import copy
    from hinge_amd import synth
    _, d0 = datasets("tiny")
    d = copy.copy(d0)
    rng = np.random.default_rng(11)
    counts = np.bincount(d.aread[d.aread != d.bread], minlength=d.n_reads)
    victim = int(np.argmax(counts))
    rows = np.nonzero((d.aread == victim) & (d.bread != victim))[0]
    cols = {k: getattr(d, k).copy() for k in ("aread", "bread", "comp", "ab", "ae", "bb", "be")}
    if mode == "huge":
        extra = rng.choice(rows, size=66000 - len(rows), replace=True)
        for k in cols:
            cols[k] = np.concatenate([cols[k], cols[k][extra]])
    else:
        hit = rows[:: max(1, len(rows) // 5)][:5]
        new_ae = d.rlen[victim] + np.array([1, 40, 120, 250, 300])[: len(hit)]
        cols["be"][hit] += new_ae - cols["ae"][hit]     # keep the B span as long as the A span (trace generator)
        cols["ae"][hit] = new_ae
    order = np.lexsort((cols["ab"], cols["comp"], cols["bread"], cols["aread"]))   # LAsort order
    for k in cols:
        setattr(d, k, cols[k][order])
    wd = str(tmp_path / ("src_" + mode))
    synth.write_dataset(d, wd, "G")
    write_ini(os.path.join(wd, "nominal.ini"))
    return wd, victim


@pytest.mark.parametrize("mode", ["huge", "past_end"])
def test_filter_fallback_reads(datasets, oracle_lib, tmp_path, mode):
    from hinge_amd import capi, stages
    src, _ = _bloated_dataset(datasets, tmp_path, mode)
    wd_o = clone_dataset(src, str(tmp_path / "oracle"))
    wd_h = clone_dataset(src, str(tmp_path / "hip"))
    assert _oracle_filter(oracle_lib, wd_o, False) == 0
    ctx = capi.Context(0)
    assert run_in(wd_h, stages.run_filter, "G", "G.las", "G", "nominal.ini", False, 0, True, False, ctx) == 0
    assert ctx.fallback_reads() == 1
    _compare(wd_o, wd_h)


def _mean_pileups(rng, kind, n_reads):
    """Pile-ups whose per-read mean coverage lands where `kind` wants it.  A read of length L whose n overlaps all span
    [0, L] has K = L / 40 + 2 bins and the sum n (K - 1): mean = n (K - 1) / K, i.e. n - 1 for n < K."""
    rlen = rng.integers(5200, 9000, size=n_reads).astype(np.int32)
    if kind == "clustered":
        cnt = rng.integers(141, 181, size=n_reads)
    elif kind == "wide":
        cnt = rng.integers(1, 700, size=n_reads)
    elif kind == "out_of_range":
        cnt = rng.integers(100, 200, size=n_reads)
        cnt[rng.integers(0, n_reads, 3)] = 4300                # means of ~4280: outside the histogram
    elif kind == "short_reads":
        cnt = rng.integers(100, 300, size=n_reads)
        rlen[rng.random(n_reads) < 0.7] = 4000                 # < 5000 bp: not in the median
    elif kind == "single":
        cnt = rng.integers(50, 90, size=n_reads)
        rlen[:] = 3000
        rlen[n_reads // 3] = 7000
    elif kind == "none":
        cnt = rng.integers(50, 90, size=n_reads)
        rlen[:] = 3000
    else:   # "negative": some pile-ups made of reversed spans (absurd input the kernels must still count like the reference)
        cnt = rng.integers(100, 200, size=n_reads)
    row_ptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    a_of = np.repeat(np.arange(n_reads), cnt)
    a_span = np.stack([np.zeros(len(a_of), np.int32), rlen[a_of]], axis=1).astype(np.int32)
    jitter = rng.integers(0, 400, size=len(a_of)).astype(np.int32)
    a_span[:, 0] += jitter
    if kind == "negative":
        bad = np.isin(a_of, rng.integers(0, n_reads, 40))
        a_span[bad] = a_span[bad][:, ::-1]
    return rlen, row_ptr, np.ascontiguousarray(a_span)


@pytest.mark.parametrize("kind,packed", [(k, p) for k in ("clustered", "wide", "out_of_range", "negative", "short_reads", "single", "none") for p in (False, True)
                                         if not (p and k == "negative")])   # (reversed spans are still inside their reads: the packed route is the others')
def test_stats_median_in_one_launch(oracle_lib, kind, packed):
    """hinge_filter_stats_median (the statistics sweep + the median of the part's own reads, no host round trip): per-read means,
    median and MIN_COV equal the oracle's profileCoverage sums through filter.cpp:642-678; the histogram form (sharded runs)
    hands out exactly the histogram of those means; the scratch is clean for the next launch (three launches, two read sets,
    one context)."""
    import ctypes
    import torch
    from hinge_amd import capi
    from hinge_amd.config import default_filter_params
    P = default_filter_params()
    ctx = capi.Context(0)
    ip = ctypes.POINTER(ctypes.c_int)
    sentinel = np.iinfo(np.int32).min
    for launch, (seed, n_reads) in enumerate([(5, 30_000), (6, 9_000), (5, 30_000)]):
        rng = np.random.default_rng(seed)
        rlen, row_ptr, a_span = _mean_pileups(rng, kind, n_reads)
        n = len(a_span)
        # expected means from the oracle's profileCoverage (cutoff 0), then filter.cpp:642-664
        want_mean = np.full(n_reads, sentinel, np.int64)
        check = rng.choice(n_reads, size=min(n_reads, 400), replace=False)     # the oracle call per read is slow: a sample + the closed form
        K = np.where(np.diff(row_ptr) > 0, 0, 0)
        ab, ae = a_span[:, 0].astype(np.int64), a_span[:, 1].astype(np.int64)
        binof = lambda v: np.where(v < 0, 0, v // 40 + 1)
        tot = np.add.reduceat(binof(ae) - binof(ab), row_ptr[:-1])
        mx = np.maximum.reduceat(np.maximum(ab, ae), row_ptr[:-1])
        K = binof(mx) + 1
        mean_closed = np.where(tot >= 0, tot // np.maximum(K, 1), -((-tot) // np.maximum(K, 1)))          # C division
        cov = np.zeros(4096, np.int32)
        for i in check:
            s, e = int(row_ptr[i]), int(row_ptr[i + 1])
            a0, a1 = np.ascontiguousarray(a_span[s:e, 0]), np.ascontiguousarray(a_span[s:e, 1])
            k = oracle_lib.oracle_profile_coverage(e - s, a0.ctypes.data_as(ip), a1.ctypes.data_as(ip), 40, 0, cov.ctypes.data_as(ip), 4096)
            assert k == K[i] and int(cov[:k].sum()) == tot[i], "closed form of the coverage sum, read %d" % i
        want_mean[rlen >= 5000] = mean_closed[rlen >= 5000]
        vals = np.sort(want_mean[want_mean != sentinel])
        b_span = np.zeros((n, 2), np.int32)
        b_flag = np.zeros(n, np.uint32)
        ctx.set_reads(rlen, None)
        if packed:
            span16, max_pile, in_range = capi.pack_spans(row_ptr, a_span, rlen)
            assert span16 is not None
            ctx.set_pileups_packed(0, n_reads - 1, row_ptr, a_span, b_span, b_flag, span16, max_pile, in_range)
        else:
            ctx.set_pileups(0, n_reads - 1, row_ptr, a_span, b_span, b_flag)
        mean_dev = torch.full((n_reads,), 7, dtype=torch.int32, device="cuda")
        ctx.attach_mean_cov(mean_dev)
        ctx.set_min_cov(5)
        if len(vals) == 0:
            with pytest.raises(capi.HingeError) as ei:
                ctx.filter_stats_median(P, fetch=True)
            assert ei.value.code == -4
            continue
        if launch == 1:     # the histogram form
            hist = torch.full((4096 + 2,), -1, dtype=torch.int32, device="cuda")
            ctx.filter_stats_median(P, hist_dev=hist)
            ctx.synchronize()
            h = hist.cpu().numpy()
            ok = (vals >= 0) & (vals < 4096)
            assert np.array_equal(h[:4096], np.bincount(vals[ok], minlength=4096)) and h[4096] == len(vals) and (h[4097] != 0) == bool((~ok).any())
            assert ctx.get_min_cov() == 5, "the histogram form leaves MIN_COV to hinge_filter_median_from_hist"
            if ok.all():
                ctx.filter_median_from_hist(P, hist)
                assert ctx.get_min_cov() == max(5, int(vals[len(vals) // 2]) // 3)
        else:
            est = ctx.filter_stats_median(P, fetch=True)
            assert est.cov_est == int(vals[len(vals) // 2]) and est.n_long == len(vals)
            assert est.total_cov == int(tot[rlen >= 5000].sum()) and est.num_slot == int(K[rlen >= 5000].sum())
            c = int(vals[len(vals) // 2])
            assert ctx.get_min_cov() == max(5, int(c / 3))
        assert np.array_equal(mean_dev.cpu().numpy().astype(np.int64), want_mean)
        ctx.attach_mean_cov(None)
    ctx.close()


@pytest.mark.parametrize("kind", ["clustered", "wide", "out_of_range", "negative", "sentinels", "single"])
def test_median_kernel_matches_nth_element(kind):
    """k_median_hist (histogram walk, radix-select tail) == sorted(values)[n / 2] as filter.cpp:660-678 uses it."""
    import torch
    from hinge_amd import capi
    from hinge_amd.config import default_filter_params
    rng = np.random.default_rng(3)
    n = 50_000
    sentinel = np.iinfo(np.int32).min
    if kind == "clustered":
        v = rng.integers(140, 180, size=n)
    elif kind == "wide":
        v = rng.integers(0, 4096, size=n)
    elif kind == "out_of_range":
        v = rng.integers(0, 200_000, size=n)
    elif kind == "negative":
        v = rng.integers(-5000, 5000, size=n)
    elif kind == "sentinels":
        v = rng.integers(100, 300, size=n)
        v[rng.random(n) < 0.7] = sentinel
    else:
        v = np.full(n, sentinel, np.int64)
        v[1234] = 77
    v = v.astype(np.int32)
    ctx = capi.Context(0)
    ctx.set_reads(np.full(n, 6000, np.int32), None)
    t = torch.from_numpy(v).cuda()
    ctx.attach_mean_cov(t.data_ptr())
    P = default_filter_params()
    ctx.set_min_cov(5)
    for _ in range(2):   # twice: the kernel must leave its scratch clean
        est = ctx.filter_median(P, 0, n - 1, fetch=True)
        valid = np.sort(v[v != sentinel])
        assert est.n_long == len(valid)
        assert est.cov_est == int(valid[len(valid) // 2])
    assert ctx.get_min_cov() == max(5, int(valid[len(valid) // 2]) // 3)
    # the sharded form: two half-range histograms, summed (what the all-reduce does), then the same median
    if kind in ("clustered", "wide", "sentinels", "single"):
        h1 = torch.zeros(4096 + 2, dtype=torch.int32, device="cuda")
        h2 = torch.zeros(4096 + 2, dtype=torch.int32, device="cuda")
        ctx.set_min_cov(5)
        ctx.filter_median_hist(P, 0, n // 2 - 1, h1)
        ctx.filter_median_hist(P, n // 2, n - 1, h2)
        h = h1 + h2
        assert int(h[4096].item()) == len(valid) and int(h[4097].item()) == 0
        assert np.array_equal(h[:4096].cpu().numpy(), np.bincount(valid, minlength=4096))
        ctx.filter_median_from_hist(P, h)
        assert ctx.get_min_cov() == max(5, int(valid[len(valid) // 2]) // 3)
    elif kind == "out_of_range":
        h1 = torch.zeros(4096 + 2, dtype=torch.int32, device="cuda")
        ctx.filter_median_hist(P, 0, n - 1, h1)
        assert int(h1[4097].item()) != 0
        ctx.filter_median_from_hist(P, h1)
        with pytest.raises(capi.HingeError):
            ctx.check()


def test_filter_long_reads_all_lds_classes(datasets, oracle_lib, tmp_path):
    """Reads of 4 kb .. 120 kb: one, two and four LDS slots per read, and reads too long for a workgroup's LDS."""
    from hinge_amd import capi, stages
    src, d = datasets("long_reads")
    assert (d.rlen <= 19000).any() and ((d.rlen > 21000) & (d.rlen <= 41000)).any() and ((d.rlen > 45000) & (d.rlen <= 85000)).any()
    slot = (19000 // 20 + 1 + 3) // 4 * 4 + 4 + 4 * 64 + 20 + 36     # bins + hot words + zero / total pads (cut_off 300)
    n_too_long = int((d.rlen // 20 >= 4 * slot - 4 * 64 - 20 - 36).sum())   # 20-bp bins beyond a workgroup's four slots
    assert n_too_long > 0
    wd_o = clone_dataset(src, str(tmp_path / "oracle"))
    wd_h = clone_dataset(src, str(tmp_path / "hip"))
    assert _oracle_filter(oracle_lib, wd_o, False) == 0
    ctx = capi.Context(0)
    assert run_in(wd_h, stages.run_filter, "G", "G.las", "G", "nominal.ini", False, 0, True, False, ctx) == 0
    assert ctx.fallback_reads() == n_too_long
    _compare(wd_o, wd_h)


def test_filter_int32_spans(datasets, oracle_lib, tmp_path, monkeypatch):
    """HINGE_NO_SPAN16=1: the streaming kernels read the int32 spans (what they do by themselves when a read is 65536+ bp,
    as in the long_reads data set) instead of the 16|16-bit copy."""
    monkeypatch.setenv("HINGE_NO_SPAN16", "1")
    for name in ("tiny_qv", "long_repeat"):
        src, _ = datasets(name)
        wd_o = clone_dataset(src, str(tmp_path / (name + "_oracle")))
        wd_h = clone_dataset(src, str(tmp_path / (name + "_hip")))
        assert _oracle_filter(oracle_lib, wd_o, False) == 0
        assert _hip_filter(wd_h, False) == 0
        _compare(wd_o, wd_h)


@pytest.mark.parametrize("name,genome,mlas", [("cfg3_nctc", 400_000, False), ("cfg4_yeast", 600_000, True), ("cfg2_ecoli160", 250_000, False)])
def test_baseline_configs_scaled_down(oracle_lib, tmp_path, name, genome, mlas):
    """BASELINE.json's other configurations (repeat-rich NCTC-like with chimeras; 8-block yeast-like with --mlas; the bench
    workload itself) at a genome size the oracle finishes in seconds: every output file of `hinge filter`."""
    import dataclasses
    from hinge_amd import synth
    spec = dataclasses.replace(synth.CONFIGS[name], genome_len=genome)
    d = synth.generate(spec)
    src = str(tmp_path / "src")
    synth.write_dataset(d, src, "G", write_bases=False)
    write_ini(os.path.join(src, "nominal.ini"))
    wd_o = clone_dataset(src, str(tmp_path / "oracle"))
    wd_h = clone_dataset(src, str(tmp_path / "hip"))
    assert _oracle_filter(oracle_lib, wd_o, mlas) == 0
    assert _hip_filter(wd_h, mlas) == 0
    _compare(wd_o, wd_h)
    assert sum((len(l.split()) - 1) // 2 for l in open(os.path.join(wd_o, "G.repeat.txt"))) > 0


def _island_pileups(rng, n_reads):
    """Pile-ups whose cutoff-300 coverage is a set of islands above MIN_COV of chosen lengths and positions: the shapes the
    longest-run search has to get right (equal runs, runs across the 64-bin word boundaries, at the read's ends, none, one bin)."""
    rlen, rows = [], []
    for r in range(n_reads):
        kind = r % 8
        if kind == 0:      # the cutoff profile has exactly 64 / 65 / 128 / 129 bins
            rl = int(rng.choice([2219, 2220, 2259, 2260, 4779, 4780, 4819, 4820]))
        else:
            rl = int(rng.integers(2600, 16000))
        isl = []
        if kind == 1:      # two or three runs of exactly the same length: the first must win
            L = 40 * int(rng.integers(2, 12))
            x = 40 * int(rng.integers(8, 20))
            for _ in range(int(rng.integers(2, 4))):
                if x + L + 700 < rl:
                    isl.append((x, L))
                x += L + 40 * int(rng.integers(16, 30))
        elif kind == 2:    # runs that straddle bins 63 | 64, 127 | 128, 191 | 192
            for edge in (64, 128, 192):
                a = 40 * (edge - int(rng.integers(1, 6)))
                L = 40 * int(rng.integers(2, 12))
                if a + L + 700 < rl:
                    isl.append((a, L))
        elif kind == 3:    # a run that reaches the read's end, one that starts at 0
            isl.append((0, 40 * int(rng.integers(3, 20))))
            isl.append((max(rl - 40 * int(rng.integers(10, 30)), 1200), rl))
        elif kind == 4:    # none at all
            pass
        elif kind == 5:    # many short ones, every few bins
            x = 400
            while x + 1200 < rl:
                isl.append((x, 40 * int(rng.integers(1, 4))))
                x += 40 * int(rng.integers(18, 40))
        else:              # random
            x = 40 * int(rng.integers(0, 30))
            while x + 900 < rl:
                L = 40 * int(rng.integers(1, 60))
                isl.append((x, L))
                x += L + 40 * int(rng.integers(16, 50))
        ab, ae = [], []
        for a, L in isl:
            depth = int(rng.integers(7, 12))          # above MIN_COV = 5 (+ a few that are not: depth 4)
            if rng.random() < 0.15:
                depth = 4
            for _ in range(depth):
                ab.append(max(a - 300, 0))
                ae.append(min(a + L + 300, rl))
        # a thin background so that the profile has bins everywhere
        for _ in range(3):
            ab.append(0)
            ae.append(rl)
        rlen.append(rl)
        rows.append((np.array(ab, np.int32), np.array(ae, np.int32)))
    return np.array(rlen, np.int32), rows


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_coverage_mask_run_shapes(seed):
    """The first longest run of bins above MIN_COV: the fast kernel (a DPP max-scan on the vector unit), the general kernel (a walk
    over ballot bits on the scalar unit) and the specification-level model (tests/spec_model.py) on adversarial profiles."""
    from hinge_amd import capi
    from hinge_amd.config import default_filter_params
    import spec_model
    rng = np.random.default_rng(seed)
    n = 240
    rlen, rows = _island_pileups(rng, n)
    row_ptr = np.zeros(n + 1, np.int64)
    for i, (ab, _) in enumerate(rows):
        row_ptr[i + 1] = row_ptr[i] + len(ab)
    m = int(row_ptr[-1])
    a_span = np.zeros((m, 2), np.int32)
    for i, (ab, ae) in enumerate(rows):
        a_span[row_ptr[i]:row_ptr[i + 1], 0] = ab
        a_span[row_ptr[i]:row_ptr[i + 1], 1] = ae
    b_span = a_span.copy()
    b_flag = ((np.arange(m) % n).astype(np.uint32))          # some other read, forward strand (hinge calling is not run)
    P = default_filter_params()
    got = {}
    for general in (0, 1):
        ctx = capi.Context(0)
        ctx.force_general_mask(general)
        ctx.set_reads(rlen, None)
        ctx.set_pileups(0, n - 1, row_ptr, a_span, b_span, b_flag)
        ctx.set_min_cov(P.min_cov)
        ctx.filter_stats(P)
        ctx.filter_median(P, 0, n - 1, fetch=True)
        min_cov = ctx.get_min_cov()
        ctx.filter_mask_annotate(P)
        assert ctx.fallback_reads() == 0
        got[general] = (ctx.get_masks(), min_cov)
        ctx.close()
    (mask_q, cmask_q, _), mc = got[0]
    (mask_g, cmask_g, _), mc_g = got[1]
    assert mc == mc_g
    assert np.array_equal(mask_q, mask_g) and np.array_equal(cmask_q, cmask_g)
    for i, (ab, ae) in enumerate(rows):
        want_mask, want_cmask = spec_model.coverage_mask(spec_model.coverage(ab, ae, P.cut_off), mc)
        assert tuple(mask_q[i]) == want_mask and tuple(cmask_q[i]) == want_cmask, (i, i % 8, tuple(mask_q[i]), want_mask, tuple(cmask_q[i]), want_cmask)


@pytest.mark.parametrize("seed", [11, 12])
def test_annotation_window_edges(seed):
    """Coverage steps placed on the edges the candidate pass must get right: the first and last bin of the annotation window
    (mask +- NO_HINGE_REGION, K0 - 3), the 64-bin word boundaries of the flag words, steps just below / above the threshold.
    Fast kernel == general kernel == specification-level model (masks, merged annotations)."""
    from hinge_amd import capi
    from hinge_amd.config import default_filter_params
    import spec_model
    rng = np.random.default_rng(seed)
    n = 200
    rlen, rows = [], []
    for r in range(n):
        rl = int(rng.integers(3200, 15000))
        base_depth = int(rng.integers(18, 40))
        ab = [0] * base_depth
        ae = [rl] * base_depth
        spots = [820 + 40 * int(rng.integers(-3, 4)), 40 * 63, 40 * 64, 40 * 127, 40 * 128, 40 * 191, 40 * 192,
                 rl - 800 + 40 * int(rng.integers(-3, 4)), 40 * (rl // 40 - 2), 40 * (rl // 40 - 3)]
        spots += [40 * int(rng.integers(1, rl // 40)) for _ in range(4)]
        for x in rng.choice(spots, size=int(rng.integers(1, 5)), replace=False):
            x = int(x) + int(rng.choice([0, 0, 1, 39, -1]))
            if not (0 < x < rl - 50):
                continue
            extra = int(rng.choice([8, 10, 11, 14, 20, 21, 30, 60]))      # around min / max_repeat_annotation and the coverage fraction
            if rng.random() < 0.5:      # a step up: reads that begin at x
                ab += [x] * extra
                ae += [rl] * extra
            else:                       # a step down: reads that end at x
                ab += [0] * extra
                ae += [x] * extra
        rlen.append(rl)
        rows.append((np.array(ab, np.int32), np.array(ae, np.int32)))
    rlen = np.array(rlen, np.int32)
    row_ptr = np.zeros(n + 1, np.int64)
    for i, (ab, _) in enumerate(rows):
        row_ptr[i + 1] = row_ptr[i] + len(ab)
    m = int(row_ptr[-1])
    a_span = np.zeros((m, 2), np.int32)
    for i, (ab, ae) in enumerate(rows):
        a_span[row_ptr[i]:row_ptr[i + 1], 0] = ab
        a_span[row_ptr[i]:row_ptr[i + 1], 1] = ae
    b_flag = ((np.arange(m) % n).astype(np.uint32))
    P = default_filter_params()
    PM = {"NHR": P.no_hinge_region, "CF": P.coverage_fraction, "MINRA": P.min_repeat_annotation, "MAXRA": P.max_repeat_annotation, "GAP": P.repeat_annotation_gap}
    got = {}
    for general in (0, 1):
        ctx = capi.Context(0)
        ctx.force_general_mask(general)
        ctx.set_reads(rlen, None)
        ctx.set_pileups(0, n - 1, row_ptr, a_span, a_span.copy(), b_flag)
        ctx.set_min_cov(P.min_cov)
        ctx.filter_stats(P)
        ctx.filter_median(P, 0, n - 1, fetch=True)
        mc = ctx.get_min_cov()
        ctx.filter_mask_annotate(P)
        off, pos, typ, _ = ctx.get_annotations()
        got[general] = (ctx.get_masks()[0], off.copy(), pos.copy(), typ.copy(), mc)
        ctx.close()
    mask_q, off_q, pos_q, typ_q, mc = got[0]
    mask_g, off_g, pos_g, typ_g, _ = got[1]
    assert np.array_equal(mask_q, mask_g) and np.array_equal(off_q, off_g)
    assert np.array_equal(pos_q[:off_q[-1]], pos_g[:off_g[-1]]) and np.array_equal(typ_q[:off_q[-1]], typ_g[:off_g[-1]])
    n_anno = 0
    for i, (ab, ae) in enumerate(rows):
        want_mask, _ = spec_model.coverage_mask(spec_model.coverage(ab, ae, P.cut_off), mc)
        assert tuple(mask_q[i]) == want_mask, (i, tuple(mask_q[i]), want_mask)
        want = spec_model.annotate(spec_model.coverage(ab, ae, 0), want_mask, mc, PM)
        have = list(zip(pos_q[off_q[i]:off_q[i + 1]].tolist(), typ_q[off_q[i]:off_q[i + 1]].tolist()))
        assert have == want, (i, have, want)
        n_anno += len(want)
    assert n_anno > n // 2      # the steps do produce annotations
