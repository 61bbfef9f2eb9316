"""The N > 1 path on CPU: world_size 2, gloo.  The exchange / orchestration layer (hinge_amd/dist.py) runs
for real; the per-block compute is replaced by tables taken from a CPU-oracle run of the same data, so
the test checks what the collectives must deliver: the global median / the --mlas running MIN_COV, the
mask table every rank sees before hinge calling (all parts in "merged" mode, only parts <= own in
"mlas" mode), and the assembled global hinge list."""
import os
import sys

import numpy as np
import pytest

import conftest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import clone_dataset, run_in

MEAN_SENTINEL = -(2 ** 31)


def _mean_cov_from_coverage(path, rlen):
    out = np.full(len(rlen), MEAN_SENTINEL, np.int64)
    for line in open(path):
        tok = line.split()
        i = int(tok[1])
        if rlen[i] < 5000:
            continue
        vals = [int(t.split(",")[1]) for t in tok[2:]]
        s = sum(vals)
        k = max(1, len(vals))
        out[i] = int(s / k) if s >= 0 else -int(-s / k)
    return out


def _pairs(path):
    rows = []
    for line in open(path):
        tok = line.split()
        i = int(tok[0])
        for j in range(1, len(tok) - 1, 2):
            rows.append((i, int(tok[j]), int(tok[j + 1])))
    return np.array(rows, np.int32).reshape(-1, 3)


class TableBackend:
    """Stands in for HipBackend: same methods, values from precomputed tables."""

    def __init__(self, lo, hi, mean_all, mask_all, hinge_rows, expect_min_cov, expect_visible_mask, ini_min_cov=5):
        self.lo, self.hi = lo, hi
        self.mean_all, self.mask_all, self.rows = mean_all, mask_all, hinge_rows
        self.expect_min_cov, self.expect_visible = expect_min_cov, expect_visible_mask
        self.ini_min_cov, self.est_cov = ini_min_cov, 0
        self.min_cov = ini_min_cov
        self.checked = []

    def attach(self, mean_cov, mask):
        self.mean_cov, self.mask = mean_cov, mask

    def begin(self):
        self.min_cov = self.ini_min_cov

    def stats(self):
        self.mean_cov[self.lo:self.hi] = torch.from_numpy(self.mean_all[self.lo:self.hi].astype(np.int32))

    def _median(self, lo, hi):
        v = self.mean_cov[lo:hi + 1]
        v = np.sort(v[v != MEAN_SENTINEL].numpy())
        return int(v[len(v) // 2])

    def median(self, lo, hi):
        c = self._median(lo, hi)
        self.min_cov = max(self.min_cov, int(c / 3))

    def median_fetch(self, lo, hi):
        return self._median(lo, hi)

    def median_hist(self, lo, hi, out=None):
        v = self.mean_cov[lo:hi + 1].numpy()
        v = v[v != MEAN_SENTINEL]
        h = np.zeros(4096 + 2, np.int32)
        ok = (v >= 0) & (v < 4096)
        h[:4096] = np.bincount(v[ok], minlength=4096)
        h[4096] = len(v)
        h[4097] = int((~ok).any())
        if out is not None:
            out.copy_(torch.from_numpy(h))
            return out
        return torch.from_numpy(h)

    def median_from_hist(self, hist):
        h = hist.numpy()
        assert h[4097] == 0
        c = int(np.searchsorted(np.cumsum(h[:4096]), h[4096] // 2, side="right"))
        self.min_cov = max(self.min_cov, int(c / 3))

    def set_min_cov(self, v):
        self.min_cov = v

    def stats_median(self, out=None):
        self.stats()
        if out is None:
            self.median(self.lo, self.hi - 1)
        else:
            self.median_hist(self.lo, self.hi - 1, out=out)

    def mask_annotate(self):
        assert self.min_cov == self.expect_min_cov, (self.min_cov, self.expect_min_cov)
        self.mask[self.lo:self.hi] = torch.from_numpy(self.mask_all[self.lo:self.hi])
        self.checked.append("min_cov")

    def hinges(self):
        assert np.array_equal(self.mask.numpy(), self.expect_visible), "mask table seen by hinge calling is wrong"
        self.checked.append("masks")

    FAKE_POS = 123456

    def hinge_rows(self, drop_last=True):
        """The oracle's .hinges.txt already stops before the last A read of the (merged or per-part) run, so a hinge is planted
        on this block's last read here: the orchestration must drop it exactly where the reference's `i < r_end` would."""
        sel = (self.rows[:, 0] >= self.lo) & (self.rows[:, 0] < self.hi)
        mine = np.concatenate([self.rows[sel], np.array([[self.hi - 1, self.FAKE_POS, 1]], np.int32)])
        if drop_last:
            mine = mine[mine[:, 0] != self.hi - 1]
        t = torch.from_numpy(np.ascontiguousarray(mine))
        return t, int(t.shape[0])

    def max_pileup(self):
        return 100

    def status_code(self):
        self.checked.append("status")
        return 0

    def regrow(self):
        raise AssertionError("no capacity error was reported")

    def raise_status(self, codes):
        raise AssertionError(codes)


def _worker(rank, world, port, mode, median, first, mean_all, mask_all, rows, expect_min_cov, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hinge_amd.dist import BlockTable, Exchange, ShardedFilter
        blocks = BlockTable(first)
        lo, hi = first[rank], first[rank + 1]
        visible = mask_all.copy()
        if mode == "mlas":
            visible[hi:] = 0
        be = TableBackend(lo, hi, mean_all, mask_all, rows, expect_min_cov[rank], visible)
        job = ShardedFilter(be, Exchange(blocks, torch.device("cpu")), mode=mode, median=median)
        got = job.step(fetch_hinges=True)
        assert be.checked == ["min_cov", "masks", "status"]
        assert np.array_equal(job.mean_cov.numpy()[lo:hi], mean_all[lo:hi].astype(np.int32))
        if mode == "merged" and median == "gather":
            assert np.array_equal(job.mean_cov.numpy(), mean_all.astype(np.int32)), "all-gather of mean coverage"
        want = rows
        if mode == "merged":   # one merged .las keeps the hinges of every block's last read except the global last one
            planted = np.array([[first[k + 1] - 1, TableBackend.FAKE_POS, 1] for k in range(world - 1)], np.int32).reshape(-1, 3)
            want = np.concatenate([np.concatenate([rows[(rows[:, 0] >= first[k]) & (rows[:, 0] < first[k + 1])], planted[k:k + 1]]) for k in range(world)])
        assert np.array_equal(got.numpy(), want), "assembled hinge list"
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


def _pipelined_worker(rank, world, port, first, tables, ret):
    """Three jobs per rank through dist.step_pipelined: every job must see its own global median and its own full mask table."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hinge_amd.dist import BlockTable, Exchange, ShardedFilter, step_pipelined
        lo, hi = first[rank], first[rank + 1]
        jobs, bes = [], []
        for mean_all, mask_all, expect in tables:
            be = TableBackend(lo, hi, mean_all, mask_all, np.zeros((0, 3), np.int32), expect, mask_all)
            jobs.append(ShardedFilter(be, Exchange(BlockTable(first), torch.device("cpu")), mode="merged", median="hist"))
            bes.append(be)
        step_pipelined(jobs)
        for be, job, (mean_all, mask_all, _) in zip(bes, jobs, tables):
            assert be.checked == ["min_cov", "masks"]
            assert np.array_equal(job.mask.numpy(), mask_all)
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


def test_step_pipelined_keeps_jobs_apart():
    import torch.multiprocessing as mp
    world, n = 2, 600
    first = [0, n // 2, n]            # equal blocks: the in-place asynchronous all-gather
    rng = np.random.default_rng(5)
    tables = []
    for k in range(3):
        mean_all = rng.integers(30 + 60 * k, 90 + 60 * k, n).astype(np.int32)
        mean_all[rng.integers(0, n, 20)] = MEAN_SENTINEL
        mask_all = np.stack([rng.integers(0, 100, n), rng.integers(100, 9000, n)], axis=1).astype(np.int32) + 1000 * k
        v = np.sort(mean_all[mean_all != MEAN_SENTINEL])
        tables.append((mean_all, mask_all, max(5, int(v[len(v) // 2]) // 3)))
    port = conftest.free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_pipelined_worker, args=(r, world, port, first, tables, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs) and len(ret) == world


def _batch_worker(rank, world, port, R, G, S, tables, ret):
    """dist.PartBatch: R parts per rank, exchanges batched (one all-reduce + G all-gathers per step)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hinge_amd.dist import PartBatch
        batch = PartBatch(R, S, torch.device("cpu"), gather_groups=G)
        assert len(batch.groups) == G and sorted(p for g in batch.groups for p in g) == list(range(R))
        bes = []
        for p in range(R):
            gi = batch.slot[p][0]
            mean_g, mask_g, expect = tables[gi]
            lo = batch.id_base(p)
            assert batch.n_ids(p) == len(mean_g) and batch.global_ids(p, rank, 0) == lo
            # hinge rows of this part: one per rank, on the block's first read, and one on its last read (dropped on the last rank)
            rows = np.array([[batch.id_base(p, r), 100 + p, -1] for r in range(world)] + [[batch.id_base(p, r) + S - 1, 7, 1] for r in range(world)], np.int32)
            be = TableBackend(lo, lo + S, mean_g, mask_g, rows, expect[p], mask_g)
            batch.set_backend(p, be)
            bes.append(be)
        batch.settle()
        for be in bes:
            assert be.checked == ["min_cov", "masks", "status"], be.checked
        for gi in range(G):
            assert np.array_equal(batch.masks[gi].numpy(), tables[gi][1])
        sums = batch.table_checksums()
        lists = batch.hinge_lists()
        for p in range(R):
            got = lists[p].numpy()
            want = [[batch.id_base(p, r), 100 + p, -1] for r in range(world)] + [[batch.id_base(p, r) + S - 1, 7, 1] for r in range(world - 1)] + \
                   [[batch.id_base(p, r) + S - 1, TableBackend.FAKE_POS, 1] for r in range(world - 1)]     # TableBackend plants one more
            assert sorted(map(tuple, got.tolist())) == sorted(map(tuple, want)), (p, got)
            # rank order
            assert list(got[:, 0]) == sorted(got[:, 0])
        ret[rank] = sums
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("R,G", [(4, 1), (4, 2), (3, 2)])
def test_part_batch_batches_the_exchanges(R, G):
    import torch.multiprocessing as mp
    world, S = 2, 150
    rng = np.random.default_rng(11 + R + G)
    cut = [round(k * R / G) for k in range(G + 1)]
    tables = []
    for gi in range(G):
        J = cut[gi + 1] - cut[gi]
        n = world * J * S
        mean_g = rng.integers(20, 400, n).astype(np.int32)
        for r in range(world):               # every part its own coverage level, so that a mixed-up histogram row shows
            for j in range(J):
                lo = (r * J + j) * S
                mean_g[lo:lo + S] += 300 * (cut[gi] + j)
                mean_g[lo + S - 5:lo + S] = MEAN_SENTINEL        # padding ids behind the block's reads
        mask_g = np.stack([rng.integers(0, 100, n), rng.integers(100, 9000, n)], axis=1).astype(np.int32) + 1000 * gi
        expect = {}
        for j in range(J):
            ids = np.concatenate([np.arange((r * J + j) * S, (r * J + j + 1) * S) for r in range(world)])
            v = np.sort(mean_g[ids][mean_g[ids] != MEAN_SENTINEL])
            expect[cut[gi] + j] = max(5, int(v[len(v) // 2]) // 3)
        tables.append((mean_g, mask_g, expect))
    port = conftest.free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_batch_worker, args=(r, world, port, R, G, S, tables, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs) and len(ret) == world
    assert ret[0] == ret[1], "mask-table checksums differ between the ranks"


@pytest.mark.parametrize("mode,median,equal_blocks", [("merged", "hist", False), ("merged", "gather", False), ("merged", "hist", True),
                                                       ("mlas", "gather", False)])
def test_sharded_filter_exchanges(oracle_lib, tmp_path, mode, median, equal_blocks):
    import dataclasses
    from hinge_amd import synth
    from hinge_amd.dist import mlas_min_cov
    from conftest import write_ini
    d = synth.generate(dataclasses.replace(synth.CONFIGS["tiny_mlas"], n_blocks=2))
    wd = str(tmp_path / "data")
    synth.write_dataset(d, wd, "G")
    write_ini(os.path.join(wd, "nominal.ini"))
    mlas = mode == "mlas"
    assert run_in(wd, oracle_lib.oracle_filter, b"G", b"G" if mlas else b"G.las", int(mlas), b"G", b"nominal.ini", b"") == 0
    first = list(d.block_first)
    if equal_blocks:   # the in-place all-gather path needs equal blocks: the split is only the exchange layer's view here
        assert d.n_reads % 2 == 0 or True
        half = d.n_reads // 2
        if d.n_reads % 2:
            pytest.skip("odd number of reads")
        first = [0, half, d.n_reads]
    mean_all = _mean_cov_from_coverage(os.path.join(wd, "G.coverage.txt"), d.rlen)
    mask_all = np.loadtxt(os.path.join(wd, "G.mas"), dtype=np.int64)[:, 1:].astype(np.int32)
    rows = _pairs(os.path.join(wd, "G.hinges.txt"))
    # expected MIN_COV per rank from the path's own definition (filter.cpp:660-678)
    def med(lo, hi):
        v = np.sort(mean_all[lo:hi][mean_all[lo:hi] != MEAN_SENTINEL])
        return int(v[len(v) // 2])
    if mlas:
        expect = mlas_min_cov(5, [med(first[k], first[k + 1]) for k in range(2)])
    else:
        expect = [max(5, int(med(0, d.n_reads) / 3))] * 2
    port = conftest.free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, mode, median, first, mean_all, mask_all, rows, expect, ret), nprocs=2, join=True)
    assert dict(ret) == {0: 1, 1: 1}


def test_mlas_min_cov_is_a_running_max():
    from hinge_amd.dist import mlas_min_cov
    assert mlas_min_cov(5, [30, 12, 60, 9]) == [10, 10, 20, 20]
    assert mlas_min_cov(5, [3, 6]) == [5, 5]
    assert mlas_min_cov(-1, [2, 0]) == [0, 0]
    assert mlas_min_cov(5, [30, 90], est_cov_override=45) == [15, 15]


# ---- sharded `hinge maximal`: exchange 4 (containment candidates) + the sequential resolution on every rank ---------------
class OracleMaximalBackend:
    """Stands in for HipMaximalBackend: the block's best overlaps are picked by the product's host code
    (dist.pick_best_pairs), classified one by one by the CPU oracle's ProcessAlignment."""

    def __init__(self, oracle_lib, pile, recs, eff, lo, hi, length_threshold, thr, use_two):
        import ctypes
        from hinge_amd.dist import pick_best_pairs
        self.active0 = ((eff[:, 1] - eff[:, 0]) >= length_threshold).astype(np.uint8)
        sel, a_of = pick_best_pairs(pile.row_ptr, pile.a_span, pile.b_span, pile.b_flag, lo, hi, self.active0, use_two, self_before=pile.self_before)
        ip, u16p = ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint16)
        toff = recs.trace_off[:-1][pile.las_index]
        tlen = recs.rec["tlen"][pile.las_index]
        rows = []
        want = np.zeros(10, np.int32)
        for k, a in zip(sel, a_of):
            b = int(pile.b_flag[k] & 0x7FFFFFFF)
            hdr = np.array([pile.a_span[k, 0], pile.a_span[k, 1], pile.b_span[k, 0], pile.b_span[k, 1], int(pile.b_flag[k] >> 31),
                            eff[a, 0], eff[a, 1], eff[b, 0], eff[b, 1]], np.int32)
            tr = recs.trace[toff[k]:toff[k] + tlen[k]].astype(np.uint16)
            oracle_lib.oracle_process_alignment(hdr.ctypes.data_as(ip), tr.ctypes.data_as(u16p), len(tr), thr[0], thr[1], thr[2], want.ctypes.data_as(ip))
            if want[4] == 3:    # BCOVERA
                rows.append((int(a), b))
        self.rows = np.array(rows, np.int32).reshape(-1, 2)

    def initial_active(self):
        return self.active0

    def candidates(self):
        t = torch.from_numpy(np.ascontiguousarray(self.rows))
        return t, int(t.shape[0])


def _maximal_worker(rank, world, port, wd, first, rlen, want_active, want_contained, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from hinge_amd import formats
        from hinge_amd.dist import BlockTable, Exchange, ShardedMaximal
        lo, hi = first[rank], first[rank + 1]
        recs = formats.read_las(os.path.join(wd, "G.%d.las" % (rank + 1)))          # this rank's block only
        pile = formats.pileups_from_las(recs, rlen)
        eff = np.loadtxt(os.path.join(wd, "G.mas"), dtype=np.int64)[:, 1:].astype(np.int32)
        be = OracleMaximalBackend(oracle.oracle_lib(), pile, recs, eff, lo, hi, 1000, (1000, 300, 0), True)
        assert len(be.rows) > 0 and be.rows[:, 0].min() >= lo and be.rows[:, 0].max() < hi
        job = ShardedMaximal(be, Exchange(BlockTable(first), torch.device("cpu")))
        active = job.step()
        assert np.array_equal(active, want_active), "maximal-read mask"
        got = ["%d\t%d" % (i, c) for i, c in enumerate(job.containing) if c >= 0]
        assert got == want_contained, ".contained.txt (the container named is the last covering B in the reference's hash-map order)"
        ret[rank] = int(active.sum())
    finally:
        dist.destroy_process_group()


def test_sharded_maximal_mask(oracle_lib, tmp_path):
    """world 2: each rank classifies its own block, the candidate rows are all-gathered, both ranks end with the
    maximal-read mask of the reference's sequential --mlas loop (= .max of the oracle)."""
    import dataclasses
    from hinge_amd import synth
    from conftest import write_ini
    d = synth.generate(dataclasses.replace(synth.CONFIGS["tiny_mlas"], n_blocks=2))
    wd = str(tmp_path / "data")
    synth.write_dataset(d, wd, "G")
    write_ini(os.path.join(wd, "nominal.ini"))
    assert run_in(wd, oracle_lib.oracle_filter, b"G", b"G", 1, b"G", b"nominal.ini", b"") == 0
    assert run_in(wd, oracle_lib.oracle_maximal, b"G", b"G", 1, b"G", b"nominal.ini") == 0
    want = np.zeros(d.n_reads, np.uint8)
    want[np.loadtxt(os.path.join(wd, "G.max"), dtype=np.int64)] = 1
    assert 0 < want.sum() < d.n_reads
    port = conftest.free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    want_contained = open(os.path.join(wd, "G.contained.txt")).read().split("\n")[:-1]
    assert len(want_contained) > 10
    mp.spawn(_maximal_worker, args=(2, port, wd, list(d.block_first), d.rlen, want, want_contained, ret), nprocs=2, join=True)
    assert dict(ret) == {0: int(want.sum()), 1: int(want.sum())}


def test_resolve_containment_order_semantics():
    """A container of lower id counts with its final state, one of higher id with its initial state; the last row of a
    removed read's group is the container .contained.txt names (maximal.cpp:780-858)."""
    from hinge_amd import capi
    a = np.array([1, 1, 1, 1, 0], np.uint8)
    c = capi.resolve_containment(a, np.array([[0, 1], [1, 2], [1, 4], [2, 3], [3, 0]], np.int32))
    assert a.tolist() == [0, 0, 0, 1, 0] and c.tolist() == [1, 4, 3, -1, -1]
    with pytest.raises(capi.HingeError):
        capi.resolve_containment(np.ones(3, np.uint8), np.array([[1, 0], [0, 1]], np.int32))     # not grouped by ascending a
    with pytest.raises(capi.HingeError):
        capi.resolve_containment(np.ones(3, np.uint8), np.array([[0, 3]], np.int32))             # id out of range


def test_pick_best_pairs_replays_std_sort_beyond_16(oracle_lib):
    """A pair of reads with more than 16 overlaps: std::sort is no longer an insertion sort there, so which of several equally
    long overlaps come first (and get classified) is libstdc++'s introsort's business; dist.pick_best_pairs follows it through
    hinge_sort_order_desc, checked against the oracle's pinned std::sort replay run twice in a row (maximal.cpp:790-805)."""
    import ctypes
    from hinge_amd import capi, dist
    ip = ctypes.POINTER(ctypes.c_int)
    rng = np.random.default_rng(1)

    def oracle_twice(keys):
        k = np.ascontiguousarray(keys, dtype=np.int32)
        p1 = np.zeros(len(k), np.int32)
        oracle_lib.oracle_sort_perm(len(k), k.ctypes.data_as(ip), 0, p1.ctypes.data_as(ip))
        k2 = np.ascontiguousarray(k[p1])
        p2 = np.zeros(len(k), np.int32)
        oracle_lib.oracle_sort_perm(len(k), k2.ctypes.data_as(ip), 0, p2.ctypes.data_as(ip))
        return p1, p1[p2]

    for n in (5, 16, 17, 40, 300, 5000):
        k = rng.integers(0, 6, n).astype(np.int64) * 100
        once, twice = oracle_twice(k)
        assert np.array_equal(capi.sort_order_desc(k, 1), once) and np.array_equal(capi.sort_order_desc(k, 2), twice), n
    oracle_lib.oracle_umap_order.argtypes = [ctypes.c_int, ip, ip]

    def umap_order(keys):
        k = np.ascontiguousarray(keys, dtype=np.int32)
        out = np.zeros(len(k), np.int32)
        n = oracle_lib.oracle_umap_order(len(k), k.ctypes.data_as(ip), out.ctypes.data_as(ip))
        return out[:n].tolist()

    differ = 0
    for trial in range(20):
        row_ptr = np.array([0, 45, 45, 147, 147], np.int64)               # read 0: pairs with B = 2 (40) and B = 3 (5); read 2: B = 0 (2), B = 1 (100)
        b_flag = np.concatenate([np.full(40, 2), np.full(5, 3), np.full(2, 0), np.full(100, 1)]).astype(np.uint32)
        ab = rng.integers(0, 5, 147).astype(np.int32) * 100
        a_span = np.stack([ab, ab + rng.integers(1, 4, 147).astype(np.int32) * 1000], 1).astype(np.int32)
        sel, a_of = dist.pick_best_pairs(row_ptr, a_span, a_span.copy(), b_flag, 0, 4, np.ones(4, np.uint8), True)
        assert list(a_of) == [0, 0, 0, 0, 2, 2, 2, 2]
        L = (a_span[:, 1] - a_span[:, 0]).astype(np.int64) * 2
        group = {(0, 2): (0, 40), (0, 3): (40, 5), (2, 0): (45, 2), (2, 1): (47, 100)}
        want = []
        for a, keys in ((0, [2, 3]), (2, [0, 1])):
            for bid in umap_order(keys):                                   # the pairs of a read in the hash map's iteration order
                s0, n = group[(a, bid)]
                _, twice = oracle_twice(L[s0:s0 + n])
                want += [s0 + int(twice[0]), s0 + int(twice[1])]
        assert list(sel) == want, (trial, list(sel), want)
        stable = sorted(s0 + int(np.argsort(-L[s0:s0 + n], kind="stable")[1]) for s0, n in ((0, 40), (47, 100)))
        differ += not set(stable) <= set(want)
    assert differ > 0, "no case in which the introsort order differs from a stable sort"


# ---- sharded `hinge layout`: exchanges 5-7 around the host steps of hinge_amd/layout.py -----------------------------------------
LAYOUT_FILES = [".garbage.txt", ".killed.hinges", ".hgraph", ".hinge.list", ".edges.hinges", ".edges.hinges2", ".edges.skipped", ".deadends.txt"]


class OracleLayoutBackend:
    """Stands in for HipLayoutBackend on CPU: ProcessAlignment and GetMatchingPosition overlap by overlap through the CPU oracle
    (both pinned to the reference's compiled code), the selection through the statement-level restatement of
    hinging.cpp:1911-2148 that tests/test_select_gpu.py holds k_select_edges against."""

    def __init__(self, oracle_lib, pile, recs, eff):
        self.lib, self.pile, self.recs, self.eff = oracle_lib, pile, recs, eff
        self.toff = recs.trace_off[:-1][pile.las_index]
        self.tlen = recs.rec["tlen"][pile.las_index]

    def _trace(self, k):
        return self.recs.trace[self.toff[k]:self.toff[k] + self.tlen[k]].astype(np.uint16)

    def matches(self, lo, hi, active, P):
        import ctypes
        from hinge_amd import layout as L
        ip, u16p = ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint16)
        pile, eff = self.pile, self.eff

        def classify(sel, a_of):
            out = np.zeros((len(sel), 10), np.int32)
            for n, (k, a) in enumerate(zip(sel, a_of)):
                b = int(pile.b_flag[k] & 0x7FFFFFFF)
                hdr = np.array([pile.a_span[k, 0], pile.a_span[k, 1], pile.b_span[k, 0], pile.b_span[k, 1], int(pile.b_flag[k] >> 31),
                                eff[a][0], eff[a][1], eff[b][0], eff[b][1]], np.int32)
                tr = self._trace(k)
                self.lib.oracle_process_alignment(hdr.ctypes.data_as(ip), tr.ctypes.data_as(u16p), len(tr), P.aln_threshold, P.theta, P.theta2,
                                                  out[n].ctypes.data_as(ip))
            return out
        return L.block_matches(classify, pile, lo, hi, active, P)

    def matching_positions(self, q_ovl, q_pos):
        import ctypes
        u16p = ctypes.POINTER(ctypes.c_uint16)
        pile = self.pile
        out = np.zeros(len(q_ovl), np.int32)
        for n, (k, pos) in enumerate(zip(q_ovl, q_pos)):
            tr = self._trace(k)
            out[n] = self.lib.oracle_matching_position(int(pile.a_span[k, 0]), int(pile.a_span[k, 1]), int(pile.b_span[k, 0]), int(pile.b_span[k, 1]),
                                                       int(pile.b_flag[k] >> 31), tr.ctypes.data_as(u16p), len(tr), int(pos))
        return out

    def select(self, active, off_fwd, off_bwd, rec, h_off, h_rec, k_off, k_rec, tol, slack):
        from test_select_gpu import reference_selection
        n = len(active)
        names = ("b", "comp", "type", "active", "weight", "eff_bb", "eff_be", "bb", "be")
        mk = lambda j: dict(zip(names, (int(v) for v in rec[j])), gid=int(j))
        fwd = [[mk(j) for j in range(int(off_fwd[i]), int(off_fwd[i + 1]))] for i in range(n)]
        bwd = [[mk(j) for j in range(int(off_bwd[i]), int(off_bwd[i + 1]))] for i in range(n)]
        hinges = [[tuple(int(v) for v in h_rec[q]) for q in range(int(h_off[i]), int(h_off[i + 1]))] for i in range(n)]
        killed = [[tuple(int(v) for v in k_rec[q]) for q in range(int(k_off[i]), int(k_off[i + 1]))] for i in range(n)]
        chosen, hp, skipped = reference_selection(active, fwd, bwd, hinges, killed, tol, slack)
        poison = np.zeros(max(len(rec), 1), np.int32)
        for g in skipped:
            poison[g] += 1
        return chosen, hp, poison[:len(rec)]


def _layout_inputs(wd, n_read):
    from hinge_amd import layout as L
    from hinge_amd.config import IniFile
    eff = np.zeros((n_read, 2), np.int64)
    for line in open(os.path.join(wd, "G.mas")):
        i, s, e = (int(t) for t in line.split())
        eff[i] = (s, e)
    maximal = np.zeros(n_read, bool)
    for line in open(os.path.join(wd, "G.max")):
        maximal[int(line)] = True
    P = L.LayoutParams.from_ini(IniFile(os.path.join(wd, "nominal.ini")))
    return eff, maximal, L.read_pairs_file(os.path.join(wd, "G.repeat.txt"), n_read), L.read_pairs_file(os.path.join(wd, "G.hinges.txt"), n_read), P


def _layout_worker(rank, world, port, wd, first, rlen, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from hinge_amd import formats
        from hinge_amd.dist import BlockTable, Exchange, ShardedLayout
        n = len(rlen)
        eff, maximal, repeats, hinges, P = _layout_inputs(wd, n)
        recs = formats.read_las(os.path.join(wd, "G.%d.las" % (rank + 1)))          # this rank's block only
        pile = formats.pileups_from_las(recs, rlen)
        be = OracleLayoutBackend(oracle.oracle_lib(), pile, recs, eff.tolist())
        job = ShardedLayout(be, Exchange(BlockTable(first), torch.device("cpu")), P, eff, maximal, repeats, hinges)
        files = job.step()
        bad = [f for f in LAYOUT_FILES if files[f] != open(os.path.join(wd, "G" + f)).read().split("\n")[:-1]]
        assert not bad, "rank %d: %s differ from the oracle's" % (rank, bad)
        ret[rank] = len(files[".edges.hinges"])
    finally:
        dist.destroy_process_group()


def test_sharded_layout_files(oracle_lib, tmp_path):
    """world 2: each rank classifies the active x active pairs of its own block, the matches and matching positions are gathered,
    every rank runs the hinge bookkeeping, selects the edges of its own reads; the eight files equal the oracle's --mlas run."""
    import dataclasses
    from hinge_amd import synth
    from conftest import write_ini
    d = synth.generate(dataclasses.replace(synth.CONFIGS["tiny_mlas"], n_blocks=2))
    wd = str(tmp_path / "data")
    synth.write_dataset(d, wd, "G")
    write_ini(os.path.join(wd, "nominal.ini"))
    assert run_in(wd, oracle_lib.oracle_filter, b"G", b"G", 1, b"G", b"nominal.ini", b"") == 0
    assert run_in(wd, oracle_lib.oracle_maximal, b"G", b"G", 1, b"G", b"nominal.ini") == 0
    assert run_in(wd, oracle_lib.oracle_layout, b"G", b"G", 1, b"G", b"G", b"nominal.ini") == 0
    n_edges = len(open(os.path.join(wd, "G.edges.hinges")).read().split("\n")) - 1
    assert n_edges > 50 and os.path.getsize(os.path.join(wd, "G.hgraph")) > 0
    port = conftest.free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_layout_worker, args=(2, port, wd, list(d.block_first), d.rlen, ret), nprocs=2, join=True)
    assert dict(ret) == {0: n_edges, 1: n_edges}
