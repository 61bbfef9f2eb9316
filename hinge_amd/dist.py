"""Multi-GPU `hinge filter` and `hinge maximal`: one process per GPU, reads sharded by DAZZ_DB block (contiguous
read-id ranges, the unit HPC.daligner/LAmerge already emit one sorted .las for: filter.cpp:35-63,474), with
the path's real exchange steps as RCCL collectives over xGMI (torch.distributed backend "nccl";
"gloo" on CPU in the tests):

  1. per-read mean coverage  -> median / MIN_COV            (filter.cpp:642-678, a global reduction)
  2. per-read masks          -> maskvec[B] of every B read  (filter.cpp:778-787 feeding :883-890)
  3. hinge lists             -> global hinge list on rank 0 (the input of `hinge layout`)
  4. containment candidates  -> maximal-read mask on every rank (maximal.cpp:780-858, ShardedMaximal below)

All three are small (4-12 bytes per read); they are latency-bound, never xGMI-bandwidth-bound, so a
plain all-gather of equal-sized padded shards is used rather than anything bucketed.

Two semantics, both exact:
  * "merged": the result equals the reference run on ONE merged .las over all blocks (global median,
    every mask visible).
  * "mlas":   the result equals the reference's sequential `--mlas` loop: MIN_COV is a running max
    over parts (prefix max over ranks) and, while part p is processed, masks of reads in later parts
    are still (0,0).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import os

import numpy as np
import torch
import torch.distributed as dist

MEAN_SENTINEL = -(2 ** 31)


@dataclass
class BlockTable:
    """Read-id range [first[k], first[k+1]) of every block; block k lives on rank k % world."""

    first: List[int]

    @property
    def n_blocks(self) -> int:
        return len(self.first) - 1

    @property
    def n_reads(self) -> int:
        return self.first[-1]

    def size(self, k: int) -> int:
        return self.first[k + 1] - self.first[k]

    @property
    def max_size(self) -> int:
        return max(self.size(k) for k in range(self.n_blocks))


def _needs_host_staging(device: torch.device, group=None) -> bool:
    """gloo moves host memory: with device tensors (several test processes sharing ONE GPU, where RCCL refuses to form a
    communicator) every collective goes through a host copy.  Never the case on the product path (backend "nccl" = RCCL)."""
    return dist.is_initialized() and device.type == "cuda" and dist.get_backend(group) == "gloo"


def _all_reduce_sum(t: torch.Tensor, group, staged: bool, async_op: bool = False):
    if staged:
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
        return None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def _all_gather_in_place(table: torch.Tensor, lo: int, hi: int, group, staged: bool, async_op: bool = False):
    """Every rank has filled rows [lo, hi) of `table` (equal chunks, rank r's at r * (hi - lo)); afterwards all rows are
    everywhere: ONE all_gather_into_tensor on the table itself."""
    if staged:
        h = torch.empty(table.shape, dtype=table.dtype)
        dist.all_gather_into_tensor(h, table[lo:hi].cpu(), group=group)
        table.copy_(h)
        return None
    return dist.all_gather_into_tensor(table, table[lo:hi], group=group, async_op=async_op)


def _host_all_gather(vals: Sequence[int], device: torch.device, group=None, force: bool = True) -> List[List[int]]:
    """A few host integers per rank -> the same list of per-rank rows on every rank (a host round trip: set-up and status
    only, never inside the per-step kernel chain)."""
    vals = [int(v) for v in vals]
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return [vals]
    world = dist.get_world_size(group)
    dev = torch.device("cpu") if _needs_host_staging(device, group) else device
    t = torch.tensor(vals, dtype=torch.int64, device=dev)
    out = torch.empty(world * len(vals), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(out, t, group=group)
    flat = out.cpu().tolist()
    return [[int(x) for x in flat[k * len(vals):(k + 1) * len(vals)]] for k in range(world)]


class Exchange:
    """The collectives of the path on equal-sized padded shards (one block per rank)."""

    def __init__(self, blocks: BlockTable, device: torch.device, group=None):
        self.blocks = blocks
        self.device = device
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        assert blocks.n_blocks == self.world, "one block per rank"
        self.S = blocks.max_size
        # HINGE_FORCE_COLLECTIVES=1 runs the collectives even with one rank (exercises the RCCL path on a 1-GPU box)
        self.force = dist.is_initialized() and os.environ.get("HINGE_FORCE_COLLECTIVES", "0") == "1"
        self.staged = _needs_host_staging(device, group)
        self._buffers = {}

    @property
    def my_range(self) -> Tuple[int, int]:
        return self.blocks.first[self.rank], self.blocks.first[self.rank + 1]

    def all_gather_rows(self, table: torch.Tensor, async_op: bool = False):
        """table[n_reads, ...]: every rank has filled the rows of its own block; on return every rank
        holds all rows.  Equal blocks: ONE in-place all_gather_into_tensor on the table itself.  Unequal blocks:
        copy-in to a cached padded shard, all-gather, one indexed copy-out (buffers and index maps are built once).
        async_op: with equal blocks the collective is only enqueued and its work handle returned (wait() orders the
        calling stream behind it); otherwise None is returned and the exchange is complete."""
        if self.world == 1 and not self.force:
            return None
        lo, hi = self.my_range
        if all(self.blocks.size(k) == self.S for k in range(self.world)):
            # in place: rank r's rows sit at r * S
            w = _all_gather_in_place(table, lo, hi, self.group, self.staged, async_op)
            return w if async_op else None
        key = (table.dtype, tuple(table.shape[1:]), table.device)
        buf = self._buffers.get(key)
        if buf is None:
            tail = tuple(table.shape[1:])
            send = torch.zeros((self.S,) + tail, dtype=table.dtype, device=table.device)
            recv = torch.empty((self.world * self.S,) + tail, dtype=table.dtype, device=table.device)
            src = np.concatenate([k * self.S + np.arange(self.blocks.size(k)) for k in range(self.world) if k != self.rank] or [np.zeros(0, np.int64)])
            dst = np.concatenate([np.arange(self.blocks.first[k], self.blocks.first[k + 1]) for k in range(self.world) if k != self.rank] or [np.zeros(0, np.int64)])
            buf = (send, recv, torch.from_numpy(src.astype(np.int64)).to(table.device), torch.from_numpy(dst.astype(np.int64)).to(table.device))
            self._buffers[key] = buf
        send, recv, src, dst = buf
        send[: hi - lo].copy_(table[lo:hi])
        if self.staged:
            h = torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_gather_into_tensor(h, send.cpu(), group=self.group)
            recv.copy_(h)
        else:
            dist.all_gather_into_tensor(recv, send, group=self.group)
        if src.numel():
            table.index_copy_(0, dst, recv.index_select(0, src))
        return None

    def all_reduce_sum(self, t: torch.Tensor, async_op: bool = False):
        if self.world == 1 and not self.force:
            return None
        return _all_reduce_sum(t, self.group, self.staged, async_op)

    def all_gather_scalar(self, v: int) -> List[int]:
        return [r[0] for r in self.all_gather_ints([v])]

    def all_gather_ints(self, vals: Sequence[int]) -> List[List[int]]:
        """A few host integers per rank -> the same list of per-rank rows on every rank (a host round trip: only used where the
        host has to decide something, never inside the per-step kernel chain)."""
        vals = [int(v) for v in vals]
        if self.world == 1 and not self.force:
            return [vals]
        dev = torch.device("cpu") if self.staged else self.device
        t = torch.tensor(vals, dtype=torch.int64, device=dev)
        out = torch.empty(self.world * len(vals), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(out, t, group=self.group)
        flat = out.cpu().tolist()
        return [[int(x) for x in flat[k * len(vals):(k + 1) * len(vals)]] for k in range(self.world)]

    def gather_lists(self, rows: torch.Tensor, count: int) -> Optional[torch.Tensor]:
        """Variable-length int32 row lists (e.g. (read, pos, type) hinges) -> concatenated in rank order
        on every rank: counts all-gather, then one padded all-gather."""
        if self.world == 1 and not self.force:
            return rows[:count]
        counts = self.all_gather_scalar(count)
        cap = max(max(counts), 1)
        send = torch.zeros((cap, rows.shape[1]), dtype=rows.dtype, device=rows.device)
        send[:count] = rows[:count]
        recv = torch.empty((self.world * cap, rows.shape[1]), dtype=rows.dtype, device=rows.device)
        if self.staged:
            h = torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_gather_into_tensor(h, send.cpu(), group=self.group)
            recv.copy_(h)
        else:
            dist.all_gather_into_tensor(recv, send, group=self.group)
        return torch.cat([recv[k * cap: k * cap + counts[k]] for k in range(self.world)], dim=0)

    def barrier(self):
        if self.world > 1:
            dist.barrier(group=self.group)


def mlas_min_cov(ini_min_cov: int, cov_ests: Sequence[int], est_cov_override: int = 0) -> List[int]:
    """MIN_COV seen by each part of the sequential --mlas loop: a running max (filter.cpp:671-678)."""
    out, m = [], ini_min_cov
    for c in cov_ests:
        if est_cov_override != 0:
            c = est_cov_override
        q = int(c / 3) if c >= 0 else -int(-c / 3)    # C division truncates toward zero
        if m < q:
            m = q
        out.append(m)
    return out


class ShardedFilter:
    """One rank's share of a sharded `hinge filter` pass.

    `backend` does the per-block compute; the product backend is HipBackend below (HIP kernels through
    the C ABI).  Tests may pass another object with the same four methods to exercise the exchange
    logic on CPU with gloo.
    """

    def __init__(self, backend, exchange: Exchange, mode: str = "merged", median: str = "hist"):
        assert mode in ("merged", "mlas")
        assert median in ("hist", "gather")
        self.b = backend
        self.x = exchange
        self.mode = mode
        # "hist": exchange 1 is a 16 KiB all-reduce of a 4096-bin histogram of the mean coverages (exact while every
        # mean is in [0, 4096); otherwise the backend reports it and "gather" - all-gather of 4 bytes per read - is the way)
        self.median = median
        n = exchange.blocks.n_reads
        dev = exchange.device
        self.mean_cov = torch.full((n,), MEAN_SENTINEL, dtype=torch.int32, device=dev)
        self.mask = torch.zeros((n, 2), dtype=torch.int32, device=dev)
        self.b.attach(self.mean_cov, self.mask)
        # The histogram form of exchange 1 is exact while every mean coverage lies in [0, 4096).  A mean is an average of
        # cutoff-0 coverage values, each between 0 and the pile-up's size, so the largest pile-up over all ranks decides it
        # once, here, instead of a device flag the host would have to read back in every step.
        if self.median == "hist" and self.mode == "merged" and max(self.x.all_gather_scalar(self.b.max_pileup())) >= 4096:
            self.median = "gather"

    def step(self, fetch_hinges: bool = True, check: bool = True, _retried: bool = False):
        """One pass.  check = True ends with the status exchange: every rank learns whether any rank overflowed a device
        buffer (all regrow and the step is run again, together) or hit an input the reference is undefined on (all raise).
        A caller that times a chain of steps passes check = False and calls ctx.check() after the chain."""
        x, b = self.x, self.b
        lo, hi = x.my_range
        b.begin()
        b.stats()                                   # fills mean_cov[lo:hi]
        if self.mode == "merged" and self.median == "hist" and (x.world > 1 or x.force):
            h = b.median_hist(lo, hi - 1)           # this block's histogram of mean coverages (device)
            x.all_reduce_sum(h)                     # exchange 1: 16 KiB
            b.median_from_hist(h)                   # same global median on every rank (device side)
        elif self.mode == "merged":                 # one rank, or the general form of exchange 1
            x.all_gather_rows(self.mean_cov)
            b.median(0, x.blocks.n_reads - 1)
        else:
            try:                                    # per-part median (host scalar); a part the reference is undefined on
                est, bad = b.median_fetch(lo, hi - 1), 0   # (no read >= 5000 bp) must stop every rank, not hang the others
            except Exception as ex:                 # noqa: BLE001 - re-raised below on every rank
                est, bad, first_error = 0, 1, ex
            rows_ = x.all_gather_ints([est, bad])   # exchange 1 (16 bytes per rank)
            if any(r[1] for r in rows_):
                if bad:
                    raise first_error
                raise RuntimeError("sharded filter: rank(s) %s cannot estimate the coverage of their part" % [k for k, r in enumerate(rows_) if r[1]])
            b.set_min_cov(mlas_min_cov(b.ini_min_cov, [r[0] for r in rows_], b.est_cov)[x.rank])
        b.mask_annotate()                           # fills mask[lo:hi]
        x.all_gather_rows(self.mask)                # exchange 2
        if self.mode == "mlas" and hi < x.blocks.n_reads:
            self.mask[hi:] = 0                      # later parts are not masked yet when part p runs
        b.hinges()
        if check:
            code = b.status_code()                  # 0, or the HINGE_E_* of this rank's pass
            codes = x.all_gather_scalar(code)
            if any(c == -3 for c in codes) and not _retried:     # HINGE_E_CAPACITY somewhere: everyone regrows, everyone reruns
                b.regrow()
                return self.step(fetch_hinges, check, True)
            if any(c != 0 for c in codes):
                b.raise_status(codes)
        if not fetch_hinges:
            return None
        # .hinges.txt stops before the part's last A read (`i < r_end`, filter.cpp:1091): one merged .las loses the hinges of
        # the global last read only, the --mlas loop those of every part's last read
        drop_last = self.mode == "mlas" or x.rank == x.world - 1
        rows, count = b.hinge_rows(drop_last)       # (read, pos, type) int32 rows on the device
        return x.gather_lists(rows, count)          # exchange 3


def step_pipelined(jobs: Sequence["ShardedFilter"]) -> None:
    """One pass of several independent sharded jobs of this rank (e.g. the .las parts it holds; every job is its own sharded
    run over the same ranks), without status exchange or hinge fetch - the chain bench.py times.

    A job's two exchanges sit between its kernels (stats -> exchange 1 -> median -> mask/annotate -> exchange 2 -> hinges), so run
    job by job every collective is exposed: eight latency-bound RCCL calls per four-part step, about as long as a part's kernels.
    Here the jobs are software-pipelined: every exchange is enqueued asynchronously as soon as its input exists and waited for
    (a stream-side wait, no host block with RCCL) only where its output is read, so exchange 1 of job j runs under the stats sweep
    of job j + 1 and exchange 2 under the next job's mask/annotate kernel; only the last one of each kind is exposed.
    Same kernels, same collectives, same results as [j.step(False, False) for j in jobs] (tests/test_dist_gloo.py).
    Jobs in "mlas" mode or with the all-gather form of exchange 1 (host decisions in between) are run one after the other.
    No status is read inside: a device buffer that overflows in this chain is only seen by the next ctx.check().  Callers run one
    synchronous pass first (ShardedFilter.step(check=True), or the staged synchronous entry points as bench.py does) so that the
    buffers are sized before the unchecked chain; PartBatch.settle() is that warm-up for the batched form."""
    if not jobs:
        return
    x0 = jobs[0].x
    if not (all(j.mode == "merged" and j.median == "hist" for j in jobs) and (x0.world > 1 or x0.force)):
        for j in jobs:
            j.step(fetch_hinges=False, check=False)
        return
    pending = []
    for j in jobs:
        lo, hi = j.x.my_range
        j.b.begin()
        j.b.stats()
        h = j.b.median_hist(lo, hi - 1)
        pending.append((h, j.x.all_reduce_sum(h, async_op=True)))      # exchange 1
    gathers = []
    for j, (h, w) in zip(jobs, pending):
        if w is not None:
            w.wait()
        j.b.median_from_hist(h)
        j.b.mask_annotate()
        gathers.append(j.x.all_gather_rows(j.mask, async_op=True))     # exchange 2
    for j, w in zip(jobs, gathers):
        if w is not None:
            w.wait()
        j.b.hinges()


class PartBatch:
    """The R .las parts a rank holds resident (R independent sharded jobs over the same ranks, "merged" semantics each), run as
    ONE chain with the exchanges of all parts **batched**: per step one all-reduce (the R coverage histograms, R x 16 KiB, one
    contiguous tensor) and one all-gather per gather group (the masks of the group's parts, 8 B per read) - 2 collectives with
    `gather_groups` = 1 instead of the 2 R of a part-by-part chain.  Every collective is a latency-bound RCCL launch (and, at
    N = 8, a ring over xGMI links), so their NUMBER is what costs.

    Id space.  For the in-place all-gather a rank's rows must be contiguous, so the parts of one gather group share one
    table: group g with J parts has world x J x S rows, part j of rank r owns rows [(r J + j) S, (r J + j + 1) S).  S = the
    largest block over all ranks and parts; ids behind a block's reads are padding (length 0, no overlaps).  A part's B ids
    are ids of its own table: `global_ids(p, owner_rank, local)` maps (rank that holds the B read's block, index inside the
    block) to them.

    With gather_groups = 2 the first group's all-gather runs under the second group's mask/annotate kernels and the second
    one under the first group's hinge kernels (asynchronous enqueue, stream-side waits)."""

    def __init__(self, n_parts: int, block_size: int, device: torch.device, group=None, gather_groups: int = 1):
        self.R, self.S, self.device, self.group = int(n_parts), int(block_size), device, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.force = dist.is_initialized() and os.environ.get("HINGE_FORCE_COLLECTIVES", "0") == "1"
        self.collectives = self.world > 1 or self.force
        self.staged = _needs_host_staging(device, group)
        G = max(1, min(int(gather_groups), self.R))
        cut = [round(k * self.R / G) for k in range(G + 1)]
        self.groups = [list(range(cut[k], cut[k + 1])) for k in range(G)]
        self.slot = {}                       # part -> (group index, index inside the group, parts in the group)
        for gi, g in enumerate(self.groups):
            for j, p in enumerate(g):
                self.slot[p] = (gi, j, len(g))
        self.masks = [torch.zeros((self.world * len(g) * self.S, 2), dtype=torch.int32, device=device) for g in self.groups]
        self.means = [torch.full((self.world * len(g) * self.S,), MEAN_SENTINEL, dtype=torch.int32, device=device) for g in self.groups]
        self.hist = torch.zeros((self.R, 4096 + 2), dtype=torch.int32, device=device)
        self.backends: List[Optional[object]] = [None] * self.R
        self.after_gather = None     # test rigs only (bench.py's fault injection): called with the group index after exchange 2; never set by product code
        self.one_sweep = os.environ.get("HINGE_ONE_SWEEP", "1") != "0"     # 0: the two-sweep pass of rounds 1-3 (k_cov_stats first)

    # ---- id space ---------------------------------------------------------------------------------------------
    def n_ids(self, p: int) -> int:
        gi, j, J = self.slot[p]
        return self.world * J * self.S

    def id_base(self, p: int, rank: Optional[int] = None) -> int:
        gi, j, J = self.slot[p]
        return ((self.rank if rank is None else rank) * J + j) * self.S

    def global_ids(self, p: int, owner_rank, local):
        gi, j, J = self.slot[p]
        return (np.asarray(owner_rank, dtype=np.int64) * J + j) * self.S + np.asarray(local, dtype=np.int64)

    def mask_table(self, p: int) -> torch.Tensor:
        return self.masks[self.slot[p][0]]

    def set_backend(self, p: int, backend) -> None:
        gi = self.slot[p][0]
        backend.attach(self.means[gi], self.masks[gi])
        self.backends[p] = backend

    def max_pileup(self) -> int:
        local = max(b.max_pileup() for b in self.backends)
        return max(r[0] for r in self._gather_ints([local]))

    # ---- collectives ------------------------------------------------------------------------------------------
    def _gather_ints(self, vals: Sequence[int]) -> List[List[int]]:
        return _host_all_gather(vals, self.device, self.group, force=self.collectives)

    def _gather_group(self, gi: int, async_op: bool = True):
        J = len(self.groups[gi])
        lo = self.rank * J * self.S
        w = _all_gather_in_place(self.masks[gi], lo, lo + J * self.S, self.group, self.staged, async_op=async_op)
        return w if async_op else None

    # ---- one pass over all parts ------------------------------------------------------------------------------
    def step(self) -> None:
        """No host synchronisation and no status exchange inside (call status() after a chain of steps)."""
        B = self.backends
        batched = hasattr(B[0], "hinges_batch")       # the product backend: the latency-bound kernels take all parts per launch
        if batched and self.one_sweep:
            # the one-sweep pass (include/hinge_hip.h): per part ONE sweep over the pile-ups with a predicted MIN_COV that also
            # yields the coverage sums, the exact median as verification (exchange 1 sums its histograms over the ranks), then
            # the ~1 % guard-band reads with the exact MIN_COV
            for b in B:
                b.begin()
            if not self.collectives:
                B[0].sweep_batch(B)
                B[0].finish_batch(B)
                B[0].hinges_batch(B)
                return
            B[0].sweep_batch(B, hist=self.hist)
            _all_reduce_sum(self.hist, self.group, self.staged, async_op=False)          # exchange 1
            B[0].median_from_hist_batch(B, self.hist)                                    # ... and the verification
            pending = []
            last = len(self.groups) - 1
            for gi, g in enumerate(self.groups):
                B[g[0]].finish_batch([B[p] for p in g])
                pending.append(self._gather_group(gi, async_op=gi != last))              # exchange 2
            for gi, g in enumerate(self.groups):
                if pending[gi] is not None:
                    pending[gi].wait()
                if self.after_gather is not None:
                    self.after_gather(gi)
                B[g[0]].hinges_batch([B[p] for p in g])
            return
        if not self.collectives:              # one rank: no exchange; the parts only share their launches
            for b in B:
                b.begin()
                b.stats()
            if batched:
                B[0].median_batch(B)
            else:
                for p, b in enumerate(B):
                    lo = self.id_base(p)
                    b.median(lo, lo + self.S - 1)
            # (last part first: its span copy is what the statistics sweeps left in the 256 MiB Infinity Cache; -1.5 us per launch)
            for b in (reversed(B) if os.environ.get("HINGE_K2_REVERSE", "1") == "1" else B):
                b.mask_annotate()
            if batched:
                B[0].hinges_batch(B)
            else:
                for b in B:
                    b.hinges()
            return
        for b in B:
            b.begin()
            b.stats()
        if batched:
            B[0].median_batch(B, hist=self.hist)
        else:
            for p, b in enumerate(B):
                lo = self.id_base(p)
                b.median_hist(lo, lo + self.S - 1, out=self.hist[p])
        # exchange 1, all parts at once.  Not asynchronous: nothing can run under it, and a synchronous collective is enqueued
        # on the calling stream's order without the event hand-over between streams an asynchronous handle costs
        _all_reduce_sum(self.hist, self.group, self.staged, async_op=False)
        if batched:
            B[0].median_from_hist_batch(B, self.hist)                                   # one launch for all parts
        pending = []
        last = len(self.groups) - 1
        for gi, g in enumerate(self.groups):
            for p in g:
                if not batched:
                    B[p].median_from_hist(self.hist[p])
                B[p].mask_annotate()
            # exchange 2, one per group; asynchronous only where the next group's kernels can run under it
            pending.append(self._gather_group(gi, async_op=gi != last))
        for gi, g in enumerate(self.groups):
            if pending[gi] is not None:
                pending[gi].wait()
            if self.after_gather is not None:   # (fault injection of the test rig: a broken exchange 2 must not go unnoticed, tests/test_dist_gpu.py)
                self.after_gather(gi)
            if batched:
                B[g[0]].hinges_batch([B[p] for p in g])
            else:
                for p in g:
                    B[p].hinges()

    def corrupt_foreign_rows(self, gi: int) -> None:
        """Fault injection for test rigs (bench.py sets `after_gather` to this under HINGE_TEST_CORRUPT_GATHER=1): zero the other
        ranks' mask rows of gather group gi, as a broken all-gather would leave them."""
        J = len(self.groups[gi])
        own = slice(self.rank * J * self.S, (self.rank + 1) * J * self.S)
        keep = self.masks[gi][own].clone()
        self.masks[gi].zero_()
        self.masks[gi][own] = keep

    def settle(self, max_rounds: int = 4) -> None:
        """Whole steps until no rank reports a full device buffer (HINGE_E_CAPACITY anywhere: every rank regrows, every rank
        reruns - the buffers' sizes depend on the exchanged masks, so they are only known after a real step).  Callers that
        time chains of step() run this once first."""
        for _ in range(max_rounds):
            self.step()
            codes = [b.status_code() for b in self.backends]
            rows = self._gather_ints(codes)
            flat = [c for r in rows for c in r]
            if all(c == 0 for c in flat):
                return
            if any(c not in (0, -3) for c in flat):
                self._raise(codes, rows)
            for b in self.backends:
                b.regrow()
        raise RuntimeError("sharded filter: device buffers still too small after %d rounds" % max_rounds)

    def _raise(self, codes, rows):
        for b, c in zip(self.backends, codes):
            if c != 0:
                b.raise_status(rows)
        raise RuntimeError("sharded filter: another rank failed (status codes per rank and part: %s)" % (rows,))

    def table_checksums(self) -> List[int]:
        """One position-weighted checksum per mask table: equal on every rank after exchange 2."""
        out = []
        for m in self.masks:
            v = m.reshape(-1).to(torch.int64)
            w = (torch.arange(v.numel(), device=v.device, dtype=torch.int64) % 65521) + 1
            out.append(int((v * w).sum().item()))
        return out

    def status(self) -> None:
        """After a chain of steps: every rank learns every rank's status codes; raises on all ranks if any pass failed."""
        codes = [b.status_code() for b in self.backends]
        rows = self._gather_ints(codes)
        if any(c != 0 for r in rows for c in r):
            self._raise(codes, rows)

    def hinge_lists(self, drop_global_last: bool = True) -> List[torch.Tensor]:
        """Exchange 3, all parts at once: per part the (read, pos, type) rows of every rank in rank order, on every rank.
        `.hinges.txt` stops before the last A read of a merged .las (`i < r_end`, filter.cpp:1091): the last rank drops it."""
        mine = [b.hinge_rows(drop_global_last and self.rank == self.world - 1) for b in self.backends]
        if not self.collectives:
            return [rows[:n] for rows, n in mine]
        counts = self._gather_ints([n for _, n in mine])                                 # [rank][part]
        cap = max(1, max(sum(r) for r in counts))
        send = torch.zeros((cap, 3), dtype=torch.int32, device=self.device)
        at = 0
        for rows, n in mine:
            send[at:at + n] = rows[:n]
            at += n
        recv = torch.empty((self.world * cap, 3), dtype=torch.int32, device=self.device)
        if self.staged:
            h = torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_gather_into_tensor(h, send.cpu(), group=self.group)
            recv.copy_(h)
        else:
            dist.all_gather_into_tensor(recv, send, group=self.group)
        out = []
        for p in range(self.R):
            parts = []
            for r in range(self.world):
                s0 = r * cap + sum(counts[r][:p])
                parts.append(recv[s0:s0 + counts[r][p]])
            out.append(torch.cat(parts, dim=0))
        return out


def resident_batch(parts, params, device: torch.device, gather_groups: int = 1, pad: int = 0, group=None):
    """A rank's resident parts -> (PartBatch, contexts): what an N-GPU run sets up once before its passes (bench.py, the GPU
    tests).  parts: objects with rlen, row_ptr (0-based over the own block), a_span, b_span, b_owner, b_local (B read =
    (rank that holds its block, index inside the block)), comp, last_a - hinge_amd/benchsets.RankPart.  Every block gets the
    same number of read ids (the largest block's over all ranks and parts, + pad); a collective call on every rank."""
    from . import capi
    S = max(max(r) for r in _host_all_gather([rp.n_reads for rp in parts], device, group)) + int(pad)
    batch = PartBatch(len(parts), S, device, group=group, gather_groups=gather_groups)
    ctxs = []
    for p, rp in enumerate(parts):
        n_ids, lo = batch.n_ids(p), batch.id_base(p)
        hi = lo + rp.n_reads                   # real reads of this rank's block: [lo, hi); ids [hi, lo + S) are padding
        rlen_t = torch.zeros(n_ids, dtype=torch.int32, device=device)
        rlen_t[lo:hi] = torch.from_numpy(np.ascontiguousarray(rp.rlen, dtype=np.int32)).to(device)
        if batch.collectives:
            _all_reduce_sum(rlen_t, group, batch.staged)
        row_ptr = np.zeros(n_ids + 1, np.int64)
        row_ptr[lo:hi + 1] = rp.row_ptr
        row_ptr[hi + 1:] = rp.row_ptr[-1]
        b_flag = batch.global_ids(p, rp.b_owner, rp.b_local).astype(np.uint32) | (np.asarray(rp.comp, dtype=np.uint32) << np.uint32(31))
        # what the ingest hands over besides the columns (hinge_amd/host/host_common.h LasPart::load does the same per record)
        span16, max_pile, in_range = capi.pack_spans(rp.row_ptr, rp.a_span, rp.rlen)
        bins = np.zeros(S, np.int32)                                   # (ids behind the block's reads: empty pile-ups)
        bins[:rp.n_reads] = capi.pile_bins(rp.row_ptr, rp.a_span, rp.rlen, int(params.reso))
        tens = (torch.from_numpy(row_ptr).to(device), torch.from_numpy(np.ascontiguousarray(rp.a_span, dtype=np.int32)).to(device),
                torch.from_numpy(np.ascontiguousarray(rp.b_span, dtype=np.int32)).to(device), torch.from_numpy(b_flag.view(np.int32)).to(device),
                None if span16 is None else torch.from_numpy(span16.view(np.int32)).to(device))
        ctx = capi.Context(device.index or 0)
        backend = HipBackend(ctx, params, rlen_t.cpu().numpy(), None, lo, lo + S - 1, tens[0], tens[1], tens[2], tens[3], span16=tens[4],
                             facts=(max_pile, in_range), last_a=lo + rp.last_a, coverage_out=True, pile_bins=bins)
        batch.set_backend(p, backend)
        ctxs.append(ctx)
    if batch.max_pileup() >= 4096:
        raise ValueError("PartBatch uses the histogram form of exchange 1: every mean coverage must be below 4096")
    return batch, ctxs


class HipBackend:
    """Per-block compute through libhinge_hip (HIP kernels); tensors are torch CUDA tensors."""

    def __init__(self, ctx, params, rlen: np.ndarray, qv_mask: Optional[np.ndarray], r_begin: int, r_end: int,
                 row_ptr: torch.Tensor, a_span: torch.Tensor, b_span: torch.Tensor, b_flag: torch.Tensor,
                 span16: Optional[torch.Tensor] = None, facts: Optional[Tuple[int, bool]] = None, last_a: Optional[int] = None,
                 coverage_out: bool = False, pile_bins: Optional[np.ndarray] = None):
        """row_ptr ... b_flag: device tensors (adopted).  facts = (max_pile, spans_in_range) and span16 (device uint32/int32
        [n_ovl + capi.span16_pad()], or None) as the ingest produced them (capi.pack_spans): without facts the library sweeps the
        spans itself (k_pileup_facts).  last_a: A read of the part's last .las record (default r_end)."""
        self.ctx, self.p = ctx, params
        self.ini_min_cov = int(params.min_cov)
        self.est_cov = int(params.est_cov)
        self.r_begin, self.r_end = r_begin, r_end
        self.last_a = r_end if last_a is None else int(last_a)
        self._hist = None
        self._bound_stream = torch.cuda.current_stream().cuda_stream
        ctx.set_stream(self._bound_stream)
        ctx.set_reads(rlen, qv_mask)
        self._tensors = (row_ptr, a_span, b_span, b_flag, span16)
        if facts is None:
            ctx.set_pileups(r_begin, r_end, row_ptr, a_span, b_span, b_flag, n_ovl=int(b_flag.shape[0]), on_device=True)
        else:
            ctx.set_pileups_packed(r_begin, r_end, row_ptr, a_span, b_span, b_flag, span16, facts[0], facts[1], n_ovl=int(b_flag.shape[0]), on_device=True)
        if pile_bins is not None:     # the ingest's per-read bin counts (capi.pile_bins): no device sweep before the one-sweep pass
            ctx.set_pile_bins(np.ascontiguousarray(pile_bins, dtype=np.int32), int(params.reso))
        ctx.coverage_out(coverage_out)
        ctx.set_min_cov(self.ini_min_cov)

    def max_pileup(self) -> int:
        return self.ctx.pileup_facts()[0]

    def status_code(self) -> int:
        """0, or the HINGE_E_* of this rank's pass.  The error is cached: check() may clear the device status."""
        from .capi import HingeError
        self._last_error = None
        try:
            self.ctx.check()
            return 0
        except HingeError as ex:
            self._last_error = ex
            return ex.code

    def raise_status(self, codes):
        """Raise this rank's own error when it has one (from the cache, without asking the device again), else say which did."""
        if getattr(self, "_last_error", None) is not None:
            raise self._last_error
        raise RuntimeError("sharded filter: another rank failed (status codes per rank: %s)" % (codes,))

    def regrow(self):
        """After HINGE_E_CAPACITY: the synchronous entry points rerun their stage on this rank's data until the device
        buffers are large enough (annotation buffer, exact-path queue and arena); the caller then repeats the step."""
        self.ctx.filter_mask_annotate(self.p)
        self.ctx.filter_hinges(self.p)

    def attach(self, mean_cov: torch.Tensor, mask: torch.Tensor):
        self.mean_cov, self.mask = mean_cov, mask
        self.ctx.attach_mean_cov(mean_cov)
        self.ctx.attach_mask_table(mask)

    def begin(self):
        # the library's launches and the collectives' stream-side waits must sit on ONE stream: bind whatever torch's current
        # stream is NOW (a caller may step under another torch.cuda.stream() than the one this backend was built under)
        cur = torch.cuda.current_stream().cuda_stream
        if cur != self._bound_stream:
            self.ctx.set_stream(cur)
            self._bound_stream = cur
        self.ctx.set_min_cov(self.ini_min_cov)    # stream-ordered 4-byte set, no host sync

    def stats(self):
        self.ctx.filter_stats(self.p)

    def stats_median(self, out: Optional[torch.Tensor] = None):
        """The statistics sweep and the median of this block's own reads in ONE launch: MIN_COV updated on the device (out None),
        or the block's histogram written to `out` (int32[4096 + 2] device row) for the all-reduce over ranks."""
        self.ctx.filter_stats_median(self.p, out)

    def median(self, lo: int, hi: int):
        self.ctx.filter_median(self.p, lo, hi, fetch=False)

    def median_fetch(self, lo: int, hi: int) -> int:
        return int(self.ctx.filter_median(self.p, lo, hi, fetch=True).cov_est)

    def median_hist(self, lo: int, hi: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """out: a contiguous int32[4096 + 2] device row to fill instead of this backend's own (PartBatch: one row per part of
        ONE tensor, so that a single all-reduce serves all of them)."""
        if out is None:
            if self._hist is None:
                self._hist = torch.zeros(4096 + 2, dtype=torch.int32, device=self.mean_cov.device)
            out = self._hist
        self.ctx.filter_median_hist(self.p, lo, hi, out)
        return out

    def median_from_hist(self, hist: torch.Tensor):
        self.ctx.filter_median_from_hist(self.p, hist)

    MEDIAN_BATCH_MAX, HINGE_BATCH_MAX = 16, 8      # parts per launch the library takes (MED_BATCH_MAX, HINGE_BATCH_MAX)

    def median_batch(self, backends, hist: Optional[torch.Tensor] = None):
        """Every part's median over its own reads in one launch (per 16 parts); hist ([R, 4096 + 2], contiguous): the histogram form."""
        from . import capi
        for k in range(0, len(backends), self.MEDIAN_BATCH_MAX):
            chunk = backends[k:k + self.MEDIAN_BATCH_MAX]
            capi.median_batch([b.ctx for b in chunk], self.p, None if hist is None else hist[k:], 0 if hist is None else int(hist.stride(0)))

    def sweep_batch(self, backends, hist: Optional[torch.Tensor] = None):
        """One-sweep pass, first half, all parts (per 16): prediction, sweep, verifying median (hist: its histogram form)."""
        from . import capi
        for k in range(0, len(backends), self.MEDIAN_BATCH_MAX):
            capi.sweep_batch_async([b.ctx for b in backends[k:k + self.MEDIAN_BATCH_MAX]], self.p, None if hist is None else hist[k:],
                                   0 if hist is None else int(hist.stride(0)))

    def finish_batch(self, backends):
        from . import capi
        for k in range(0, len(backends), self.MEDIAN_BATCH_MAX):
            capi.finish_batch_async([b.ctx for b in backends[k:k + self.MEDIAN_BATCH_MAX]], self.p)

    def hinges_batch(self, backends):
        from . import capi
        for k in range(0, len(backends), self.HINGE_BATCH_MAX):
            capi.hinges_batch_async([b.ctx for b in backends[k:k + self.HINGE_BATCH_MAX]], self.p)

    def median_from_hist_batch(self, backends, hist: torch.Tensor):
        """All of a rank's parts at once (per 16): backend k takes row k of hist ([R, 4096 + 2], contiguous)."""
        from . import capi
        for k in range(0, len(backends), self.MEDIAN_BATCH_MAX):
            capi.median_from_hist_batch([b.ctx for b in backends[k:k + self.MEDIAN_BATCH_MAX]], self.p, hist[k:], int(hist.stride(0)))

    def set_min_cov(self, v: int):
        self.ctx.set_min_cov(v)

    def mask_annotate(self):
        self.ctx.filter_mask_annotate_async(self.p)

    def hinges(self):
        self.ctx.filter_hinges_async(self.p)

    def hinge_rows(self, drop_last: bool = True):
        off, pos, typ, ish = self.ctx.get_annotations()
        reads = np.repeat(np.arange(self.r_begin, self.r_end + 1, dtype=np.int32), np.diff(off).astype(np.int64))
        sel = ish.astype(bool)
        if drop_last:
            sel &= reads != self.last_a
        rows = np.stack([reads[sel], pos[sel], typ[sel]], axis=1).astype(np.int32) if sel.any() else np.zeros((0, 3), np.int32)
        t = torch.from_numpy(np.ascontiguousarray(rows)).to(self.mask.device)
        return t, int(t.shape[0])


# ---- hinge maximal ---------------------------------------------------------------------------------------------
def pick_best_pairs(row_ptr: np.ndarray, a_span: np.ndarray, b_span: np.ndarray, b_flag: np.ndarray, lo: int, hi: int,
                    active: np.ndarray, use_two_matches: bool = True, self_before: Optional[np.ndarray] = None,
                    both_active: bool = False, n_sorts: int = 2) -> Tuple[np.ndarray, np.ndarray]:
    """The overlaps `hinge maximal` (both_active = False, n_sorts = 2: maximal.cpp:615-654, 780-850) or `hinge layout`
    (both_active = True, n_sorts = 1: hinging.cpp:478-602) classifies for the reads [lo, hi) of a block: for every (A, B) pair
    of an active A read the longest overlap, and the second longest with use_two_matches (longest = std::sort by compare_overlap,
    i.e. descending aepos - abpos + bepos - bbpos, equal lengths where libstdc++'s introsort leaves them).  Returns (sel, a_of):
    indices into the pile-up arrays; reads ascending, the pairs of a read in the ITERATION ORDER of the reference's
    std::unordered_map - the order `.contained.txt`'s container column and the tie order of the weight sort in layout depend on
    (hinge_pick_pairs of the library runs the very container).  self_before: formats.Pileups.self_before."""
    from . import capi
    act = np.ascontiguousarray(active, dtype=np.uint8)
    return capi.pick_pairs(row_ptr, a_span, b_span, b_flag, lo, hi, accept_a=act, accept_b=act if both_active else None,
                           self_before=self_before, two_matches=use_two_matches, n_sorts=n_sorts)


class ShardedMaximal:
    """One rank's share of a sharded `hinge maximal`: classification of the block's best overlaps on this rank's GPU,
    ONE exchange (exchange 4: the (a, b) rows of overlaps in which B covers A, 8 bytes each, all-gathered in rank =
    read-id order), then every rank resolves containment over the gathered rows - a sequential pass in read-id order
    (a container of lower id counts with its final state, one of higher id with its initial state), the same for
    one merged .las and for the reference's --mlas loop because a read's pile-up lies in its own block.
    step() returns the maximal-read mask (uint8 [n_reads]); `containing` then holds, for every removed read, the container
    `.contained.txt` names (the LAST covering B in the reference's hash-map order: the backends pick the pairs in that order)."""

    def __init__(self, backend, exchange: Exchange):
        self.b, self.x = backend, exchange
        self.containing: Optional[np.ndarray] = None

    def step(self) -> np.ndarray:
        from . import capi
        rows, count = self.b.candidates()           # int32 [m, 2] on the exchange device, ascending a
        allrows = self.x.gather_lists(rows, count)   # exchange 4
        active = np.ascontiguousarray(self.b.initial_active(), dtype=np.uint8).copy()
        self.containing = capi.resolve_containment(active, allrows.cpu().numpy())
        return active


class HipMaximalBackend:
    """Per-block compute of `hinge maximal` through libhinge_hip: trim + classify (k_trim_classify) of the block's
    selected overlaps.  eff = the global mask table (.mas / exchange 2), [n_reads, 2]."""

    def __init__(self, ctx, rlen: np.ndarray, eff: np.ndarray, r_begin: int, r_end: int, row_ptr: np.ndarray, a_span: np.ndarray,
                 b_span: np.ndarray, b_flag: np.ndarray, trace: np.ndarray, trace_off: np.ndarray, tlen: np.ndarray, tbytes: int,
                 length_threshold: int, aln_threshold: int, theta: int, theta2: int, use_two_matches: bool, device: torch.device,
                 self_before: Optional[np.ndarray] = None):
        self.ctx = ctx
        self.lo, self.hi = r_begin, r_end + 1
        self.arr = (row_ptr, a_span, b_span, b_flag)
        self.thr = (int(aln_threshold), int(theta), int(theta2))
        self.use_two = bool(use_two_matches)
        self.device = device
        self.self_before = self_before
        eff = np.ascontiguousarray(eff, dtype=np.int32).reshape(-1, 2)
        self.active0 = ((eff[:, 1] - eff[:, 0]) >= length_threshold).astype(np.uint8)   # maximal.cpp:560-563
        ctx.set_reads(rlen, None)
        ctx.set_pileups(r_begin, r_end, row_ptr, a_span, b_span, b_flag)
        ctx.set_trim(True)
        ctx.set_traces(trace, trace_off, tlen, tbytes)
        ctx.set_eff_reads(eff)

    def initial_active(self) -> np.ndarray:
        return self.active0

    def candidates(self):
        from . import capi
        row_ptr, a_span, b_span, b_flag = self.arr
        sel, a_of = pick_best_pairs(row_ptr, a_span, b_span, b_flag, self.lo, self.hi, self.active0, self.use_two, self_before=self.self_before)
        types = self.ctx.trim_classify_types(sel, a_of, *self.thr)
        hit = types == capi.MT_BCOVERA
        rows = np.stack([a_of[hit], (b_flag[sel[hit]] & np.uint32(0x7FFFFFFF)).astype(np.int32)], axis=1).astype(np.int32)
        t = torch.from_numpy(np.ascontiguousarray(rows.reshape(-1, 2))).to(self.device)
        return t, int(t.shape[0])


# ---- hinge layout ----------------------------------------------------------------------------------------------
class ShardedLayout:
    """One rank's share of a sharded `hinge layout` ("layout's selection shards by A again; edges are gathered to
    rank 0").  Reads shard by DB block as in the other stages; hinge_amd/layout.py holds the steps, this class the exchanges:

      exchange 5  the classified matches of every block (72 bytes each, a few thousand per block) -> every rank: the hinge
                  bookkeeping of hinging.cpp:1262-1675 is global and sequential, and small - every rank runs it on the same rows
      exchange 6  GetMatchingPosition of the block's hinges through its matches (32 bytes per query) -> every rank
      exchange 7  the edges each rank selected for its own reads (k_select_edges) -> rank 0, which prints them in read order

    step() returns the output files as {suffix: lines} (on every rank; only rank 0 needs them).  `backend`: per-block compute
    (HipLayoutBackend: k_trim_classify, k_matching_position, k_select_edges through the C ABI)."""

    def __init__(self, backend, exchange: Exchange, params, eff: np.ndarray, maximal: np.ndarray, repeats, hinges):
        self.b, self.x, self.P = backend, exchange, params
        self.eff = np.asarray(eff, dtype=np.int64).reshape(-1, 2)
        self.maximal, self.repeats, self.hinges = maximal, repeats, hinges

    def _gather(self, rows: np.ndarray) -> np.ndarray:
        t = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int32)).to(self.x.device)
        return self.x.gather_lists(t, int(t.shape[0])).cpu().numpy()

    def step(self):
        from . import layout as L
        x, P = self.x, self.P
        n = x.blocks.n_reads
        lo, hi = x.my_range
        active, garbage = L.initial_activity(self.eff, self.maximal, P)
        mine, contained = self.b.matches(lo, hi, active, P)
        rows = self._gather(mine)                                                           # exchange 5
        gone = self._gather(np.array(contained, np.int32).reshape(-1, 1)).reshape(-1)       # "[contained] Should not happen" (:590-600)
        active[gone] = 0
        rows, off_fwd, off_bwd = L.weight_order(rows, n)
        q, q_ovl, q_pos = L.hinge_queries(rows, off_fwd, off_bwd, lo, hi, active, self.hinges)
        if len(q):
            q[:, L.Q_POSB] = self.b.matching_positions(q_ovl, q_pos)
        queries = self._gather(q)                                                           # exchange 6
        book = L.bookkeeping(n, active, rows, off_fwd, off_bwd, queries, self.repeats, self.hinges, P)
        # selection of this block's own reads: the other reads' lists are left out (their walks belong to their ranks)
        own = np.zeros(n, bool)
        own[lo:hi] = True
        cnt_f = np.where(own, np.diff(off_fwd), 0)
        cnt_b = np.where(own, np.diff(off_bwd), 0)
        keep = np.concatenate([np.arange(off_fwd[i], off_fwd[i + 1]) for i in range(lo, hi)] + [np.arange(off_bwd[i], off_bwd[i + 1]) for i in range(lo, hi)]
                              or [np.zeros(0, np.int64)]).astype(np.int64)
        sub = rows[keep]
        sf = np.concatenate([[0], np.cumsum(cnt_f)]).astype(np.int64)
        sb = (np.concatenate([[0], np.cumsum(cnt_b)]) + int(cnt_f.sum())).astype(np.int64)
        rec, h_off, h_rec, k_off, k_rec = L.selection_tables(n, sub, self.hinges, book["h_active"], book["new_killed"])
        chosen, hpos, poison = self.b.select(active, sf, sb, rec, h_off, h_rec, k_off, k_rec, P.hinge_tolerance, P.hinge_slack)
        # exchange 7: per own read and direction (read, direction, row picked in the GLOBAL row array or -1, hinge_pos), then the poison hits
        picks = np.array([(i, d, int(keep[chosen[d][i]]) if chosen[d][i] >= 0 else -1, int(hpos[d][i])) for i in range(lo, hi) for d in (0, 1)],
                         np.int32).reshape(-1, 4)
        hits = np.array([(int(keep[j]), int(poison[j])) for j in np.nonzero(poison)[0]], np.int32).reshape(-1, 2)
        picks, hits = self._gather(picks), self._gather(hits)
        chosen_all = np.full((2, n), -1, np.int64)
        hpos_all = np.full((2, n), -1, np.int64)
        chosen_all[picks[:, 1], picks[:, 0]] = picks[:, 2]
        hpos_all[picks[:, 1], picks[:, 0]] = picks[:, 3]
        poison_all = np.zeros(len(rows), np.int64)
        poison_all[hits[:, 0]] = hits[:, 1]
        return L.print_files(n, active, self.eff, rows, off_fwd, off_bwd, chosen_all, hpos_all, poison_all, self.hinges, book, garbage)


class HipLayoutBackend:
    """Per-block compute of `hinge layout` through libhinge_hip: the block's pile-ups and trace points resident on this rank's
    GPU; ProcessAlignment (k_trim_classify), GetMatchingPosition (k_matching_position) and the selection (k_select_edges)."""

    def __init__(self, ctx, rlen: np.ndarray, eff: np.ndarray, pile, trace: np.ndarray, trace_off: np.ndarray, tlen: np.ndarray, tbytes: int):
        self.ctx, self.pile = ctx, pile
        ctx.set_reads(rlen, None)
        nz = np.nonzero(np.diff(pile.row_ptr))[0]            # the block's A range: first and last read with an overlap
        r_begin, r_end = (int(nz[0]), int(nz[-1])) if len(nz) else (0, 0)
        ctx.set_pileups(r_begin, r_end, pile.row_ptr, pile.a_span, pile.b_span, pile.b_flag)
        ctx.set_trim(True)
        ctx.set_traces(trace, trace_off, tlen, tbytes)
        ctx.set_eff_reads(np.ascontiguousarray(eff, dtype=np.int32).reshape(-1, 2))

    def matches(self, lo: int, hi: int, active: np.ndarray, P):
        from . import layout as L
        classify = lambda sel, a_of: self.ctx.trim_classify(sel, a_of, P.aln_threshold, P.theta, P.theta2)
        return L.block_matches(classify, self.pile, lo, hi, active, P)

    def matching_positions(self, q_ovl: np.ndarray, q_pos: np.ndarray) -> np.ndarray:
        return self.ctx.matching_position(q_ovl, q_pos)

    def select(self, active, off_fwd, off_bwd, rec, h_off, h_rec, k_off, k_rec, tolerance: int, slack: int):
        return self.ctx.select_edges(active, off_fwd, off_bwd, rec, h_off, h_rec, k_off, k_rec, tolerance, slack)
