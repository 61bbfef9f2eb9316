"""Seeded synthetic input of the whole chain filter -> maximal -> layout -> clip -> draft-path -> draft (-> consensus): reads WITH
bases off a planted genome and read-vs-read overlaps WITH real trace points.  Test / bench tooling; no
reference code.

`hinge_amd.synth` makes overlaps as intervals only (the three graph stages never look at a base); `hinge draft` realigns reads
base by base between trace points (draft.cpp:213-216: recoverAlignment + getAlignmentTags), so here every read is made by an
explicit edit script against the genome and the trace points of an overlap A x B are the composition of the two scripts:
for each multiple of `tspace` on A the genome position under it, and B's position there.  The `diffs` of a segment is the number
of edit operations of BOTH reads inside it (+2: a boundary can fall inside an inserted base) - an upper bound of the segment's
edit distance, which is all the realigner needs of it (LAInterface.cpp:3444-3456 sizes its waves from it).

Reads lie on either strand; a circular genome lets reads wrap around the origin.  The read DB is written with bases; the .las
holds both directed records of every pair whose genome intervals share at least `min_ovl` bases, in LAsort order.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from . import formats


@dataclass(frozen=True)
class DraftSpec:
    genome_len: int = 60_000
    circular: bool = False
    coverage: float = 25.0
    read_len: tuple = (6_000, 12_000)
    p_sub: float = 0.0
    p_ins: float = 0.0
    p_del: float = 0.0
    min_ovl: int = 1_500
    tspace: int = 100
    repeat: Optional[tuple] = None     # (length, first copy at, second copy at): the second stretch becomes a copy of the first
    seed: int = 1


@dataclass
class DraftData:
    spec: DraftSpec
    genome: np.ndarray                 # uint8 0..3
    reads: List[np.ndarray]            # as stored in the DB
    g0: np.ndarray                     # genome interval of every read (g0 may exceed genome_len - it wraps - on a circular genome)
    g1: np.ndarray
    strand: np.ndarray                 # 1: stored as the reverse complement of the genome's forward strand
    rec: np.ndarray                    # formats.LAS_REC_DTYPE
    trace: np.ndarray
    trace_off: np.ndarray

    @property
    def rlen(self) -> np.ndarray:
        return np.asarray([len(r) for r in self.reads], np.int32)


def revcomp(b: np.ndarray) -> np.ndarray:
    return (3 - b[::-1]).astype(np.uint8)


def _make_read(rng, gseq: np.ndarray, spec: DraftSpec):
    """One read over the genome stretch gseq: (forward-strand bases, P, C) with P[i] = read bases emitted for the stretch's
    positions < i (P[n] = the read's length) and C[i] = edit operations at those positions."""
    n = len(gseq)
    ins = rng.random(n) < spec.p_ins
    r = rng.random(n)
    dele = r < spec.p_del
    sub = ~dele & (r < spec.p_del + spec.p_sub)
    main = gseq.copy()
    main[sub] = (main[sub] + rng.integers(1, 4, size=int(sub.sum()))) % 4
    E = np.zeros((n, 2), np.uint8)
    M = np.zeros((n, 2), bool)
    E[:, 0] = rng.integers(0, 4, size=n); M[:, 0] = ins
    E[:, 1] = main; M[:, 1] = ~dele
    out = E[M]
    P = np.concatenate([[0], np.cumsum(M.sum(axis=1))]).astype(np.int64)
    C = np.concatenate([[0], np.cumsum(ins.astype(np.int64) + dele + sub)]).astype(np.int64)
    return out.astype(np.uint8), P, C


def generate(spec: DraftSpec) -> DraftData:
    rng = np.random.default_rng(spec.seed)
    G = spec.genome_len
    genome = rng.integers(0, 4, size=G, dtype=np.uint8)
    if spec.repeat:
        L, p1, p2 = spec.repeat
        genome[p2:p2 + L] = genome[p1:p1 + L]
    n_reads = int(G * spec.coverage / (0.5 * (spec.read_len[0] + spec.read_len[1])))
    lens = rng.integers(spec.read_len[0], spec.read_len[1] + 1, size=n_reads)
    if spec.circular:
        starts = rng.integers(0, G, size=n_reads)
    else:
        starts = rng.integers(-lens // 3, G - 2 * lens // 3, size=n_reads)
    g0 = np.maximum(starts, 0) if not spec.circular else starts
    g1 = np.minimum(starts + lens, G) if not spec.circular else starts + lens
    keep = g1 - g0 >= max(spec.min_ovl, 2000)
    g0, g1 = g0[keep], g1[keep]
    if not spec.circular:      # both genome ends covered from their first / last base
        g0[np.argmin(g0)] = 0
        g1[np.argmax(g1)] = G
    order = rng.permutation(len(g0))     # read ids carry no position information
    g0, g1 = g0[order], g1[order]
    n = len(g0)
    strand = rng.integers(0, 2, size=n).astype(np.uint8)
    reads, Ps, Cs = [], [], []
    for i in range(n):
        idx = np.arange(g0[i], g1[i]) % G
        fwd, P, C = _make_read(rng, genome[idx], spec)
        reads.append(revcomp(fwd) if strand[i] else fwd)
        Ps.append(P); Cs.append(C)
    rlen = np.asarray([len(r) for r in reads], np.int64)
    # pairs whose genome intervals intersect by >= min_ovl (on a circular genome an interval may be seen shifted by +-G)
    recs, traces = [], []
    ts = spec.tspace
    tdt = np.uint8 if ts <= 125 else np.dtype("<u2")
    shifts = (0, G, -G) if spec.circular else (0,)
    for x in range(n):
        for sh in shifts:
            lo = np.maximum(g0[x], g0 + sh)
            hi = np.minimum(g1[x], g1 + sh)
            for y in np.nonzero(hi - lo >= spec.min_ovl)[0]:
                if y == x:
                    continue
                recs_xy = _record(x, int(y), int(lo[y]), int(hi[y]), sh, g0, g1, strand, rlen, Ps, Cs, ts)
                if recs_xy is not None:
                    recs.append(recs_xy[0]); traces.append(np.asarray(recs_xy[1], dtype=tdt).reshape(-1))
    order = sorted(range(len(recs)), key=lambda i: (recs[i][7], recs[i][8], recs[i][6], recs[i][2]))     # LAsort: aread, bread, comp, abpos
    rec = np.zeros(len(recs), dtype=formats.LAS_REC_DTYPE)
    for o, i in enumerate(order):
        r = recs[i]
        rec[o] = (r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8])
    tr = [traces[i].view(np.uint8) for i in order]
    toff = np.concatenate([[0], np.cumsum([len(t) for t in tr])]).astype(np.int64)
    return DraftData(spec, genome, reads, g0, g1, strand, rec, np.concatenate(tr) if tr else np.zeros(0, np.uint8), toff)


def _record(x, y, lo, hi, sh, g0, g1, strand, rlen, Ps, Cs, ts):
    """The directed record A = x, B = y over the genome interval [lo, hi) (in x's coordinates; y's are shifted by sh).
    Works in the frame A is stored in: the genome's forward strand if strand[x] == 0, its reverse strand otherwise."""
    PA, PB, CA, CB = Ps[x], Ps[y], Cs[x], Cs[y]
    la, lb = int(rlen[x]), int(rlen[y])
    ia0, ia1 = lo - g0[x], hi - g0[x]                 # indices into x's tables
    ib0, ib1 = lo - (g0[y] + sh), hi - (g0[y] + sh)
    if strand[x] == 0:
        comp = int(strand[y])
        abpos, aepos = int(PA[ia0]), int(PA[ia1])
        if aepos - abpos < 50:
            return None
        # boundaries: multiples of ts strictly inside (abpos, aepos)
        bounds = np.arange((abpos // ts + 1) * ts, aepos, ts)
        gi = np.searchsorted(PA, bounds, side="left")           # smallest table index with PA >= boundary
        gi = np.clip(gi, ia0, ia1)
        ga = np.concatenate([[ia0], gi, [ia1]])
        gb = ga - ia0 + ib0
        bpos = PB[gb]
        ops = (CA[ga[1:]] - CA[ga[:-1]]) + (CB[gb[1:]] - CB[gb[:-1]]) + 2
        bbpos, bepos = int(bpos[0]), int(bpos[-1])
    else:
        comp = 1 - int(strand[y])
        abpos, aepos = la - int(PA[ia1]), la - int(PA[ia0])
        if aepos - abpos < 50:
            return None
        bounds = np.arange((abpos // ts + 1) * ts, aepos, ts)
        # the largest table index whose position in the reverse frame is still >= the boundary: PA[g] <= la - boundary
        gi = np.searchsorted(PA, la - bounds, side="right") - 1
        gi = np.clip(gi, ia0, ia1)
        ga = np.concatenate([[ia1], gi, [ia0]])                 # descending along the genome
        gb = ga - ia0 + ib0
        bpos = lb - PB[gb]
        ops = (CA[ga[:-1]] - CA[ga[1:]]) + (CB[gb[:-1]] - CB[gb[1:]]) + 2
        bbpos, bepos = int(bpos[0]), int(bpos[-1])
    adv = np.diff(bpos)
    if bepos - bbpos < 50 or (adv < 0).any():
        return None
    tmax = 255 if ts <= 125 else 65535
    if adv.max() > tmax or ops.max() > tmax:
        return None
    pairs = np.stack([ops, adv], axis=1)
    return (2 * len(pairs), int(ops.sum()), abpos, bbpos, aepos, bepos, comp, x, y), pairs


def write_dataset(d: DraftData, directory: str, name: str = "G") -> str:
    """NAME.db (+ .idx, .bps with the bases) and NAME.las."""
    os.makedirs(directory, exist_ok=True)
    formats.write_db(os.path.join(directory, name), d.rlen, bases=d.reads)
    formats.write_las(os.path.join(directory, name + ".las"), formats.LasRecords(d.spec.tspace, d.rec, d.trace, d.trace_off))
    return os.path.join(directory, name)


CONFIGS = {
    # noise-free reads: every stage's coordinates can be checked exactly - the draft must BE the genome
    "draft_clean": DraftSpec(genome_len=60_000, coverage=22.0, seed=31),
    "draft_clean_circular": DraftSpec(genome_len=50_000, circular=True, coverage=25.0, seed=32),
    "draft_noisy": DraftSpec(genome_len=60_000, coverage=28.0, p_sub=0.02, p_ins=0.05, p_del=0.03, seed=33),
    "draft_noisy_circular": DraftSpec(genome_len=45_000, circular=True, coverage=30.0, p_sub=0.015, p_ins=0.04, p_del=0.025, seed=34),
    # a 3 kb two-copy repeat that most reads do not span: hinges, a branching graph, several contigs
    "draft_repeat": DraftSpec(genome_len=90_000, coverage=30.0, p_sub=0.01, p_ins=0.03, p_del=0.02, repeat=(3_000, 20_000, 60_000), seed=35),
    "draft_twobyte": DraftSpec(genome_len=40_000, coverage=22.0, p_sub=0.01, p_ins=0.03, p_del=0.02, tspace=200, seed=36),
}
