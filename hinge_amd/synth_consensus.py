"""Seeded synthetic input of `hinge consensus`: a draft DB (contigs, with bases), a read DB (with bases)
and the draft-vs-reads .las with real trace points - what `HPC.daligner draft reads` + LAmerge hand to
src/consensus/consensus.cpp (demo/ecoli_demo/run.sh:38-42).  Test / bench tooling; no reference code.

Model.  A draft contig is random bases.  The "truth" differs from the draft at sparse, planted places (the draft's own errors:
a base the draft lacks, a base it has too much, a wrong base); a read is a stretch of the truth with independent noise
(substitutions, insertions, deletions), on either strand, optionally with unaligned flanks.  Because every read is made by an
explicit edit script against the draft, the A <-> B correspondence is known exactly: the trace points are (edit operations,
B bases) per `tspace` bases of the draft, as DALIGNER writes them (src/include/align.h:98-110), and the recorded operation
counts bound the realignment's wave count (LAInterface.cpp:3444-3456 sizes its arrays from them).
Planted low-coverage windows (fewer than three reads) exercise the lower-case branch of consensus.cpp:232-238.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

from . import formats


@dataclass(frozen=True)
class ConsensusSpec:
    n_contigs: int = 3
    contig_len: Tuple[int, int] = (12_000, 30_000)
    coverage: float = 25.0
    read_len: Tuple[int, int] = (1_500, 6_000)
    p_sub: float = 0.03
    p_ins: float = 0.05
    p_del: float = 0.03
    draft_errors_per_kb: float = 2.0       # planted differences between draft and truth
    p_carry: float = 0.92                  # a read shows a planted difference with this probability
    low_cov_windows: int = 2               # per contig: windows that keep at most two reads
    flank_max: int = 40                    # unaligned read bases on either side of the alignment
    short_alignments: int = 2              # per contig: alignments below min_length / below the 210-column chop limit
    duplicate_b: int = 1                   # per contig: a second alignment of a read that already has one (remove_multialign)
    empty_contigs: int = 0                 # contigs without any alignment (printed as they are, consensus.cpp:158-162)
    tspace: int = 100
    seed: int = 1


@dataclass
class ConsensusData:
    spec: ConsensusSpec
    contigs: List[np.ndarray]          # uint8 0..3
    reads: List[np.ndarray]            # uint8 0..3, as stored in the read DB
    rec: np.ndarray                    # formats.LAS_REC_DTYPE, sorted by aread
    trace: np.ndarray                  # the trace bytes of the file: (diffs, b advance) pairs, uint8 or little-endian uint16
    trace_off: np.ndarray              # int64 [n + 1], in bytes

    @property
    def n_alignments(self) -> int:
        return int(self.rec.shape[0])


def revcomp(b: np.ndarray) -> np.ndarray:
    return (3 - b[::-1]).astype(np.uint8)


def _make_read(rng, draft: np.ndarray, planted, ab: int, ae: int, spec: ConsensusSpec):
    """One read over draft[ab:ae): returns (aligned B bases, per-segment (diffs, b advance) list).  planted = (kind[L], base[L]):
    kind 0 none, 1 the truth has `base` in front of this draft base, 2 the draft base is one too many, 3 it should be `base`.
    Vectorised over the stretch: per draft position up to three bases come out (planted insertion, noise insertion, the
    position's own base unless deleted), in that order."""
    ts = spec.tspace
    n = ae - ab
    P = np.arange(ab, ae)
    pk, pb = planted[0][ab:ae], planted[1][ab:ae]
    carried = (pk > 0) & (rng.random(n) < spec.p_carry)
    c_ins, c_del, c_sub = carried & (pk == 1), carried & (pk == 2), carried & (pk == 3)
    n_ins = rng.random(n) < spec.p_ins
    other = ~(c_del | c_sub)
    r2 = rng.random(n)
    n_del = other & (r2 < spec.p_del)
    n_sub = other & ~n_del & (r2 < spec.p_del + spec.p_sub)
    E = np.zeros((n, 3), dtype=np.uint8)
    M = np.zeros((n, 3), dtype=bool)
    E[:, 0] = pb; M[:, 0] = c_ins
    E[:, 1] = rng.integers(0, 4, size=n); M[:, 1] = n_ins
    main = draft[ab:ae].copy()
    main[n_sub] = (main[n_sub] + rng.integers(1, 4, size=int(n_sub.sum()))) % 4
    main[c_sub] = pb[c_sub]
    E[:, 2] = main; M[:, 2] = ~(c_del | n_del)
    out = E[M]
    ops = c_ins.astype(np.int64) + n_ins + c_del + c_sub + n_del + n_sub
    seg = P // ts - ab // ts
    nseg = int(seg[-1]) + 1
    d = np.bincount(seg, weights=ops, minlength=nseg).astype(np.int64)
    b = np.bincount(seg, weights=M.sum(axis=1), minlength=nseg).astype(np.int64)
    return out, list(zip(d.tolist(), b.tolist()))


def generate(spec: ConsensusSpec) -> ConsensusData:
    rng = np.random.default_rng(spec.seed)
    contigs, reads = [], []
    recs, traces = [], []
    n_total = spec.n_contigs + spec.empty_contigs
    tdt = np.uint8 if spec.tspace <= 125 else np.dtype("<u2")      # trace values: one byte up to tspace 125, two beyond (align.h:64-69)
    tmax = 255 if spec.tspace <= 125 else 65535
    for c in range(n_total):
        L = int(rng.integers(spec.contig_len[0], spec.contig_len[1] + 1))
        draft = rng.integers(0, 4, size=L, dtype=np.uint8)
        contigs.append(draft)
        if c >= spec.n_contigs:
            continue
        planted = (np.zeros(L, np.uint8), np.zeros(L, np.uint8))
        for pos in rng.choice(np.arange(200, L - 200), size=max(1, int(L / 1000 * spec.draft_errors_per_kb)), replace=False):
            k = int(rng.integers(1, 4))
            planted[0][pos] = k
            planted[1][pos] = int(rng.integers(0, 4)) if k == 1 else int((draft[pos] + rng.integers(1, 4)) % 4)
        lows = []
        for _ in range(spec.low_cov_windows):
            s = int(rng.integers(500, max(501, L - 1500)))
            lows.append((s, s + int(rng.integers(150, 600))))
        n_reads = int(L * spec.coverage / (0.5 * (spec.read_len[0] + spec.read_len[1])))
        spans = []
        for _ in range(n_reads):
            ln = int(rng.integers(spec.read_len[0], spec.read_len[1] + 1))
            ab = int(rng.integers(-ln // 2, L - ln // 2))
            ab, ae = max(ab, 0), min(ab + ln, L)
            if ae - ab < 400:
                continue
            spans.append((ab, ae))
        # thin the low-coverage windows out: at most two reads may touch each
        kept = []
        touch = [0] * len(lows)
        for ab, ae in spans:
            hit = [w for w, (s, e) in enumerate(lows) if ab < e and ae > s]
            if any(touch[w] >= 2 for w in hit):
                continue
            for w in hit:
                touch[w] += 1
            kept.append((ab, ae))
        for _ in range(spec.short_alignments):     # short ones: below the 210-column chop limit / below min_length
            ab = int(rng.integers(0, L - 300))
            kept.append((ab, ab + int(rng.integers(120, 260))))
        first_read_of_contig = len(reads)
        for ab, ae in kept:
            B, pairs = _make_read(rng, draft, planted, ab, ae, spec)
            if len(B) == 0:
                continue
            fl = int(rng.integers(0, spec.flank_max + 1)) if spec.flank_max else 0
            fr = int(rng.integers(0, spec.flank_max + 1)) if spec.flank_max else 0
            whole = np.concatenate([rng.integers(0, 4, size=fl, dtype=np.uint8), B, rng.integers(0, 4, size=fr, dtype=np.uint8)])
            comp = int(rng.integers(0, 2))
            reads.append(revcomp(whole) if comp else whole)
            assert all(0 <= d <= tmax and 0 <= b <= tmax for d, b in pairs)
            recs.append((2 * len(pairs), sum(d for d, _ in pairs), ab, fl, ae, fl + len(B), comp, c, len(reads) - 1))
            traces.append(np.asarray(pairs, dtype=tdt).reshape(-1))
        for _ in range(spec.duplicate_b):          # a read with two alignments to the same contig
            if len(reads) == first_read_of_contig:
                break
            k = int(rng.integers(first_read_of_contig, len(reads)))
            j = next(i for i in range(len(recs) - 1, -1, -1) if recs[i][8] == k)
            old = recs[j]
            # the same read once more over the first part of its stretch: a fresh (valid) script is not needed for the part that
            # is re-used - take the leading segments of the existing trace up to a segment boundary
            prs = traces[j].reshape(-1, 2)
            nseg = max(1, len(prs) // 2)
            a_end = min(((old[2] // spec.tspace) + nseg) * spec.tspace, old[4])
            if a_end >= old[4]:
                continue
            sub = prs[:nseg]
            recs.append((2 * nseg, int(sub[:, 0].sum()), old[2], old[3], a_end, old[3] + int(sub[:, 1].sum()), old[6], c, k))
            traces.append(sub.reshape(-1).copy())
    order = sorted(range(len(recs)), key=lambda i: (recs[i][7], i))
    rec = np.zeros(len(recs), dtype=formats.LAS_REC_DTYPE)
    for o, i in enumerate(order):
        r = recs[i]
        rec[o] = (r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8])
    tr = [traces[i].view(np.uint8) for i in order]                 # as the bytes of the file
    toff = np.concatenate([[0], np.cumsum([len(t) for t in tr])]).astype(np.int64)
    return ConsensusData(spec, contigs, reads, rec, np.concatenate(tr) if tr else np.zeros(0, np.uint8), toff)


def write_dataset(d: ConsensusData, directory: str, draft: str = "draft", reads: str = "reads", min_length: int = 500) -> str:
    """draft.db + reads.db (with .bps), draft.reads.las, nominal.ini with [consensus] min_length."""
    os.makedirs(directory, exist_ok=True)
    formats.write_db(os.path.join(directory, draft), np.asarray([len(c) for c in d.contigs], np.int32), bases=d.contigs)
    formats.write_db(os.path.join(directory, reads), np.asarray([len(r) for r in d.reads], np.int32), bases=d.reads)
    formats.write_las(os.path.join(directory, "%s.%s.las" % (draft, reads)), formats.LasRecords(d.spec.tspace, d.rec, d.trace, d.trace_off))
    with open(os.path.join(directory, "nominal.ini"), "w") as f:
        f.write("[consensus]\nmin_length = %d;\n" % min_length)
    return directory


CONFIGS = {
    "cns_tiny": ConsensusSpec(n_contigs=2, contig_len=(3_000, 5_000), coverage=12.0, read_len=(600, 1_800), seed=11),
    "cns_small": ConsensusSpec(n_contigs=3, contig_len=(12_000, 30_000), coverage=25.0, seed=12, empty_contigs=1),
    "cns_noisy": ConsensusSpec(n_contigs=2, contig_len=(8_000, 12_000), coverage=30.0, p_sub=0.05, p_ins=0.09, p_del=0.05, seed=13),
    "cns_clean": ConsensusSpec(n_contigs=2, contig_len=(6_000, 9_000), coverage=8.0, p_sub=0.0, p_ins=0.0, p_del=0.0, draft_errors_per_kb=3.0, p_carry=1.0, seed=14),
    "cns_twobyte": ConsensusSpec(n_contigs=2, contig_len=(7_000, 10_000), coverage=14.0, read_len=(900, 3_000), tspace=200, seed=16),
    "cns_midsize": ConsensusSpec(n_contigs=3, contig_len=(60_000, 90_000), coverage=22.0, read_len=(2_000, 9_000), p_sub=0.04, p_ins=0.07, p_del=0.04, seed=17, low_cov_windows=3),
    "cns_bench": ConsensusSpec(n_contigs=4, contig_len=(900_000, 1_300_000), coverage=30.0, read_len=(3_000, 11_000), seed=15, low_cov_windows=3),
}
