"""Seeded synthetic DAZZ_DB + DALIGNER overlap generator.

The reference's datasets are fetched by its demo scripts (demo/ecoli_demo/run.sh:1) and need
DALIGNER, neither of which exists here, so BASELINE.json's configs are restated as synthetic
inputs: a linear genome with planted repeat families, reads sampled
uniformly on both strands (optionally chimeric), overlaps = true interval intersections of at
least ``min_ovl`` bases plus the repeat-induced cross-copy local alignments that create the
coverage jumps `hinge filter` annotates, written in the exact on-disk formats of
``hinge_amd.formats``.  Everything is vectorised numpy so the E. coli 160x restatement
(~3e7 overlap records) is generated in well under a minute.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

from . import formats


@dataclass
class SynthSpec:
    genome_len: int = 120_000
    coverage: float = 40.0
    len_dist: str = "uniform"          # "uniform" | "lognormal"
    len_min: int = 3000
    len_max: int = 11000
    len_mean: float = 8500.0           # lognormal mean
    len_sigma: float = 0.35            # lognormal sigma (of log)
    n_repeat_families: int = 1
    repeat_len: Tuple[int, int] = (4000, 4000)
    repeat_copies: Tuple[int, int] = (2, 2)
    inverted_copies: bool = True
    chimera_frac: float = 0.0
    min_ovl: int = 1000
    end_jitter: int = 25
    indel_max: int = 6
    tspace: int = 100
    seed: int = 1
    n_blocks: int = 1
    with_qv: bool = False
    tie_quantum: int = 0               # >0: snap alignment end points to this grid (heavy ties)
    short_reads: int = 0               # reads of 500..999 bp (below the default length_threshold)
    orphan_reads: int = 0              # interior reads left without a single overlap
    self_overlap_reads: int = 0        # reads given A == B records (filter.cpp:538-561, .self.flag)
    orphan_ends: int = 0               # this many reads at EACH end of the id range without overlaps: they fall
                                       # outside [first A, last A] and get no .mas line (filter.cpp:515-517)
    trace_jitter: int = 0              # >0: the B advance of interior trace segments varies by up to +-trace_jitter per
                                       # segment (real PacBio traces: ~85-115 at tspace 100), the sum preserved: a hash of
                                       # (record, segment) moves bases between the two segments of every interior pair


@dataclass
class SynthData:
    spec: SynthSpec
    rlen: np.ndarray                   # int32 [n_reads]
    aread: np.ndarray
    bread: np.ndarray
    comp: np.ndarray                   # uint8
    ab: np.ndarray
    ae: np.ndarray
    bb: np.ndarray                     # forward-strand B coordinates
    be: np.ndarray
    block_first: List[int] = field(default_factory=list)
    qv: Optional[List[np.ndarray]] = None

    @property
    def n_reads(self) -> int:
        return int(self.rlen.shape[0])

    @property
    def novl(self) -> int:
        return int(self.aread.shape[0])


def _expand_pairs(t0: np.ndarray, t1: np.ndarray, min_ovl: int):
    """All (i, j), i < j in t0-sorted order, whose intervals intersect by >= min_ovl."""
    order = np.argsort(t0, kind="stable")
    s0 = t0[order]
    s1 = t1[order]
    n = len(s0)
    hi = np.searchsorted(s0, s1 - min_ovl, side="right")       # j < hi  <=>  t0_j <= t1_i - min_ovl
    lo = np.arange(n) + 1
    cnt = np.maximum(hi - lo, 0).astype(np.int64)
    tot = int(cnt.sum())
    if tot == 0:
        z = np.zeros(0, np.int64)
        return z, z
    idt = np.int32 if tot < 2**31 - 1 and n < 2**31 - 1 else np.int64
    i_idx = np.repeat(np.arange(n, dtype=idt), cnt)
    # j runs lo[i] .. hi[i]-1 within each i group: global arange minus the group's start offset
    shift = (lo.astype(np.int64) - (np.cumsum(cnt) - cnt))
    j_idx = (np.arange(tot, dtype=np.int64) + np.repeat(shift, cnt)).astype(idt)
    s0c = s0.astype(np.int32) if s1.max() < 2**31 - 1 else s0
    s1c = s1.astype(np.int32) if s1.max() < 2**31 - 1 else s1
    ok = (np.minimum(s1c[i_idx], s1c[j_idx]) - s0c[j_idx]) >= min_ovl
    return order[i_idx[ok]], order[j_idx[ok]]


def d_bits_ok(n_reads: int, max_len: int) -> bool:
    return 2 * int(n_reads).bit_length() + 1 + int(max_len).bit_length() + 1 <= 62


def _entry_map(roff, gs, ge, s, g_of_t_sign, g_of_t_off):
    """An "entry" maps a frame interval [t0, t1) onto a read linearly: rpos(t) = c + sg * t, with
    g(t) = g_of_t_off + g_of_t_sign * t and rpos(g) = roff + (g - gs) if s > 0 else roff + (ge - g)."""
    sg = s * g_of_t_sign
    c = np.where(s > 0, roff + g_of_t_off - gs, roff + ge - g_of_t_off)
    return c, sg


class _Records:
    """Directed overlap records accumulated by the generators (lists of arrays, concatenated by the caller)."""

    def __init__(self, spec: SynthSpec, rng):
        self.spec, self.rng = spec, rng
        self.a, self.b, self.ab, self.ae, self.bb, self.be, self.comp = [], [], [], [], [], [], []

    def emit(self, i, j, e_read, e_t0, e_t1, e_c, e_sg):
        """Emit both directed records for entry pairs (i, j)."""
        spec, rng = self.spec, self.rng
        for (x, y) in ((i, j), (j, i)):
            lo = np.maximum(e_t0[x], e_t0[y])
            hi = np.minimum(e_t1[x], e_t1[y])
            if spec.end_jitter > 0:
                lo = lo + rng.integers(0, spec.end_jitter + 1, size=len(lo))
                hi = hi - rng.integers(0, spec.end_jitter + 1, size=len(hi))
            if spec.tie_quantum > 0:
                q = spec.tie_quantum
                lo = ((lo + q - 1) // q) * q
                hi = (hi // q) * q
            ok = hi - lo >= max(spec.min_ovl - 2 * spec.end_jitter - 2 * spec.tie_quantum, 200)
            lo, hi, xx, yy = lo[ok], hi[ok], x[ok], y[ok]
            pa0 = e_c[xx] + e_sg[xx] * lo
            pa1 = e_c[xx] + e_sg[xx] * hi
            pb0 = e_c[yy] + e_sg[yy] * lo
            pb1 = e_c[yy] + e_sg[yy] * hi
            ab = np.minimum(pa0, pa1)
            ae = np.maximum(pa0, pa1)
            bb = np.minimum(pb0, pb1)
            be = np.maximum(pb0, pb1)
            if spec.indel_max > 0:
                d = rng.integers(0, spec.indel_max + 1, size=len(bb))
                side = rng.random(len(bb)) < 0.5
                bb = np.where(side, bb + d, bb)
                be = np.where(side, be, be - d)
            self.a.append(e_read[xx])
            self.b.append(e_read[yy])
            self.ab.append(ab)
            self.ae.append(ae)
            self.bb.append(bb)
            self.be.append(be)
            self.comp.append((e_sg[xx] != e_sg[yy]).astype(np.uint8))


def plant_repeats(spec: SynthSpec, rng):
    """Repeat families as non-overlapping copies on the genome: (fam_len, copies = [(family, pos, orient)])."""
    G = spec.genome_len
    fam_len: List[int] = []
    copies = []
    occupied: List[Tuple[int, int]] = []
    for f in range(spec.n_repeat_families):
        L = int(rng.integers(spec.repeat_len[0], spec.repeat_len[1] + 1))
        k = int(rng.integers(spec.repeat_copies[0], spec.repeat_copies[1] + 1))
        fam_len.append(L)
        for _ in range(k):
            for _try in range(200):
                p = int(rng.integers(2000, max(2001, G - L - 2000)))
                if all(p + L + 3000 < a or p > b + 3000 for a, b in occupied):
                    occupied.append((p, p + L))
                    o = 1 if (not spec.inverted_copies or rng.random() < 0.7) else -1
                    copies.append((f, p, o))
                    break
    return fam_len, copies


def repeat_records(rec: _Records, fam_len, copies, seg_read, seg_roff, seg_gs, seg_ge, seg_s):
    """Cross-copy local alignments of every repeat family (frame = the family's own coordinate): what creates the coverage
    jumps `hinge filter` annotates.  Appends to rec."""
    spec = rec.spec
    for f, L in enumerate(fam_len):
        e_read, e_t0, e_t1, e_c, e_sg, e_copy = [], [], [], [], [], []
        for ci, (ff, p, o) in enumerate(copies):
            if ff != f:
                continue
            x0 = np.maximum(seg_gs, p)
            x1 = np.minimum(seg_ge, p + L)
            hit = np.nonzero(x1 - x0 >= spec.min_ovl)[0]
            if len(hit) == 0:
                continue
            if o > 0:
                t0 = x0[hit] - p
                t1 = x1[hit] - p
                cc, ss = _entry_map(seg_roff[hit], seg_gs[hit], seg_ge[hit], seg_s[hit], 1, p)
            else:
                t0 = p + L - x1[hit]
                t1 = p + L - x0[hit]
                cc, ss = _entry_map(seg_roff[hit], seg_gs[hit], seg_ge[hit], seg_s[hit], -1, p + L)
            e_read.append(seg_read[hit]); e_t0.append(t0); e_t1.append(t1)
            e_c.append(cc); e_sg.append(ss); e_copy.append(np.full(len(hit), ci))
        if not e_read:
            continue
        e_read = np.concatenate(e_read); e_t0 = np.concatenate(e_t0); e_t1 = np.concatenate(e_t1)
        e_c = np.concatenate(e_c); e_sg = np.concatenate(e_sg); e_copy = np.concatenate(e_copy)
        i, j = _expand_pairs(e_t0, e_t1, spec.min_ovl)
        keep = e_copy[i] != e_copy[j]
        rec.emit(i[keep], j[keep], e_read, e_t0, e_t1, e_c, e_sg)


def generate(spec: SynthSpec) -> SynthData:
    rng = np.random.default_rng(spec.seed)
    G = spec.genome_len

    fam_len, copies = plant_repeats(spec, rng)

    # ---- reads ---------------------------------------------------------------------------
    mean_len = (spec.len_min + spec.len_max) / 2 if spec.len_dist == "uniform" else spec.len_mean
    n_reads = max(4, int(round(G * spec.coverage / mean_len)))
    if spec.len_dist == "uniform":
        lens = rng.integers(spec.len_min, spec.len_max + 1, size=n_reads)
    else:
        mu = np.log(spec.len_mean) - 0.5 * spec.len_sigma ** 2
        lens = np.clip(rng.lognormal(mu, spec.len_sigma, size=n_reads), spec.len_min, spec.len_max).astype(np.int64)
    lens = np.minimum(lens, G // 2)
    if spec.short_reads > 0:
        lens[:spec.short_reads] = rng.integers(500, 1000, size=spec.short_reads)
    starts = rng.integers(0, G - lens + 1)
    # reads are stored in DB order = order of genome start only loosely (shuffle like a real run)
    perm = rng.permutation(n_reads)
    lens, starts = lens[perm], starts[perm]
    strand = np.where(rng.random(n_reads) < 0.5, 1, -1).astype(np.int64)

    # segments: (read, roff, gs, ge, strand)
    seg_read = [np.arange(n_reads, dtype=np.int64)]
    seg_roff = [np.zeros(n_reads, np.int64)]
    seg_gs = [starts.astype(np.int64)]
    seg_ge = [(starts + lens).astype(np.int64)]
    seg_s = [strand]
    n_chim = int(round(spec.chimera_frac * n_reads))
    if n_chim > 0:
        chim = rng.choice(n_reads, size=n_chim, replace=False)
        cut = (lens[chim] * rng.uniform(0.3, 0.7, size=n_chim)).astype(np.int64)
        # first part keeps [gs, gs+cut) (strand +) or [ge-cut, ge) (strand -); second part elsewhere
        L2 = lens[chim] - cut
        gs2 = rng.integers(0, G - L2 + 1)
        s2 = np.where(rng.random(n_chim) < 0.5, 1, -1).astype(np.int64)
        g0, g1 = seg_gs[0].copy(), seg_ge[0].copy()
        plus = strand[chim] == 1
        g1[chim[plus]] = g0[chim[plus]] + cut[plus]
        g0[chim[~plus]] = g1[chim[~plus]] - cut[~plus]
        seg_gs[0], seg_ge[0] = g0, g1
        seg_read.append(chim.astype(np.int64))
        seg_roff.append(cut)
        seg_gs.append(gs2.astype(np.int64))
        seg_ge.append((gs2 + L2).astype(np.int64))
        seg_s.append(s2)
    seg_read = np.concatenate(seg_read)
    seg_roff = np.concatenate(seg_roff)
    seg_gs = np.concatenate(seg_gs)
    seg_ge = np.concatenate(seg_ge)
    seg_s = np.concatenate(seg_s)

    rec = _Records(spec, rng)

    # ---- true overlaps: frame = genome --------------------------------------------------
    c, sg = _entry_map(seg_roff, seg_gs, seg_ge, seg_s, 1, 0)
    i, j = _expand_pairs(seg_gs, seg_ge, spec.min_ovl)
    keep = seg_read[i] != seg_read[j]
    rec.emit(i[keep], j[keep], seg_read, seg_gs, seg_ge, c, sg)

    # ---- repeat-induced overlaps: frame = repeat coordinate of each family ---------------
    repeat_records(rec, fam_len, copies, seg_read, seg_roff, seg_gs, seg_ge, seg_s)
    rec_a, rec_b, rec_ab, rec_ae, rec_bb, rec_be, rec_comp = rec.a, rec.b, rec.ab, rec.ae, rec.bb, rec.be, rec.comp

    aread = np.concatenate(rec_a).astype(np.int32)
    bread = np.concatenate(rec_b).astype(np.int32)
    ab = np.concatenate(rec_ab).astype(np.int32)
    ae = np.concatenate(rec_ae).astype(np.int32)
    bb = np.concatenate(rec_bb).astype(np.int32)
    be = np.concatenate(rec_be).astype(np.int32)
    comp = np.concatenate(rec_comp).astype(np.uint8)
    rl = lens.astype(np.int32)
    ok = (ab >= 0) & (ae <= rl[aread]) & (bb >= 0) & (be <= rl[bread]) & (ae - ab >= 100) & (be - bb >= 100)
    if spec.orphan_reads > 0:
        orphan = np.zeros(n_reads, bool)
        orphan[rng.choice(np.arange(1, n_reads - 1), size=spec.orphan_reads, replace=False)] = True
        ok &= ~orphan[aread] & ~orphan[bread]
    if spec.orphan_ends > 0:
        k = spec.orphan_ends
        ok &= (aread >= k) & (aread < n_reads - k) & (bread >= k) & (bread < n_reads - k)
    aread, bread, ab, ae, bb, be, comp = (v[ok] for v in (aread, bread, ab, ae, bb, be, comp))
    if spec.self_overlap_reads > 0:
        # tandem-like self matches; every other chosen read is long and gets enough of them to cross
        # the 4.5x self-coverage flag (filter.cpp:552-561), the rest stay below it
        long_ids = np.nonzero(rl > 10000)[0]
        pick = rng.choice(long_ids if len(long_ids) >= spec.self_overlap_reads else np.arange(n_reads),
                          size=spec.self_overlap_reads, replace=False)
        sa, sab, sae, sbb, sbe = [], [], [], [], []
        for k, r in enumerate(pick):
            L = int(rl[r])
            nrec = 7 if k % 2 == 0 else 2
            for _ in range(nrec):
                span = int(L * rng.uniform(0.35, 0.45))
                a0 = int(rng.integers(0, L - 2 * span - 1))
                b0 = int(rng.integers(a0 + span // 2, L - span))
                sa.append(r); sab.append(a0); sae.append(a0 + span)
                sbb.append(b0); sbe.append(b0 + span - int(rng.integers(0, 5)))
        sa = np.asarray(sa, np.int32)
        aread = np.concatenate([aread, sa]); bread = np.concatenate([bread, sa])
        ab = np.concatenate([ab, np.asarray(sab, np.int32)]); ae = np.concatenate([ae, np.asarray(sae, np.int32)])
        bb = np.concatenate([bb, np.asarray(sbb, np.int32)]); be = np.concatenate([be, np.asarray(sbe, np.int32)])
        comp = np.concatenate([comp, np.zeros(len(sa), np.uint8)])

    # LAsort order: (aread, bread, comp, abpos)
    if d_bits_ok(len(rl), int(rl.max())):
        kb = int(len(rl)).bit_length()
        pb = int(rl.max()).bit_length() + 1
        key = (((aread.astype(np.int64) << kb) | bread.astype(np.int64)) << 1 | comp.astype(np.int64)) << pb | ab.astype(np.int64)
        order = np.argsort(key, kind="stable")
        del key
    else:
        order = np.lexsort((ab, comp, bread, aread))
    aread, bread, ab, ae, bb, be, comp = (v[order] for v in (aread, bread, ab, ae, bb, be, comp))

    # every read needs >= 1 overlap at both ends of the id range (the reference leaves reads
    # outside [first A, last A] without a .mas line: src/filter/filter.cpp:516-517,696)
    nb = max(1, spec.n_blocks)
    block_first = [int(round(k * n_reads / nb)) for k in range(nb)] + [n_reads]

    qv = None
    if spec.with_qv:
        qv = []
        for L in rl:
            nseg = (int(L) + spec.tspace - 1) // spec.tspace
            q = rng.integers(5, 38, size=nseg).astype(np.uint8)
            nbad = int(rng.integers(0, 3))
            for _ in range(nbad):
                s = int(rng.integers(0, max(1, nseg)))
                w = int(rng.integers(1, 5))
                q[s:s + w] = 45
            qv.append(q)
    return SynthData(spec=spec, rlen=rl, aread=aread, bread=bread, comp=comp, ab=ab, ae=ae,
                     bb=bb, be=be, block_first=block_first, qv=qv)


def _mix32(x):
    """lowbias32 on uint64 arrays holding 32-bit values (the same function as mix32 in tools_c/synth_io.c)."""
    m = np.uint64(0xFFFFFFFF)
    x = x & m
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & m
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & m
    x ^= x >> np.uint64(16)
    return x


def trace_jitter_delta(rec, j, jitter):
    """Bases that segment j of record `rec` (global record index) hands to segment j + 1: in [-jitter, jitter]."""
    h = _mix32((np.asarray(rec, np.uint64) * np.uint64(0x9E3779B1)) ^ (np.asarray(j, np.uint64) * np.uint64(0x85EBCA6B)))
    return (h % np.uint64(2 * jitter + 1)).astype(np.int64) - jitter


def make_traces(d: SynthData, sel: Optional[np.ndarray] = None):
    """(diffs, b-advance) byte pairs per tspace-segment of A (src/include/align.h:98-110).

    Interior segments advance A by exactly tspace, so only the first and last pair of every overlap
    need per-record arithmetic; the B-side length difference (<= indel_max) is spread one base per
    segment from the front.  With spec.trace_jitter the interior segments' B advances then vary
    (pairs (1, 2), (3, 4), ... of interior segments exchange up to trace_jitter bases, so B's total is kept)."""
    ts = d.spec.tspace
    ab = (d.ab if sel is None else d.ab[sel]).astype(np.int64)
    ae = (d.ae if sel is None else d.ae[sel]).astype(np.int64)
    bb = (d.bb if sel is None else d.bb[sel]).astype(np.int64)
    be = (d.be if sel is None else d.be[sel]).astype(np.int64)
    n = len(ab)
    nseg = (ae + ts - 1) // ts - ab // ts
    off = np.concatenate([[0], np.cumsum(nseg)]).astype(np.int64)
    tot = int(off[-1])
    base = (ab // ts) * ts
    first_len = np.where(nseg == 1, ae - ab, base + ts - ab)
    last_len = np.where(nseg == 1, ae - ab, ae - (base + (nseg - 1) * ts))
    adv = np.full(tot, ts, dtype=np.int16)
    dif = np.full(tot, ts // 8, dtype=np.uint8)
    if n:
        adv[off[:-1]] = first_len
        adv[off[1:] - 1] = last_len
        dif[off[:-1]] = first_len // 8
        dif[off[1:] - 1] = last_len // 8
        diff = (be - bb) - (ae - ab)
        mag = np.abs(diff)
        sgn = np.sign(diff).astype(np.int16)
        for j in range(int(min(mag.max(), nseg.max()))):
            m = (mag > j) & (nseg > j)
            adv[off[:-1][m] + j] += sgn[m]
        rem = np.maximum(mag - nseg, 0)
        if rem.any():
            m = rem > 0
            adv[off[1:][m] - 1] += (sgn[m] * rem[m]).astype(np.int16)
        J = int(d.spec.trace_jitter)
        if J > 0 and tot:
            recid = np.arange(d.novl, dtype=np.int64) if sel is None else np.asarray(sel, np.int64)
            j = np.arange(tot, dtype=np.int64) - np.repeat(off[:-1], nseg)
            ns = np.repeat(nseg, nseg)
            giver = np.nonzero((j % 2 == 1) & (j + 1 <= ns - 2))[0]
            delta = trace_jitter_delta(np.repeat(recid, nseg)[giver], j[giver], J).astype(np.int16)
            adv[giver] += delta
            adv[giver + 1] -= delta
        assert adv.min() >= 0 and adv.max() <= (255 if ts <= formats.TRACE_XOVR else 65535), (adv.min(), adv.max())
    if ts <= formats.TRACE_XOVR:       # one byte per trace value (LAInterface.cpp:607-614)
        tr = np.empty(2 * tot, dtype=np.uint8)
        tr[0::2] = dif
        tr[1::2] = adv.astype(np.uint8)
        return tr, (2 * off).astype(np.int64)
    tr16 = np.empty(2 * tot, dtype="<u2")   # two bytes per trace value
    tr16[0::2] = dif
    tr16[1::2] = adv.astype(np.uint16)
    return tr16.view(np.uint8), (4 * off).astype(np.int64)


def to_las_records(d: SynthData, sel: Optional[np.ndarray] = None) -> formats.LasRecords:
    idx = np.arange(d.novl) if sel is None else sel
    tr, toff = make_traces(d, idx)
    rec = np.zeros(len(idx), dtype=formats.LAS_REC_DTYPE)
    comp = d.comp[idx].astype(np.int32)
    blen = d.rlen[d.bread[idx]]
    rec["tlen"] = ((toff[1:] - toff[:-1]) // (1 if d.spec.tspace <= formats.TRACE_XOVR else 2)).astype(np.int32)
    rec["diffs"] = ((d.ae[idx] - d.ab[idx]) // 8).astype(np.int32)
    rec["abpos"] = d.ab[idx]
    rec["aepos"] = d.ae[idx]
    rec["bbpos"] = np.where(comp == 1, blen - d.be[idx], d.bb[idx])
    rec["bepos"] = np.where(comp == 1, blen - d.bb[idx], d.be[idx])
    rec["flags"] = comp.astype(np.uint32)
    rec["aread"] = d.aread[idx]
    rec["bread"] = d.bread[idx]
    return formats.LasRecords(tspace=d.spec.tspace, rec=rec, trace=tr, trace_off=toff)


_synthio = None


def _synthio_lib():
    """hinge_amd/lib/libhinge_synthio.so (hinge_amd/tools_c/synth_io.c): the same bytes as to_las_records + write_las,
    streamed from C.  None if it has not been built."""
    global _synthio
    if _synthio is None:
        import ctypes
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libhinge_synthio.so")
        if not os.path.exists(path):
            _synthio = False
        else:
            lib = ctypes.CDLL(path)
            lib.synth_write_las.restype = ctypes.c_int
            lib.synth_write_las.argtypes = [ctypes.c_char_p, ctypes.c_int64] + [ctypes.c_void_p] + [ctypes.c_int32] * 2 + [ctypes.c_void_p] * 8
            _synthio = lib
    return _synthio or None


def write_las_file(d: SynthData, path: str, sel: Optional[np.ndarray] = None, fast: Optional[bool] = None) -> None:
    """NAME.las of the data set (or of the records sel).  fast = None: the C writer when it is built and applies (one-byte
    traces), else the numpy writer; both produce the same bytes (tests/test_host_ingest.py)."""
    lib = _synthio_lib() if fast in (None, True) else None
    if lib is not None and d.spec.tspace <= formats.TRACE_XOVR:
        cols = [np.ascontiguousarray(v, dtype=t) for v, t in ((d.aread, np.int32), (d.bread, np.int32), (d.comp, np.uint8), (d.ab, np.int32),
                                                             (d.ae, np.int32), (d.bb, np.int32), (d.be, np.int32), (d.rlen, np.int32))]
        s = None if sel is None else np.ascontiguousarray(sel, dtype=np.int64)
        n = d.novl if s is None else len(s)
        rc = lib.synth_write_las(path.encode(), n, None if s is None else s.ctypes.data, d.spec.tspace, int(d.spec.trace_jitter),
                                 *[c.ctypes.data for c in cols])
        if rc == -2:
            raise AssertionError("trace generator cannot express this indel / trace-spacing combination")
        if rc != 0:
            raise OSError("cannot write %s" % path)
        return
    assert fast is not True, "libhinge_synthio.so is not built"
    formats.write_las(path, to_las_records(d, sel))


def write_dataset(d: SynthData, directory: str, name: str = "G", write_bases: bool = True) -> str:
    """Write NAME.db/.idx/.bps (+qual track), NAME.las and, for n_blocks > 1, NAME.k.las."""
    import os
    os.makedirs(directory, exist_ok=True)
    db = os.path.join(directory, name)
    formats.write_db(db, d.rlen, block_first=d.block_first, write_bases=write_bases)
    if d.qv is not None:
        formats.write_qual_track(db, d.qv)
    write_las_file(d, os.path.join(directory, name + ".las"))
    if d.spec.n_blocks > 1:
        for k in range(d.spec.n_blocks):
            lo, hi = d.block_first[k], d.block_first[k + 1]
            sel = np.nonzero((d.aread >= lo) & (d.aread < hi))[0]
            write_las_file(d, os.path.join(directory, "%s.%d.las" % (name, k + 1)), sel)
    return db


def write_paf_dataset(d: SynthData, directory: str, name: str = "G", gz: bool = False) -> None:
    """NAME.fasta + NAME.paf of the same reads and overlaps (the reference's --fasta / --paf input mode)."""
    import os
    os.makedirs(directory, exist_ok=True)
    ext = ".gz" if gz else ""
    formats.write_fasta(os.path.join(directory, name + ".fasta" + ext), d.rlen, seed=d.spec.seed, gz=gz)
    formats.write_paf(os.path.join(directory, name + ".paf" + ext), d.rlen, d.aread, d.bread, d.comp, d.ab, d.ae, d.bb, d.be, gz=gz)


def to_pileups(d: SynthData) -> formats.Pileups:
    """SoA pile-ups straight from the generator (no file round trip) for bench.py."""
    keep = d.aread != d.bread
    a = d.aread[keep]
    counts = np.bincount(a, minlength=d.n_reads).astype(np.int64)
    row_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    sl = ~keep
    return formats.Pileups(
        n_reads=d.n_reads,
        row_ptr=row_ptr,
        a_span=np.ascontiguousarray(np.stack([d.ab[keep], d.ae[keep]], axis=1).astype(np.int32)),
        b_span=np.ascontiguousarray(np.stack([d.bb[keep], d.be[keep]], axis=1).astype(np.int32)),
        b_flag=np.ascontiguousarray(d.bread[keep].astype(np.uint32) | (d.comp[keep].astype(np.uint32) << np.uint32(31))),
        las_index=np.nonzero(keep)[0].astype(np.int64),
        self_a=d.aread[sl].astype(np.int32),
        self_span=np.stack([d.ab[sl], d.ae[sl], d.bb[sl], d.be[sl]], axis=1).astype(np.int32),
    )


# BASELINE.json configs restated.  Round 5: the data sets that reach ProcessAlignment /
# GetMatchingPosition at size carry jittered traces (per-segment B advances of tspace +- 15 %, as real PacBio traces vary) -
# `tiny`, `ties`, `deep` ... keep the near-uniform ones, so both shapes stay covered.
CONFIGS = {
    "tiny": SynthSpec(genome_len=120_000, coverage=40, seed=7),
    "tiny_qv": SynthSpec(genome_len=120_000, coverage=40, seed=8, with_qv=True),
    "tiny_mlas": SynthSpec(genome_len=150_000, coverage=35, seed=9, n_blocks=3, n_repeat_families=2, trace_jitter=12),
    "ties": SynthSpec(genome_len=100_000, coverage=60, seed=10, tie_quantum=100, end_jitter=0, indel_max=0,
                      n_repeat_families=2, repeat_copies=(2, 3)),
    "chimera": SynthSpec(genome_len=150_000, coverage=40, seed=11, chimera_frac=0.05, n_repeat_families=2, trace_jitter=15),
    "long_repeat": SynthSpec(genome_len=150_000, coverage=80, len_min=3000, len_max=8000, repeat_len=(12000, 12000),
                             repeat_copies=(2, 2), inverted_copies=False, seed=23),
    # reads from 4 kb to 120 kb: every LDS-slot class of the mask/annotate kernel (1, 2, 4 slots, and > 91 kb: general kernel)
    "long_reads": SynthSpec(genome_len=400_000, coverage=40, len_dist="lognormal", len_mean=25000, len_sigma=0.7, len_min=4000,
                            len_max=120000, repeat_len=(9000, 9000), repeat_copies=(3, 3), seed=31, trace_jitter=15),
    # trace spacing 200: two bytes per trace value on disk (tspace > 125), a QV track at that spacing
    "tspace200": SynthSpec(genome_len=150_000, coverage=45, seed=41, tspace=200, with_qv=True, n_repeat_families=2,
                           repeat_len=(6000, 6000), repeat_copies=(2, 2), trace_jitter=28),
    # reads below length_threshold, reads without any overlap, A == B records on both sides of the
    # self-coverage flag (filter.cpp:538-561)
    "edges": SynthSpec(genome_len=130_000, coverage=42, seed=43, len_max=14000, min_ovl=500, short_reads=24,
                       orphan_reads=6, self_overlap_reads=6, with_qv=True),
    # 450x over a 3-copy repeat: pile-ups of 2049-4096 overlaps with undecided annotations (the full-size instance of k_hinge_call)
    "deep": SynthSpec(genome_len=50_000, coverage=450, seed=53, n_repeat_families=1, repeat_len=(5000, 5000), repeat_copies=(3, 3)),
    "orphan_ends": SynthSpec(genome_len=100_000, coverage=40, seed=47, orphan_ends=3),
    "cfg1_ecoli_demo": SynthSpec(genome_len=4_600_000, coverage=30, seed=1, n_repeat_families=5,
                                 repeat_len=(1000, 5000), repeat_copies=(2, 3), trace_jitter=15),
    "cfg2_ecoli160": SynthSpec(genome_len=4_600_000, coverage=160, len_dist="lognormal", len_mean=8500,
                               len_min=1500, len_max=40000, seed=2, n_repeat_families=1,
                               repeat_len=(5000, 5000), repeat_copies=(7, 7), trace_jitter=15),
    "cfg3_nctc": SynthSpec(genome_len=5_000_000, coverage=100, len_dist="lognormal", len_mean=8000,
                           len_min=1500, len_max=40000, seed=3, n_repeat_families=40,
                           repeat_len=(1000, 8000), repeat_copies=(2, 6), chimera_frac=0.02, trace_jitter=15),
    "cfg4_yeast": SynthSpec(genome_len=12_000_000, coverage=80, len_dist="lognormal", len_mean=8000,
                            len_min=1500, len_max=40000, seed=4, n_repeat_families=20,
                            repeat_len=(1000, 6000), repeat_copies=(2, 5), n_blocks=8, trace_jitter=15),
    # config 5 (HBM-roofline stress): generated on the device by hinge_amd.synth_device (no .las of this size is ever written).
    # SURVEY 8(d) quotes ~10^9 overlaps for the whole 100 Mb genome; with overlaps of >= 1 kb between 7 kb reads at 100x the
    # model gives ~170 per read, 2.4e8 in all, so "one rank's share of 10^9" (>= 1.25e8 overlaps) is the 52 Mb block below.
    "cfg5_stress": SynthSpec(genome_len=100_000_000, coverage=100, len_dist="lognormal", len_mean=7000, len_min=1500,
                             len_max=40000, seed=5, n_repeat_families=60, repeat_len=(1000, 8000), repeat_copies=(2, 5), n_blocks=8),
    "cfg5_share": SynthSpec(genome_len=52_000_000, coverage=100, len_dist="lognormal", len_mean=7000, len_min=1500,
                            len_max=40000, seed=5, n_repeat_families=30, repeat_len=(1000, 8000), repeat_copies=(2, 5)),
}
