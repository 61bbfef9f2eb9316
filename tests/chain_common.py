"""The last link of the chain test: reads mapped onto the contigs `hinge draft` wrote, as a draft-vs-reads .las for `hinge consensus`.

The reference's pipeline gets these alignments from DALIGNER (out of scope).  Here they come from what the generator
knows: every read of hinge_amd.synth_draft is an explicit edit script against the planted genome, so once a contig has one too - a
global alignment of the contig with the genome stretch it covers, made below by anchoring on exact k-mers and Needleman-Wunsch
between the anchors - the trace points of contig x read are the composition of the two scripts, exactly as the generator composes
read x read (synth_draft._record).  The same alignment gives a contig's identity with the genome: the number the chain test asserts."""
import numpy as np

K = 18


def to_bases(s: str) -> np.ndarray:
    lut = np.full(256, 0, np.uint8)
    for ch, v in (("a", 0), ("c", 1), ("g", 2), ("t", 3), ("A", 0), ("C", 1), ("G", 2), ("T", 3)):
        lut[ord(ch)] = v
    return lut[np.frombuffer(s.encode(), np.uint8)]


def _kmer_codes(b: np.ndarray, k: int = K) -> np.ndarray:
    n = len(b) - k + 1
    if n <= 0:
        return np.zeros(0, np.int64)
    code = np.zeros(n, np.int64)
    for j in range(k):
        code = code * 4 + b[j:j + n].astype(np.int64)
    return code


def nw(a: np.ndarray, b: np.ndarray):
    """Global alignment of a with b (match 0, mismatch / gap 1: an edit distance).  Returns the operations from start to end:
    0 = a[i] with b[j] (equal), 1 = substitution, 2 = a[i] alone (insertion in a), 3 = b[j] alone (deletion from a)."""
    n, m = len(a), len(b)
    H = np.zeros((n + 1, m + 1), np.int32)
    H[0] = np.arange(m + 1)
    jj = np.arange(m + 1, dtype=np.int32)
    for i in range(1, n + 1):
        prev = H[i - 1]
        t = np.empty(m + 1, np.int32)
        t[0] = i
        t[1:] = np.minimum(prev[:-1] + (b != a[i - 1]), prev[1:] + 1)
        H[i] = np.minimum.accumulate(t - jj) + jj          # H[i][j] = min over k <= j of t[k] + (j - k)
    ops = []
    i, j = n, m
    while i > 0 or j > 0:
        if i > 0 and j > 0 and H[i, j] == H[i - 1, j - 1] + (a[i - 1] != b[j - 1]):
            ops.append(0 if a[i - 1] == b[j - 1] else 1); i -= 1; j -= 1
        elif i > 0 and H[i, j] == H[i - 1, j] + 1:
            ops.append(2); i -= 1
        else:
            ops.append(3); j -= 1
    return ops[::-1], int(H[n, m])


class ContigOnGenome:
    """A contig placed on the genome: strand, the contig stretch [q0, q1) (in the frame oriented like the genome's forward strand) that
    lies between its first and last anchor, the genome stretch [g0, g1) it covers, the tables P / C over that genome stretch as
    synth_draft keeps them for a read (P[i] = contig position under genome position g0 + i, C[i] = edit operations before it),
    and the identity of the stretch: equal columns / alignment columns."""

    def __init__(self, seq: np.ndarray, genome: np.ndarray):
        self.length = len(seq)
        gk = _kmer_codes(genome)
        order = np.argsort(gk, kind="stable")
        sk = gk[order]
        uniq = np.ones(len(sk), bool)
        uniq[1:] &= sk[1:] != sk[:-1]
        uniq[:-1] &= sk[:-1] != sk[1:]
        ukeys, upos = sk[uniq], order[uniq]
        best = None
        for strand, q in ((0, seq), (1, (3 - seq[::-1]).astype(np.uint8))):
            qk = _kmer_codes(q)
            at = np.searchsorted(ukeys, qk)
            at = np.minimum(at, len(ukeys) - 1)
            hit = ukeys[at] == qk
            qp, gp = np.nonzero(hit)[0], upos[at[hit]]
            if best is None or len(qp) > len(best[1]):
                best = (strand, qp, gp, q)
        self.strand, qp, gp, q = best
        assert len(qp) >= 20, "the contig does not anchor on the genome"
        # keep the anchors of the main diagonal band, ascending in both coordinates and at least K apart (greedy)
        diag = gp - qp
        ok = np.abs(diag - np.median(diag)) < max(2000, len(q) // 10)
        qp, gp = qp[ok], gp[ok]
        keep_q, keep_g = [int(qp[0])], [int(gp[0])]
        for x, y in zip(qp[1:], gp[1:]):
            if x >= keep_q[-1] + K and y >= keep_g[-1] + K and abs((y - keep_g[-1]) - (x - keep_q[-1])) < 400:
                keep_q.append(int(x)); keep_g.append(int(y))
        self.q0, self.g0 = keep_q[0], keep_g[0]
        self.q1, self.g1 = keep_q[-1] + K, keep_g[-1] + K
        ng = self.g1 - self.g0
        emit = np.zeros(ng, np.int64)
        ops_at = np.zeros(ng, np.int64)
        cols = eq = 0
        for a in range(len(keep_q)):
            # the anchor itself: K equal columns
            g = keep_g[a] - self.g0
            emit[g:g + K] += 1
            cols += K; eq += K
            if a + 1 == len(keep_q):
                break
            qa, ga, qb, gb = keep_q[a] + K, keep_g[a] + K, keep_q[a + 1], keep_g[a + 1]
            if qa == qb and ga == gb:
                continue
            ops, _ = nw(q[qa:qb], genome[ga:gb])
            g = ga - self.g0
            for op in ops:
                cols += 1
                if op <= 1:
                    emit[g] += 1; ops_at[g] += op; eq += op == 0; g += 1
                elif op == 2:                      # a contig base with no genome base: in front of the next genome position
                    emit[g] += 1; ops_at[g] += 1   # (g < ng: an anchor follows)
                else:
                    ops_at[g] += 1; g += 1
        self.P = self.q0 + np.concatenate([[0], np.cumsum(emit)]).astype(np.int64)
        self.C = np.concatenate([[0], np.cumsum(ops_at)]).astype(np.int64)
        assert self.P[-1] == self.q1, (self.P[-1], self.q1)
        self.identity = eq / max(cols, 1)
        self.columns = cols

    def inner_identity(self, margin):
        """1 - edit operations per genome position over the stretch's interior: `margin` genome positions left out at either end
        (a draft's first and last few kb are its weakest: thin coverage, and the reference cuts the prefix / suffix of strand-1
        ends from the wrong strand - tests/draft_common.py inner_mismatches)."""
        lo, hi = margin, len(self.C) - 1 - margin
        assert hi > lo
        return 1.0 - float(self.C[hi] - self.C[lo]) / (hi - lo), hi - lo


def draft_reads_las(d, contigs, min_ovl=1500):
    """(formats.LasRecords, [ContigOnGenome]) of the contigs (uint8 base arrays, as written to the draft DB) against the reads of the
    synth_draft data set d: both directed... no - the A side is always the contig (what `hinge consensus` reads)."""
    from hinge_amd import formats, synth_draft as sd
    assert not d.spec.circular
    n = len(d.reads)
    ts = d.spec.tspace
    tdt = np.uint8 if ts <= 125 else np.dtype("<u2")
    # the reads' tables again (the generator does not keep them): same seed, same draws
    rng = np.random.default_rng(d.spec.seed)
    d2 = sd.generate(d.spec)
    assert all(np.array_equal(x, y) for x, y in zip(d.reads, d2.reads))
    Ps, Cs = _read_tables(d.spec)
    placed, recs, traces = [], [], []
    for ci, seq in enumerate(contigs):
        c = ContigOnGenome(seq, d.genome)
        placed.append(c)
        g0e = np.concatenate([d.g0, [c.g0]]); g1e = np.concatenate([d.g1, [c.g1]])
        strande = np.concatenate([d.strand, [c.strand]]).astype(np.uint8)
        rlene = np.concatenate([d.rlen.astype(np.int64), [c.length]])
        Pse, Cse = Ps + [c.P], Cs + [c.C]
        lo = np.maximum(c.g0, d.g0); hi = np.minimum(c.g1, d.g1)
        for y in np.nonzero(hi - lo >= min_ovl)[0]:
            r = sd._record(n, int(y), int(lo[y]), int(hi[y]), 0, g0e, g1e, strande, rlene, Pse, Cse, ts)
            if r is None:
                continue
            h = r[0]
            recs.append((h[0], h[1], h[2], h[3], h[4], h[5], h[6], ci, int(y)))
            traces.append(np.asarray(r[1], dtype=tdt).reshape(-1))
    order = sorted(range(len(recs)), key=lambda i: (recs[i][7], recs[i][8], recs[i][6], recs[i][2]))
    rec = np.zeros(len(recs), dtype=formats.LAS_REC_DTYPE)
    for o, i in enumerate(order):
        rec[o] = recs[i]
    tr = [traces[i].view(np.uint8) for i in order]
    toff = np.concatenate([[0], np.cumsum([len(t) for t in tr])]).astype(np.int64)
    return formats.LasRecords(ts, rec, np.concatenate(tr) if tr else np.zeros(0, np.uint8), toff), placed


def _read_tables(spec):
    """The P / C tables of every read of synth_draft.generate(spec): the generator's own draws, replayed."""
    from hinge_amd import synth_draft as sd
    out_P, out_C = [], []
    orig = sd._make_read

    def spy(rng, gseq, sp):
        r = orig(rng, gseq, sp)
        out_P.append(r[1]); out_C.append(r[2])
        return r
    sd._make_read = spy
    try:
        sd.generate(spec)
    finally:
        sd._make_read = orig
    return out_P, out_C


def polish(d, wd, draft_fasta: bytes, run_consensus, min_len=5000):
    """`hinge consensus` over the contigs of a draft FASTA: writes the draft DB (`draft`), the contig-vs-read alignments
    (`draft.G.las`; the read DB is the data set's own `G`) and the [consensus] section, calls run_consensus(wd) -> FASTA bytes,
    and places draft and consensus contigs on the planted genome.  Returns [(name, draft placement, consensus placement, text)]."""
    import os
    from hinge_amd import formats
    import draft_common as dc
    ctgs = [(nm, s) for nm, s in dc.contigs_of(draft_fasta) if len(s) >= min_len]
    assert ctgs, "no contig of %d+ bases" % min_len
    bases = [to_bases(s) for _, s in ctgs]
    las, placed = draft_reads_las(d, bases)
    formats.write_db(os.path.join(wd, "draft"), np.array([len(b) for b in bases], np.int32), bases=bases)
    formats.write_las(os.path.join(wd, "draft.G.las"), las)
    with open(os.path.join(wd, "nominal.ini"), "a") as f:
        f.write("\n[consensus]\nmin_length = 2000;\n")
    fa = run_consensus(wd)
    out = dc.contigs_of(fa)
    assert len(out) == len(ctgs), (len(out), len(ctgs))
    res = []
    for (nm, _), p, (_, seq) in zip(ctgs, placed, out):
        res.append((nm, p, ContigOnGenome(to_bases(seq), d.genome), seq))
    return res
