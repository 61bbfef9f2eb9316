"""On-disk formats of the HINGE filter / maximal / layout path (numpy side).

Writers and readers for the DAZZ_DB stub/index/track files and the DALIGNER
``.las`` overlap file, used by the synthetic-data generator, the tests and
``bench.py``.  The product's C++ ingest (``hinge_amd/host/las_reader.cpp``,
``db_reader.cpp``) reads the same bytes; this module is the independent
Python statement of the formats so the two can be checked against each other.

Reference for the layouts (file:line under the reference tree):
  * ``.las`` header ``int64 novl; int32 tspace``  - src/lib/LAInterface.cpp:604-605
  * ``.las`` record = ``Overlap`` minus its leading pointer (40 bytes)
                                                   - src/include/align.h:126-132,332-337,
                                                     src/lib/align.c:3042-3049
  * trace = ``tlen`` bytes (tspace <= 125) of (diffs, b-advance) pairs
                                                   - src/include/align.h:98-110,
                                                     src/lib/LAInterface.cpp:607-614
  * ``NAME.db`` text stub                          - src/include/DB.h:299-303
  * ``.NAME.idx`` = HITS_DB (112 B) + n x HITS_READ (40 B)
                                                   - src/include/DB.h:214-288
  * ``.NAME.qual.anno/.data`` track                - src/lib/DB.c:1097-1100,1238-1270
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

TRACE_XOVR = 125          # src/include/align.h:58
DB_QV = 0x03FF            # src/include/DB.h:210
DB_CSS = 0x0400
DB_BEST = 0x0800          # src/include/DB.h:212
COMP_FLAG = 0x1           # src/include/align.h:155  COMP(x) = (x & 0x1)

HITS_DB_FMT = "<iiii4fi4xqiiiii4xqi4xqqq"       # 112 bytes
HITS_DB_SIZE = struct.calcsize(HITS_DB_FMT)
assert HITS_DB_SIZE == 112

HITS_READ_DTYPE = np.dtype(
    {
        "names": ["origin", "rlen", "fpulse", "boff", "coff", "flags"],
        "formats": ["<i4", "<i4", "<i4", "<i8", "<i8", "<i4"],
        "offsets": [0, 4, 8, 16, 24, 32],
        "itemsize": 40,
    }
)

LAS_REC_DTYPE = np.dtype(
    {
        "names": ["tlen", "diffs", "abpos", "bbpos", "aepos", "bepos", "flags", "aread", "bread"],
        "formats": ["<i4", "<i4", "<i4", "<i4", "<i4", "<i4", "<u4", "<i4", "<i4"],
        "offsets": [0, 4, 8, 12, 16, 20, 24, 28, 32],
        "itemsize": 40,
    }
)


# --------------------------------------------------------------------------------------
# DAZZ_DB
# --------------------------------------------------------------------------------------

def db_paths(db_name: str) -> Tuple[str, str, str, str]:
    """(stub, idx, bps, track-prefix) for ``--db db_name`` (name may or may not end in .db)."""
    d, b = os.path.split(db_name)
    if b.endswith(".db"):
        b = b[:-3]
    d = d or "."
    return (
        os.path.join(d, b + ".db"),
        os.path.join(d, "." + b + ".idx"),
        os.path.join(d, "." + b + ".bps"),
        os.path.join(d, "." + b),
    )


def write_db(
    db_name: str,
    rlen: np.ndarray,
    block_first: Optional[Sequence[int]] = None,
    cutoff: int = 0,
    all_flag: int = 1,
    flags: Optional[np.ndarray] = None,
    write_bases: bool = True,
    bases: Optional[Sequence[np.ndarray]] = None,
) -> None:
    """Write NAME.db, .NAME.idx and (optionally) .NAME.bps for reads of the given lengths.
    bases: one uint8 array of 0..3 (A C G T) per read - packed four to a byte, first base in the two high bits
    (Compress_Read, src/lib/DB.c:239-262); without it the .bps is all zeroes (the graph stages never read it).

    ``block_first`` = untrimmed first-read index of every block plus the total (DBsplit's
    table); defaults to a single block.  With ``cutoff=0, all_flag=1`` nothing is trimmed.
    """
    rlen = np.asarray(rlen, dtype=np.int32)
    n = int(rlen.shape[0])
    stub, idx, bps, _ = db_paths(db_name)
    if flags is None:
        flags = np.full(n, DB_BEST | 850, dtype=np.int32)
    flags = np.asarray(flags, dtype=np.int32)
    keep = ((flags & DB_BEST) >= (0 if all_flag else DB_BEST)) & (rlen >= cutoff)
    treads = int(keep.sum())
    if block_first is None:
        block_first = [0, n]
    block_first = list(block_first)
    tcum = np.concatenate([[0], np.cumsum(keep)])
    with open(stub, "w") as f:
        f.write("files = %9d\n" % 1)
        f.write("  %9d %s %s\n" % (n, "synth", "synth"))
        f.write("blocks = %9d\n" % (len(block_first) - 1))
        f.write("size = %9d cutoff = %9d all = %1d\n" % (200, cutoff, all_flag))
        for u in block_first:
            f.write(" %9d %9d\n" % (u, int(tcum[u])))
    rec = np.zeros(n, dtype=HITS_READ_DTYPE)
    rec["origin"] = np.arange(n, dtype=np.int32)
    rec["rlen"] = rlen
    rec["fpulse"] = 0
    nbytes = (rlen.astype(np.int64) + 3) >> 2
    rec["boff"] = np.concatenate([[0], np.cumsum(nbytes)[:-1]]) if n else 0
    rec["coff"] = -1
    rec["flags"] = flags
    hdr = struct.pack(
        HITS_DB_FMT,
        n, treads, cutoff, all_flag, 0.25, 0.25, 0.25, 0.25,
        int(rlen.max()) if n else 0, int(rlen.astype(np.int64).sum()),
        n, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    )
    with open(idx, "wb") as f:
        f.write(hdr)
        f.write(rec.tobytes())
    if write_bases and bases is not None:
        with open(bps, "wb") as f:
            for i in range(n):
                f.write(pack_bases(bases[i]).tobytes())
    elif write_bases:
        with open(bps, "wb") as f:
            total = int(nbytes.sum())
            chunk = bytes(1 << 20)
            while total > 0:
                w = min(total, len(chunk))
                f.write(chunk[:w])
                total -= w


def pack_bases(b: np.ndarray) -> np.ndarray:
    """uint8 0..3 per base -> DAZZ_DB's 2-bit form, (len + 3) // 4 bytes, the tail padded with zeroes."""
    b = np.asarray(b, dtype=np.uint8)
    pad = (-len(b)) % 4
    q = np.concatenate([b, np.zeros(pad, np.uint8)]).reshape(-1, 4)
    return ((q[:, 0] << 6) | (q[:, 1] << 4) | (q[:, 2] << 2) | q[:, 3]).astype(np.uint8)


def read_bases(db_name: str, idx: Optional[dict] = None) -> List[np.ndarray]:
    """Every (trimmed) read's bases (uint8 0..3) from .NAME.bps."""
    if idx is None:
        idx = read_db_index(db_name)
    _, _, bps, _ = db_paths(db_name)
    raw = np.fromfile(bps, dtype=np.uint8)
    out = []
    for ln, off in zip(idx["rlen"], idx["boff"]):
        pk = raw[int(off):int(off) + (int(ln) + 3) // 4]
        un = np.stack([(pk >> 6) & 3, (pk >> 4) & 3, (pk >> 2) & 3, pk & 3], axis=1).reshape(-1)[:int(ln)]
        out.append(un.astype(np.uint8))
    return out


def read_db_index(db_name: str) -> dict:
    """Read .NAME.idx + NAME.db; returns trimmed read lengths the way Open_DB+Trim_DB see them
    (src/lib/DB.c:395-578, 585-683)."""
    stub, idx, _, _ = db_paths(db_name)
    with open(idx, "rb") as f:
        hdr = struct.unpack(HITS_DB_FMT, f.read(HITS_DB_SIZE))
        ureads, treads = hdr[0], hdr[1]
        rec = np.frombuffer(f.read(ureads * 40), dtype=HITS_READ_DTYPE)
    cutoff, all_flag = 0, 1
    with open(stub) as f:
        lines = f.read().split("\n")
    for ln in lines:
        if ln.startswith("size ="):
            toks = ln.replace("=", " ").split()
            cutoff = int(toks[3])
            all_flag = int(toks[5])
    if cutoff <= 0 and all_flag:
        keep = np.ones(ureads, dtype=bool)
    else:
        keep = ((rec["flags"] & DB_BEST) >= (0 if all_flag else DB_BEST)) & (rec["rlen"] >= cutoff)
    return {
        "ureads": ureads,
        "treads": treads,
        "cutoff": cutoff,
        "all": all_flag,
        "rlen": rec["rlen"][keep].astype(np.int32),
        "keep": keep,
        "boff": rec["boff"][keep].astype(np.int64),
    }


def write_qual_track(db_name: str, qv_per_read: List[np.ndarray]) -> None:
    """Write the ``qual`` track (.NAME.qual.anno / .NAME.qual.data): one byte per tspace segment."""
    _, _, _, pre = db_paths(db_name)
    n = len(qv_per_read)
    offs = np.zeros(n + 1, dtype=np.int64)
    for i, q in enumerate(qv_per_read):
        offs[i + 1] = offs[i] + len(q)
    with open(pre + ".qual.anno", "wb") as f:
        f.write(struct.pack("<ii", n, 8))
        f.write(offs.tobytes())
    with open(pre + ".qual.data", "wb") as f:
        for q in qv_per_read:
            f.write(np.asarray(q, dtype=np.uint8).tobytes())


def read_qual_track(db_name: str) -> Optional[List[np.ndarray]]:
    _, _, _, pre = db_paths(db_name)
    if not os.path.exists(pre + ".qual.anno"):
        return None
    with open(pre + ".qual.anno", "rb") as f:
        tracklen, size = struct.unpack("<ii", f.read(8))
        offs = np.frombuffer(f.read(8 * (tracklen + 1)), dtype=np.int64)
    data = np.fromfile(pre + ".qual.data", dtype=np.uint8)
    out = [data[offs[i]:offs[i + 1]] for i in range(tracklen)]
    # A track with one entry per UNTRIMMED read on a trimmed DB: the entries of the reads Trim_DB drops are skipped (what the
    # oracle and the C++ host reader do; the reference's own getQV crashes on this combination, tests/test_oracle_pinned.py)
    try:
        idx = read_db_index(db_name)
    except OSError:
        return out
    keep = idx.get("keep")
    if keep is not None and tracklen == len(keep) and int(np.sum(keep)) != tracklen:
        out = [q for q, k in zip(out, keep) if k]
    return out


# --------------------------------------------------------------------------------------
# .las
# --------------------------------------------------------------------------------------

@dataclass
class LasRecords:
    """Raw .las records exactly as stored on disk (B coordinates of complemented overlaps are in
    the reverse-complemented B frame) plus the concatenated trace bytes."""

    tspace: int
    rec: np.ndarray            # LAS_REC_DTYPE [novl]
    trace: np.ndarray          # uint8 [sum tlen * tbytes]
    trace_off: np.ndarray      # int64 [novl + 1] byte offsets into trace

    @property
    def novl(self) -> int:
        return int(self.rec.shape[0])


def write_las(path: str, recs: LasRecords) -> None:
    tbytes = 1 if recs.tspace <= TRACE_XOVR else 2
    novl = recs.novl
    rec_b = recs.rec.view(np.uint8).reshape(novl, 40) if novl else np.zeros((0, 40), np.uint8)
    tlen_b = (recs.trace_off[1:] - recs.trace_off[:-1]).astype(np.int64)
    assert np.array_equal(tlen_b, recs.rec["tlen"].astype(np.int64) * tbytes)
    # interleave records and traces into one byte buffer: both keep their relative order, so a
    # boolean "is header byte" mask places them without per-byte index arrays
    out_off = np.concatenate([[0], np.cumsum(40 + tlen_b)]).astype(np.int64)
    total = int(out_off[-1])
    buf = np.zeros(total, dtype=np.uint8)
    if novl:
        marker = np.zeros(total + 1, dtype=np.int8)
        marker[out_off[:-1]] = 1
        if tlen_b.min() > 0:
            marker[out_off[:-1] + 40] = -1
        else:
            np.add.at(marker, out_off[:-1] + 40, -1)  # a zero-length trace makes two records adjacent
        is_hdr = np.cumsum(marker[:-1], dtype=np.int8).astype(bool)
        buf[is_hdr] = rec_b.reshape(-1)
        buf[~is_hdr] = recs.trace
    with open(path, "wb") as f:
        f.write(struct.pack("<qi", novl, recs.tspace))
        f.write(buf.tobytes())


def read_las(path: str) -> LasRecords:
    """Sequential header-hop parse (record boundaries are only discoverable through tlen)."""
    raw = np.fromfile(path, dtype=np.uint8)
    novl, tspace = struct.unpack("<qi", raw[:12].tobytes())
    tbytes = 1 if tspace <= TRACE_XOVR else 2
    rec = np.zeros(novl, dtype=LAS_REC_DTYPE)
    toff = np.zeros(novl + 1, dtype=np.int64)
    pos = 12
    starts = np.zeros(novl, dtype=np.int64)
    mv = memoryview(raw)
    for i in range(novl):
        starts[i] = pos
        tlen = struct.unpack_from("<i", mv, pos)[0]
        toff[i + 1] = toff[i] + tlen * tbytes
        pos += 40 + tlen * tbytes
    if novl:
        idx = starts[:, None] + np.arange(40)[None, :]
        rec = raw[idx.reshape(-1)].reshape(novl, 40).copy().view(LAS_REC_DTYPE).reshape(novl)
        tl = (toff[1:] - toff[:-1])
        tot = int(tl.sum())
        owner = np.repeat(np.arange(novl), tl)
        within = np.arange(tot) - np.repeat(toff[:-1], tl)
        trace = raw[starts[owner] + 40 + within].copy()
    else:
        trace = np.zeros(0, np.uint8)
    return LasRecords(tspace=tspace, rec=rec, trace=trace, trace_off=toff)


@dataclass
class Pileups:
    """SoA pile-up arrays in the layout the C-ABI takes (include/hinge_hip.h): the overlaps of
    every A read in .las order, self-overlaps removed, B coordinates on the forward strand
    (the flip of src/lib/LAInterface.cpp:1619-1626 already applied)."""

    n_reads: int
    row_ptr: np.ndarray     # int64 [n_reads + 1]
    a_span: np.ndarray      # int32 [n, 2]  (abpos, aepos)
    b_span: np.ndarray      # int32 [n, 2]  (bbpos, bepos) forward strand
    b_flag: np.ndarray      # uint32 [n]    bread | comp << 31
    las_index: np.ndarray   # int64 [n] index of the record in the .las file
    self_a: np.ndarray      # int32 [m] A id of the removed self-overlaps
    self_span: np.ndarray   # int32 [m, 4] abpos, aepos, bbpos', bepos'
    self_before: np.ndarray = None   # int32 [n_reads]: -1, or the kept overlaps of the read in front of its first A == B record (_self_before)

    @property
    def n_ovl(self) -> int:
        return int(self.b_flag.shape[0])


def pileups_from_las(recs: LasRecords, rlen: np.ndarray) -> Pileups:
    r = recs.rec
    n_reads = int(len(rlen))
    comp = (r["flags"] & COMP_FLAG).astype(np.int32)
    blen = np.asarray(rlen, dtype=np.int32)[r["bread"]] if recs.novl else np.zeros(0, np.int32)
    bb = np.where(comp == 1, blen - r["bepos"], r["bbpos"]).astype(np.int32)
    be = np.where(comp == 1, blen - r["bbpos"], r["bepos"]).astype(np.int32)
    is_self = r["aread"] == r["bread"]
    keep = ~is_self
    a = r["aread"][keep]
    if len(a) > 1:
        assert np.all(a[1:] >= a[:-1]), ".las must be sorted by A read"
    counts = np.bincount(a, minlength=n_reads).astype(np.int64)
    row_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    a_span = np.stack([r["abpos"][keep], r["aepos"][keep]], axis=1).astype(np.int32)
    b_span = np.stack([bb[keep], be[keep]], axis=1).astype(np.int32)
    b_flag = (r["bread"][keep].astype(np.uint32) | (comp[keep].astype(np.uint32) << np.uint32(31)))
    self_span = np.stack(
        [r["abpos"][is_self], r["aepos"][is_self], bb[is_self], be[is_self]], axis=1
    ).astype(np.int32)
    return Pileups(
        n_reads=n_reads,
        row_ptr=row_ptr,
        a_span=np.ascontiguousarray(a_span),
        b_span=np.ascontiguousarray(b_span),
        b_flag=np.ascontiguousarray(b_flag),
        las_index=np.nonzero(keep)[0].astype(np.int64),
        self_a=r["aread"][is_self].astype(np.int32),
        self_span=self_span,
        self_before=_self_before(r["aread"], is_self, n_reads),
    )


def las_image_table(recs: LasRecords, pile: Pileups):
    """What hinge_set_las_image takes besides the file's bytes: (win_base[(n + 63) // 64 + 1], rec_rel[n]) - the kept overlaps in
    windows of 64: where every window's first record lies in the image (last entry: where the last kept overlap ends) and every
    kept record's byte offset behind that.  The image itself is np.fromfile(path, np.uint8)."""
    tbytes = 1 if recs.tspace <= TRACE_XOVR else 2
    size = 40 + recs.rec["tlen"].astype(np.int64) * tbytes
    starts = 12 + np.concatenate([[0], np.cumsum(size)[:-1]]).astype(np.int64) if recs.novl else np.zeros(0, np.int64)
    return image_windows(starts[pile.las_index], size[pile.las_index])


def image_windows(kept_start: np.ndarray, kept_size: np.ndarray):
    n = len(kept_start)
    nw = (n + 63) // 64
    win_base = np.zeros(nw + 1, np.int64)
    win_base[:nw] = kept_start[::64]
    win_base[nw] = kept_start[n - 1] + kept_size[n - 1] if n else 12
    rel = kept_start - np.repeat(win_base[:nw], 64)[:n] if n else np.zeros(0, np.int64)
    assert n == 0 or (rel.min() >= 0 and rel.max() < 2 ** 32)
    return win_base, rel.astype(np.uint32)


def _self_before(aread: np.ndarray, is_self: np.ndarray, n_reads: int) -> np.ndarray:
    """For every read: -1 if it has no A == B record, else how many of its OTHER records precede the first one in the file
    (the key A takes part in the insertion order of the reference's (A, B) hash map: include/hinge_hip.h hinge_pick_pairs)."""
    out = np.full(n_reads, -1, np.int32)
    idx = np.nonzero(is_self)[0]
    if len(idx):
        kept_before = np.cumsum(~is_self) - (~is_self)          # kept records in front of record j
        row_first = np.searchsorted(aread, aread[idx], side="left")
        first_of_read = np.ones(len(idx), bool)
        first_of_read[1:] = aread[idx][1:] != aread[idx][:-1]
        sel = idx[first_of_read]
        out[aread[sel]] = (kept_before[sel] - kept_before[row_first[first_of_read]]).astype(np.int32)
    return out


# ---- FASTA + PAF (the reference's second input mode: filter.cpp:289-291,499-503) -----------------------------
def read_name(i: int, length: int) -> str:
    """PacBio-style name whose middle field is the 1-based read id (get_id_from_string, LAInterface.cpp:4808-4819)."""
    return "synth/%d/0_%d" % (i + 1, length)


def write_fasta(path: str, rlen: np.ndarray, seed: int = 0, line: int = 80, gz: bool = False) -> None:
    """Reads of the given lengths (random bases: only the length reaches the path), `line` bases per line."""
    import gzip
    rng = np.random.default_rng(seed)
    alphabet = np.frombuffer(b"ACGT", np.uint8)
    op = gzip.open if gz else open
    with op(path, "wb") as f:
        for i, n in enumerate(np.asarray(rlen).tolist()):
            f.write((">%s\n" % read_name(i, n)).encode())
            seq = alphabet[rng.integers(0, 4, size=n)].tobytes()
            for k in range(0, n, line):
                f.write(seq[k:k + line] + b"\n")


def write_paf(path: str, rlen: np.ndarray, aread, bread, comp, ab, ae, bb, be, gz: bool = False) -> None:
    """One 12-column PAF line per overlap; target coordinates on the forward strand of B (minimap convention)."""
    import gzip
    rlen = np.asarray(rlen)
    op = gzip.open if gz else open
    with op(path, "wb") as f:
        for a, b, c, s0, e0, s1, e1 in zip(np.asarray(aread).tolist(), np.asarray(bread).tolist(), np.asarray(comp).tolist(),
                                           np.asarray(ab).tolist(), np.asarray(ae).tolist(), np.asarray(bb).tolist(), np.asarray(be).tolist()):
            ml = min(e0 - s0, e1 - s1)
            f.write(("%s\t%d\t%d\t%d\t%s\t%s\t%d\t%d\t%d\t%d\t%d\t255\n" % (
                read_name(a, int(rlen[a])), int(rlen[a]), s0, e0, "-" if c else "+", read_name(b, int(rlen[b])), int(rlen[b]), s1, e1,
                ml, max(e0 - s0, e1 - s1))).encode())
