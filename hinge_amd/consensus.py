"""`hinge consensus` (consensus/consensus.cpp:77-288) driven from Python over the C ABI (hinge_consensus_*): the single-rank case
and the sharded one.  The C++ executable (hinge_amd/host/consensus_main.cpp) is what users run; this is its twin for the tests,
for N-GPU runs and for callers that hold the inputs in memory.

Multi-GPU (SURVEY 8(e) for this stage): the path shards by CONTIG - a contig's votes come from its own alignments only - and has
no exchange step: rank r realigns, votes and calls the contigs it owns (dealt by aligned bases, heaviest first), the strings,
statistics and chop offsets are gathered to rank 0, which writes the FASTA and the text in contig order.  No collective sits on
the data path; the one gather at the end is a few MB.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import formats
from .config import IniFile


@dataclass
class Selection:
    """Which alignments vote (consensus.cpp:126-150): per contig the first `count` of its alignments in the order
    std::sort(compare_overlap_aln) leaves them in - remove_multialign works on a COPY of the vector, only its count reaches main()."""
    order: List[np.ndarray]        # per contig: indices into the .las records, sorted
    count: List[int]               # per contig: how many of them are used
    n_listed: int                  # records that pass getAlignment(res, 0, n_alns)'s A-read range filter


def select_alignments(las: formats.LasRecords, n_contigs: int, min_length: int) -> Selection:
    from . import capi
    rec = las.rec
    keep = np.nonzero(rec["aread"].astype(np.int64) + 1 <= las.novl)[0]        # A reads 1 .. n_alns (1-based), LAInterface.cpp:1800-1890
    order, count = [], []
    a = rec["aread"][keep]
    for c in range(n_contigs):
        idx = keep[a == c]
        if len(idx) > 1:
            key = (rec["aepos"][idx].astype(np.int64) - rec["abpos"][idx]) + (rec["bepos"][idx].astype(np.int64) - rec["bbpos"][idx])
            idx = idx[capi.sort_order_desc(key, 1)]
        seen, r = set(), 0
        for k in idx:
            if int(rec["aepos"][k]) - int(rec["abpos"][k]) >= min_length and int(rec["bread"][k]) not in seen:
                seen.add(int(rec["bread"][k]))
                r += 1
        order.append(idx)
        count.append(r)
    return Selection(order, count, int(len(keep)))


@dataclass
class ContigResult:
    text: bytes
    stats: Tuple[int, int, int, int, int, int, int]     # sum_coverage, contig_length, good, insertions, deletions, low coverage, consensus length
    offsets: List[int]                                  # chop_end's return value per used alignment, in the order of use


class HipConsensusBackend:
    """The contigs of `mine` through libhinge_hip (one hinge_consensus_run over all their used alignments)."""

    def __init__(self, ctx, draft_db: str, read_db: str):
        from . import capi
        self.cns = capi.Consensus(ctx, draft_db, read_db)

    def run(self, las: formats.LasRecords, sel: Selection, mine: Sequence[int]) -> Dict[int, ContigResult]:
        picks, first = [], {}
        for c in mine:
            first[c] = len(picks)
            picks.extend(int(k) for k in sel.order[c][:sel.count[c]])
        self.cns.run(las, picks)
        offs = self.cns.offsets()
        out = {}
        for c in mine:
            text, st = self.cns.contig(c)
            out[c] = ContigResult(text, (int(st.sum_coverage), int(st.contig_length), int(st.good_bases), int(st.insertions), int(st.deletions),
                                         int(st.low_coverage_bases), int(st.consensus_length)),
                                  [int(v) for v in offs[first[c]:first[c] + sel.count[c]]])
        return out


def deal_contigs(sel: Selection, las: formats.LasRecords, world: int) -> List[List[int]]:
    """Contigs to ranks by the aligned bases of their voting alignments, heaviest first to the least loaded rank (ties: lowest
    rank): deterministic, the same on every rank."""
    w = []
    for c, idx in enumerate(sel.order):
        used = idx[:sel.count[c]]
        w.append(int((las.rec["aepos"][used].astype(np.int64) - las.rec["abpos"][used]).sum()) if len(used) else 0)
    load = [0] * world
    mine: List[List[int]] = [[] for _ in range(world)]
    for c in sorted(range(len(w)), key=lambda c: (-w[c], c)):
        r = min(range(world), key=lambda r: (load[r], r))
        mine[r].append(c)
        load[r] += w[c] + 1
    return [sorted(m) for m in mine]


def format_outputs(sel: Selection, results: Dict[int, ContigResult], n_reads: int, las_novl: int, contig_len: Sequence[int], min_length: int,
                   nfiles: Tuple[int, int] = (1, 1)) -> Tuple[bytes, bytes]:
    """(FASTA, stdout text) as consensus.cpp writes them."""
    n_contigs = len(contig_len)
    log = ["length threshold:%d" % min_length, "%d files" % nfiles[0], "%d files" % nfiles[1], "# Contigs:%d" % n_contigs, "# Reads:%d" % n_reads,
           "# Alignments:%d" % las_novl, "%d" % sel.n_listed]
    log += ["%d %d" % (c, len(sel.order[c])) for c in range(n_contigs)]
    log.append("Getting read lengths")
    log += ["%d\t%d" % (c, contig_len[c]) for c in range(n_contigs)]
    log.append("Building consensus sequences...")
    fasta = []
    for c in range(n_contigs):
        r = results[c]
        log.append("Contig %d: %d reads" % (c, sel.count[c]))
        fasta.append(b">Consensus%d\n" % c + r.text + b"\n")
        if sel.count[c] == 0:
            continue
        log += ["%d" % v for v in r.offsets]
        s, alen, good, ins, dels, low, clen = r.stats
        log += ["Average coverage: %f" % ((1.0 * s) / alen), "Good bases: %d/%d" % (good, alen), "Insertions: %d/%d" % (ins, alen), "Deletions: %d/%d" % (dels, alen),
                "Low coverage bases: %d/%d" % (low, alen), "Consensus length: %d" % clen]
    return b"".join(fasta), ("\n".join(log) + "\n").encode()


def run_consensus(draft_db: str, read_db: str, las_path: str, out_path: str, config: str, device: int = 0, group=None, backend=None) -> Tuple[int, bytes]:
    """`hinge consensus DRAFT READS LAS OUT INI`: (exit code, stdout text).  With a torch.distributed `group` (or an initialised default
    group) of more than one rank the contigs are sharded over the ranks and rank 0 writes; every rank returns the same text."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    if rank == 0:
        open(out_path, "w").close()                       # (std::ofstream out(name_out) comes first, consensus.cpp:85)
    ini = IniFile(config)
    if ini.error < 0:
        return 1, ("Can't load %s\n" % config).encode()
    min_length = ini.get_int("consensus", "min_length", -1)
    idx1, idx2 = formats.read_db_index(draft_db), formats.read_db_index(read_db)
    las = formats.read_las(las_path)
    n_contigs = len(idx1["rlen"])
    sel = select_alignments(las, n_contigs, min_length)
    mine = deal_contigs(sel, las, world)
    if backend is None:
        from .capi import Context
        backend = HipConsensusBackend(Context(device), draft_db, read_db)
    part = backend.run(las, sel, mine[rank])
    if world > 1:
        gathered: List[Optional[dict]] = [None] * world
        dist.all_gather_object(gathered, part, group=group)
        results: Dict[int, ContigResult] = {}
        for g in gathered:
            results.update(g)
    else:
        results = part
    fasta, text = format_outputs(sel, results, len(idx2["rlen"]), las.novl, [int(v) for v in idx1["rlen"]], min_length)
    if rank == 0:
        with open(out_path, "wb") as f:
            f.write(fasta)
    return 0, text
