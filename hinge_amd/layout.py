"""Host side of `hinge layout` over the C ABI, in the pieces a sharded run needs (hinge_amd/dist.py ShardedLayout; one process
with one block is the plain stage).  The installed command-line tool is the C++ program hinge_amd/host/layout_main.cpp; this
is the same stage cut along the exchange steps of SURVEY 8(e):

  block_matches     per block, on its rank's GPU   GetAlignment (hinging.cpp:398-412, 478-602): the active x active pairs in the
                                                   reference's hash-map order (hinge_pick_pairs), ProcessAlignment of the best one
                                                   or two overlaps of each (k_trim_classify)
  weight_order      every rank, the gathered rows  std::sort(compare_overlap_weight) of every read's two lists (:1066-1071)
  hinge_queries     per block, on its rank's GPU   GetMatchingPosition of every hinge of the block's reads through every eligible
                                                   match (k_matching_position; :1365-1640 needs nothing else from the traces)
  bookkeeping       every rank, the gathered rows  hinges / killed hinges (:1170-1208), hinges removed by a bridging match
                                                   (:1262-1321), the hinge graph and its components (:1365-1675): sequential and
                                                   global in the reference, a few thousand rows - every rank runs it on the same input
  select            per block, on its rank's GPU   the greedy selection (:1911-2148, k_select_edges) of the block's own reads
  print_files       rank 0, the gathered picks     PrintOverlapToFile(2) lines, .edges.skipped, .deadends.txt, .hinge.list ...

All arithmetic on overlaps and trace points happens in the HIP kernels; what is here is bookkeeping on a few ints per match.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Sequence, Tuple

import numpy as np

FORWARD, BACKWARD, BCOVERA, FORWARD_INTERNAL, BACKWARD_INTERNAL = 0, 1, 3, 12, 13   # LAInterface.h:30-33
# columns of a match row (int32): what the later steps and the printers read of an LOverlap (LAInterface.h:76-110)
(M_A, M_B, M_COMP, M_TYPE, M_ACTIVE, M_WEIGHT, M_LENGTH, M_EAB, M_EAE, M_EBB, M_EBE, M_AB, M_AE, M_BB, M_BE, M_OVL, M_DIR) = range(17)
M_COLS = 18
# columns of a hinge-query row (int32): read, hinge index, direction of the list, B read, comp, match type, matching position on B
(Q_A, Q_K, Q_DIR, Q_B, Q_COMP, Q_TYPE, Q_POSB) = range(7)
Q_COLS = 8


@dataclass
class LayoutParams:
    """The ini values `hinge layout` reads (hinging.cpp:775-803)."""
    length_threshold: int
    aln_threshold: int
    theta: int
    theta2: int = 0
    hinge_slack: int = 1000
    hinge_tolerance: int = 150
    kill_hinge_overlap: int = 300
    kill_hinge_internal: int = 40
    matching_hinge_slack: int = 200
    min_connected_component_size: int = 8
    use_two_matches: bool = True

    @staticmethod
    def from_ini(ini) -> "LayoutParams":
        g = ini.get_int
        return LayoutParams(g("filter", "length_threshold", -1), g("filter", "aln_threshold", -1), g("filter", "theta", -1), g("filter", "theta2", 0),
                            g("layout", "hinge_slack", 1000), g("layout", "hinge_tolerance", 150), g("layout", "kill_hinge_overlap", 300),
                            g("layout", "kill_hinge_internal", 40), g("layout", "matching_hinge_slack", 200),
                            g("layout", "min_connected_component_size", 8), bool(g("layout", "use_two_matches", 1)))


def read_pairs_file(path: str, n_read: int) -> List[List[Tuple[int, int]]]:
    """.repeat.txt / .hinges.txt as hinging.cpp:887-936 reads them: `read` then (r1, r2) pairs, kept only if both are non-zero."""
    out: List[List[Tuple[int, int]]] = [[] for _ in range(n_read)]
    for line in open(path):
        tok = line.split()
        if not tok:
            continue
        num, rest = int(tok[0]), [int(t) for t in tok[1:]]
        out[num] = []
        for k in range(0, len(rest), 2):
            r1, r2 = rest[k], (rest[k + 1] if k + 1 < len(rest) else 0)
            if r1 != 0 and r2 != 0:
                out[num].append((r1, r2))
    return out


def initial_activity(eff: np.ndarray, maximal: np.ndarray, P: LayoutParams) -> Tuple[np.ndarray, List[int]]:
    """(active, garbage): masks shorter than length_threshold go to .garbage.txt (hinging.cpp:954-960), then only the reads of
    .max stay active (:398-412)."""
    eff = np.asarray(eff, dtype=np.int64).reshape(-1, 2)
    short = (eff[:, 1] - eff[:, 0]) < P.length_threshold
    return (~short & np.asarray(maximal, dtype=bool)).astype(np.uint8), np.nonzero(short)[0].tolist()


def block_matches(classify, pile, lo: int, hi: int, active: np.ndarray, P: LayoutParams) -> Tuple[np.ndarray, List[int]]:
    """GetAlignment for the reads [lo, hi) of one block.  classify(sel, a_of) -> int32 [n, 10] is ProcessAlignment on the block's
    GPU (capi.Context.trim_classify).  Returns the match rows of matches_forward / matches_backward in the order the reference
    appends them, and the reads it would report as contained ("Should not happen": never after `hinge maximal`)."""
    from .dist import pick_best_pairs
    sel, a_of = pick_best_pairs(pile.row_ptr, pile.a_span, pile.b_span, pile.b_flag, lo, hi, active, P.use_two_matches,
                                self_before=pile.self_before, both_active=True, n_sorts=1)
    if len(sel) == 0:
        return np.zeros((0, M_COLS), np.int32), []
    cls = np.asarray(classify(sel, a_of), dtype=np.int32).reshape(-1, 10)
    b = (pile.b_flag[sel] & np.uint32(0x7FFFFFFF)).astype(np.int32)
    comp = (pile.b_flag[sel] >> np.uint32(31)).astype(np.int32)
    typ = cls[:, 4]
    contained = sorted(set(a_of[(typ == BCOVERA) & (cls[:, 5] != 0) & (np.asarray(active)[b] != 0)].tolist()))
    fwd = (typ == FORWARD) | (typ == FORWARD_INTERNAL)
    keep = fwd | (typ == BACKWARD) | (typ == BACKWARD_INTERNAL)
    rows = np.zeros((int(keep.sum()), M_COLS), np.int32)
    k = np.nonzero(keep)[0]
    rows[:, M_A], rows[:, M_B], rows[:, M_COMP] = a_of[k], b[k], comp[k]
    rows[:, M_TYPE], rows[:, M_ACTIVE], rows[:, M_WEIGHT], rows[:, M_LENGTH] = typ[k], cls[k, 5], cls[k, 6], cls[k, 7]
    rows[:, M_EAB], rows[:, M_EAE], rows[:, M_EBB], rows[:, M_EBE] = cls[k, 0], cls[k, 1], cls[k, 2], cls[k, 3]
    rows[:, M_AB], rows[:, M_AE] = pile.a_span[sel[k], 0], pile.a_span[sel[k], 1]
    rows[:, M_BB], rows[:, M_BE] = pile.b_span[sel[k], 0], pile.b_span[sel[k], 1]
    rows[:, M_OVL] = sel[k]                 # index into the OWNER's pile-up arrays: only its rank asks the GPU about this overlap again
    rows[:, M_DIR] = np.where(fwd[k], 0, 1)
    return rows, contained


def weight_order(rows: np.ndarray, n_read: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """matches_forward[i] / matches_backward[i] of every read after std::sort(compare_overlap_weight) (hinging.cpp:1066-1071;
    equal weights where libstdc++ leaves them: hinge_sort_order_desc on the list in append order).  Returns (rows re-ordered:
    all forward lists, then all backward lists; off_fwd, off_bwd) in the layout hinge_select_edges takes."""
    from . import capi
    out = np.zeros_like(rows)
    offs = []
    at = 0
    for d in (0, 1):
        part = np.nonzero(rows[:, M_DIR] == d)[0]
        a = rows[part, M_A]
        order = np.argsort(a, kind="stable")                   # reads ascending, append order inside a read
        part, a = part[order], a[order]
        counts = np.bincount(a, minlength=n_read).astype(np.int64)
        off = np.concatenate([[0], np.cumsum(counts)]) + at
        for i in np.nonzero(counts > 1)[0]:
            s, e = int(off[i] - at), int(off[i + 1] - at)
            w = rows[part[s:e], M_WEIGHT].astype(np.int64)
            if np.any(w[1:] > w[:-1]) or len(w) > 16:          # (already descending and short: an insertion sort leaves it alone)
                part[s:e] = part[s:e][capi.sort_order_desc(w, 1)]
        out[at:at + len(part)] = rows[part]
        offs.append(off.astype(np.int64))
        at += len(part)
    return out, offs[0], offs[1]


def hinge_queries(rows: np.ndarray, off_fwd: np.ndarray, off_bwd: np.ndarray, lo: int, hi: int, active: np.ndarray,
                  hinges: Sequence[Sequence[Tuple[int, int]]]) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """The (overlap, position) pairs the hinge graph asks GetMatchingPosition for, for the reads [lo, hi), in the graph loop's
    order (hinging.cpp:1365-1640: read, hinge, forward list, backward list).  Returns (query rows without Q_POSB, q_ovl, q_pos)."""
    q, ovl, pos = [], [], []
    for i in range(lo, hi):
        if not active[i] or not hinges[i]:
            continue
        lists = ((0, range(int(off_fwd[i]), int(off_fwd[i + 1])), (FORWARD, FORWARD_INTERNAL)),
                 (1, range(int(off_bwd[i]), int(off_bwd[i + 1])), (BACKWARD, BACKWARD_INTERNAL)))
        for k, (hpos, _) in enumerate(hinges[i]):
            for d, rng, kinds in lists:
                for j in rng:
                    m = rows[j]
                    if m[M_ACTIVE] and m[M_TYPE] in kinds and active[m[M_B]]:
                        q.append((i, k, d, int(m[M_B]), int(m[M_COMP]), int(m[M_TYPE]), 0, 0))
                        ovl.append(int(m[M_OVL]))
                        pos.append(int(hpos))
    return (np.array(q, np.int32).reshape(-1, Q_COLS), np.array(ovl, np.int64), np.array(pos, np.int32))


def bookkeeping(n_read: int, active: np.ndarray, rows: np.ndarray, off_fwd: np.ndarray, off_bwd: np.ndarray, queries: np.ndarray,
                repeats: Sequence[Sequence[Tuple[int, int]]], hinges: Sequence[Sequence[Tuple[int, int]]], P: LayoutParams) -> Dict[str, object]:
    """hinging.cpp:1170-1208 (hinges, killed hinges), :1262-1321 (hinges a bridging match removes), :1365-1640 (hinge graph over
    the gathered matching positions, new killed hinges), :1644-1675 (components smaller than min_connected_component_size)."""
    h_active = [[True] * len(hinges[i]) for i in range(n_read)]
    killed = []
    for i in range(n_read):
        surviving = set(hinges[i])
        killed.append([(pos, typ) for (pos, typ) in repeats[i] if (pos, typ) not in surviving])
    ko, ki = P.kill_hinge_overlap, P.kill_hinge_internal
    for i in range(n_read):
        if not active[i] or not hinges[i]:
            continue
        for j in range(int(off_fwd[i]), int(off_fwd[i + 1])):
            m = rows[j]
            if m[M_ACTIVE] and active[m[M_B]]:
                for k, (pos, typ) in enumerate(hinges[i]):
                    if typ == 1 and ((m[M_TYPE] == FORWARD_INTERNAL and m[M_EAB] < pos + ki) or (m[M_TYPE] == FORWARD and m[M_EAB] < pos - ko)):
                        h_active[i][k] = False
        for j in range(int(off_bwd[i]), int(off_bwd[i + 1])):
            m = rows[j]
            if m[M_ACTIVE] and active[m[M_B]]:
                for k, (pos, typ) in enumerate(hinges[i]):
                    if typ == -1 and ((m[M_TYPE] == BACKWARD_INTERNAL and m[M_EAE] > pos - ki) or (m[M_TYPE] == BACKWARD and m[M_EAE] > pos + ko)):
                        h_active[i][k] = False
    # hinge graph: union-find over (read, hinge) nodes numbered read by read (a hinge without an edge is a component of one)
    base = np.concatenate([[0], np.cumsum([len(h) for h in hinges])]).astype(np.int64)
    parent = list(range(int(base[-1])))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    hgraph: List[str] = []
    new_killed: List[List[Tuple[int, int]]] = [[] for _ in range(n_read)]
    slack = P.matching_hinge_slack
    for qrow in queries.tolist():
        i, k, d, bid, comp, mtype, pos_b = qrow[:7]
        hpos, htyp = hinges[i][k]
        want = -htyp if comp else htyp
        straight = 1 if d == 0 else -1
        for l, (bpos, btyp) in enumerate(hinges[bid]):
            if pos_b - slack < bpos < pos_b + slack and want == btyp:
                parent[find(int(base[i]) + k)] = find(int(base[bid]) + l)
                hgraph.append("%d %d %d %d %d %d" % ((i, bid, hpos, bpos, 1, comp) if htyp == straight else (bid, i, bpos, hpos, 1, comp)))
        for (bpos, btyp) in killed[bid]:
            if pos_b - slack < bpos < pos_b + slack:
                if want == btyp:
                    hgraph.append("%d %d %d %d %d %d" % ((i, bid, hpos, bpos, 0, comp) if htyp == straight else (bid, i, bpos, hpos, 0, comp)))
                # (the forward block collects inside the type test, the backward block outside it: hinging.cpp:1478 / :1617)
                if (d == 0 and want == btyp and mtype == FORWARD) or (d == 1 and mtype == BACKWARD):
                    new_killed[i].append((hpos, htyp))
    size: Dict[int, int] = {}
    for x in range(len(parent)):
        r = find(x)
        size[r] = size.get(r, 0) + 1
    for i in range(n_read):
        for k in range(len(hinges[i])):
            if size[find(int(base[i]) + k)] < P.min_connected_component_size:
                h_active[i][k] = False
    return {"h_active": h_active, "killed": killed, "new_killed": new_killed, "hgraph": hgraph}


def selection_tables(n_read: int, rows: np.ndarray, hinges, h_active, new_killed):
    """The arrays hinge_select_edges takes besides the offsets: match_rec[., 9], h_off, h_rec[., 3], k_off, k_rec[., 2]."""
    rec = np.ascontiguousarray(rows[:, [M_B, M_COMP, M_TYPE, M_ACTIVE, M_WEIGHT, M_EBB, M_EBE, M_BB, M_BE]], dtype=np.int32)
    h_off = np.concatenate([[0], np.cumsum([len(h) for h in hinges])]).astype(np.int64)
    h_rec = np.array([(pos, typ, 1 if h_active[i][k] else 0) for i in range(n_read) for k, (pos, typ) in enumerate(hinges[i])], np.int32).reshape(-1, 3)
    k_off = np.concatenate([[0], np.cumsum([len(k) for k in new_killed])]).astype(np.int64)
    k_rec = np.array([pk for i in range(n_read) for pk in new_killed[i]], np.int32).reshape(-1, 2)
    return rec, h_off, h_rec, k_off, k_rec


def edge_line(m: np.ndarray, eff: np.ndarray) -> str:
    """PrintOverlapToFile (hinging.cpp:188-248)."""
    a, b = int(m[M_A]), int(m[M_B])
    hinged = -1 if m[M_TYPE] in (FORWARD, BACKWARD) else 1
    if m[M_TYPE] in (FORWARD, FORWARD_INTERNAL):
        v = (a, b, m[M_LENGTH], 0, m[M_COMP], hinged, m[M_EAB], m[M_EAE], m[M_EBB], m[M_EBE], eff[a][0], eff[a][1], eff[b][0], eff[b][1])
    else:
        v = (b, a, m[M_LENGTH], m[M_COMP], 0, hinged, m[M_EBB], m[M_EBE], m[M_EAB], m[M_EAE], eff[b][0], eff[b][1], eff[a][0], eff[a][1])
    return "%d %d %d %d %d %d [%d %d] [%d %d] [%d %d] [%d %d]" % tuple(int(x) for x in v) + " [%d %d] [%d %d]" % (m[M_AB], m[M_AE], m[M_BB], m[M_BE])


def edge_line2(m: np.ndarray, eff: np.ndarray, hinge_pos: int) -> str:
    """PrintOverlapToFile2 (hinging.cpp:253-344)."""
    a, b, t = int(m[M_A]), int(m[M_B]), int(m[M_TYPE])
    if t in (FORWARD, FORWARD_INTERNAL):
        v = (a, b, m[M_LENGTH], 0, m[M_COMP], 0 if t == FORWARD else 1, -1 if t == FORWARD else hinge_pos,
             m[M_EAB], m[M_EAE], m[M_EBB], m[M_EBE], eff[a][0], eff[a][1], eff[b][0], eff[b][1])
    else:
        v = (b, a, m[M_LENGTH], m[M_COMP], 0, 0 if t == BACKWARD else -1, -1 if t == BACKWARD else hinge_pos,
             m[M_EBB], m[M_EBE], m[M_EAB], m[M_EAE], eff[b][0], eff[b][1], eff[a][0], eff[a][1])
    return "%d %d %d %d %d %d %d [%d %d] [%d %d] [%d %d] [%d %d]" % tuple(int(x) for x in v)


def print_files(n_read: int, active: np.ndarray, eff: np.ndarray, rows: np.ndarray, off_fwd: np.ndarray, off_bwd: np.ndarray,
                chosen: np.ndarray, hpos: np.ndarray, poison: np.ndarray, hinges, book: Dict[str, object], garbage: List[int]) -> Dict[str, List[str]]:
    """The files of hinging.cpp:1201-1208 (.killed.hinges), :1354-1491 (.hgraph), :1694-1704 (.hinge.list), :1911-2148
    (.edges.hinges, .edges.hinges2, .edges.skipped, .deadends.txt) and :954-958 (.garbage.txt), as lists of lines.
    chosen / hpos: [2, n_read] from the selection (index into rows, -1 = dead end); poison: hits per row."""
    edges, edges2, skipped, deadends = [], [], [], []
    for i in range(n_read):
        if not active[i]:
            continue
        for d, off, name in ((0, off_fwd, "forward"), (1, off_bwd, "backward")):
            for j in range(int(off[i]), int(off[i + 1])):              # the walk prints a poisoned match once per poisoning hinge,
                skipped += [edge_line(rows[j], eff)] * int(poison[j])  # in list order, before it reaches what it keeps
            c = int(chosen[d][i])
            if c >= 0:
                edges.append(edge_line(rows[c], eff))
                edges2.append(edge_line2(rows[c], eff, int(hpos[d][i])))
            else:
                deadends.append("%d\t matches_%s size: %d" % (i, name, int(off[i + 1] - off[i])))
    h_active, killed = book["h_active"], book["killed"]
    return {".garbage.txt": ["%d" % i for i in garbage],
            ".killed.hinges": ["%d " % i + "".join("%d %d " % (typ, pos) for (pos, typ) in killed[i]) for i in range(n_read)],
            ".hgraph": list(book["hgraph"]),
            ".hinge.list": ["%d %d %d" % (i, pos, typ) for i in range(n_read) for k, (pos, typ) in enumerate(hinges[i]) if active[i] and h_active[i][k]],
            ".edges.hinges": edges, ".edges.hinges2": edges2, ".edges.skipped": skipped, ".deadends.txt": deadends}
