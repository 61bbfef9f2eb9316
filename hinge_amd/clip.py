"""`hinge clip` - the first consumer of `hinge layout`'s files: graph construction from `.edges.hinges` /
`.hinge.list`, dead-end clipping, Z-edge clipping and bubble bursting on the strand-symmetric read graph (`G0` / `G1`), loop
resolution (`G2`), with `[layout] aggressive_pruning` Y pruning (`G3`), and their sparsified (`Gs G2s G3s`) and strand-overlaid
(`Gc G2c G3c`) forms, written as GraphML: every graph file the reference's script writes.

Restated from the BEHAVIOUR of the reference's scripts/pruning_and_clipping.py (its reader :1295-1419 and the three operations
:197-262, :331-390, :561-622 as its main body applies them, :1436-1480), not from its text: this module has its own graph type
(the reference drives networkx 1.9 under Python 2), works on (read, strand) pairs instead of "read_strand" strings, and treats a
clipped path and its mirror image as one operation.  Host code by nature - a few thousand vertices, pointer chasing; the reference
runs it in the Python interpreter too.

**PARITY UNPINNED, and knowingly order-dependent.**  The reference cannot run in this image (Python 2, networkx 1.9, ujson,
colormap), so nothing here is compared with its output.  Its results depend on iteration orders this module cannot reproduce:
it walks `set`s of node-name strings and `successors()` lists in CPython 2's hash-table order.  The operations commute
whenever the paths they remove do not touch each other, so for those graphs any order gives the reference's graph; where two
candidate paths overlap (two dead ends into one junction, two Z edges out of one vertex - it keeps its last way out, so the
one visited first is cut -, a bubble - the branch listed FIRST among a vertex's two successors is the one removed), the
reference's own outcome is an accident of string hashes.  Here the orders are fixed and documented:
vertices in order of first appearance in `.edges.hinges`, successors in order of edge insertion.  tests/test_clip.py checks
the three operations on hand-built graphs, the strand symmetry of every result, and order-independence on the layout output
of the synthetic data sets (three of four are order-free; `long_repeat` has two competing Z edges).

Round 4: loop resolution (:625-839 -> `resolve_loops`, `resolve_repeat`) and Y pruning (:841-893 -> `prune_ys`), from their
behaviour as well; the copies loop resolution makes of a repeat's vertices are (read, strand, 'B') here, 'B' + name there.
Round 5: the remaining graphs - `Gs G2s` (+ `G3s`): the graph sparsified to at most 1 000 vertices (`sparsify`: the behaviour of
`random_condensation_sym`, :456-500), `Gc G2c` (+ `G3c`): each with its strand overlay (`overlay_strands`, :1108-1116: every vertex
joined with its mirror image in both directions).  A graph of up to 1 000 vertices is written as it is - the reference's loop does
not run then -, so for those the sparsified files are as deterministic as `G1` / `G2`.  Above 1 000 vertices the reference draws
vertices with an UNSEEDED `random.randrange` (:467): its own files differ from run to run, there is nothing to be identical to.
This module draws from `random.Random(HINGE_CLIP_SEED)` (default 0: reproducible; "random" seeds from the OS as the reference
does) and does what the reference's loop does per draw: a vertex with one way in and one way out, on an unbranched stretch, is
spliced out on both strands, the new edge keeping `intersection` only if both of its halves had it.  Ground-truth and colour
annotation (needs the reference genome / matplotlib) stay out.

    python -m hinge_amd.clip G.edges.hinges G.hinge.list <suffix> [nominal.ini]
"""
from __future__ import annotations

import configparser
import os
import sys
from typing import Dict, Iterable, Iterator, List, Optional, Set, Tuple
from xml.sax.saxutils import escape, quoteattr

Node = Tuple[int, int]          # (read id, strand)


def mirror(v: Node) -> Node:
    """The same read end seen from the other strand (pruning_and_clipping.py:191-194).  A vertex is (read, strand) or - the copy
    loop resolution makes of a repeat's vertices, the reference's 'B' + name - (read, strand, 'B')."""
    return (v[0], 1 - v[1]) + tuple(v[2:])


def node_name(v: Node) -> str:
    return "".join(v[2:]) + "%d_%d" % (v[0], v[1])


class StrandGraph:
    """Directed graph with attribute dictionaries on vertices and edges; adjacency in insertion order (Python dicts)."""

    def __init__(self):
        self.attr: Dict[Node, dict] = {}
        self.out: Dict[Node, Dict[Node, dict]] = {}
        self.inn: Dict[Node, Dict[Node, dict]] = {}

    # ---- construction -------------------------------------------------------------------------------------------------------
    def add_node(self, v: Node) -> None:
        if v not in self.attr:
            self.attr[v] = {}
            self.out[v] = {}
            self.inn[v] = {}

    def add_edge(self, u: Node, v: Node, **attr) -> None:
        """A second add of the same edge updates its attributes in place (what the reference's graph library does)."""
        self.add_node(u)
        self.add_node(v)
        d = self.out[u].get(v)
        if d is None:
            d = {}
            self.out[u][v] = d
            self.inn[v][u] = d
        d.update(attr)

    def copy(self) -> "StrandGraph":
        g = StrandGraph()
        for v, a in self.attr.items():
            g.add_node(v)
            g.attr[v].update(a)
        for u, nb in self.out.items():
            for v, a in nb.items():
                g.add_edge(u, v, **a)
        return g

    # ---- queries ------------------------------------------------------------------------------------------------------------
    def __contains__(self, v: Node) -> bool:
        return v in self.attr

    def __len__(self) -> int:
        return len(self.attr)

    def nodes(self) -> List[Node]:
        return list(self.attr)

    def edges(self) -> Iterator[Tuple[Node, Node, dict]]:
        for u, nb in self.out.items():
            for v, a in nb.items():
                yield u, v, a

    def has_edge(self, u: Node, v: Node) -> bool:
        return u in self.out and v in self.out[u]

    def successors(self, v: Node) -> List[Node]:
        return list(self.out[v])

    def predecessors(self, v: Node) -> List[Node]:
        return list(self.inn[v])

    def out_degree(self, v: Node) -> int:
        return len(self.out[v])

    def in_degree(self, v: Node) -> int:
        return len(self.inn[v])

    def n_edges(self) -> int:
        return sum(len(nb) for nb in self.out.values())

    # ---- removal ------------------------------------------------------------------------------------------------------------
    def remove_edge(self, u: Node, v: Node) -> bool:
        if not self.has_edge(u, v):
            return False
        del self.out[u][v]
        del self.inn[v][u]
        return True

    def remove_node(self, v: Node) -> bool:
        if v not in self.attr:
            return False
        for w in list(self.out[v]):
            del self.inn[w][v]
        for w in list(self.inn[v]):
            del self.out[w][v]
        del self.out[v], self.inn[v], self.attr[v]
        return True

    def is_strand_symmetric(self) -> bool:
        """u -> v present iff mirror(v) -> mirror(u) is, every vertex with its mirror image."""
        return all(mirror(v) in self.attr for v in self.attr) and all(self.has_edge(mirror(v), mirror(u)) for u, v, _ in self.edges())


# ---- reading the layout's files -----------------------------------------------------------------------------------------------
def _bracket(tok: str) -> int:
    return int(tok.strip("[]"))


def read_edges(path: str) -> StrandGraph:
    """`.edges.hinges` -> graph (pruning_and_clipping.py:1314-1371).  A line `A B len sa sb hinged [..] [..] [..] [..] [..] [..]`
    (hinging.cpp:188-248) is the edge A_sa -> B_sb AND its mirror image B_(1-sb) -> A_(1-sa), with A's and B's coordinates
    exchanged on the mirror.  `intersection` = 1 on an edge (and its mirror) that a later line names again."""
    g = StrandGraph()
    with open(path) as f:
        for line in f:
            t = line.split()
            if len(t) < 5:
                continue
            a, b = (int(t[0]), int(t[3])), (int(t[1]), int(t[4]))
            seen = 1 if g.has_edge(a, b) else 0
            ma = (_bracket(t[6]), _bracket(t[7]))
            mb = (_bracket(t[8]), _bracket(t[9]))
            ra = (_bracket(t[-4]), _bracket(t[-3]))
            rb = (_bracket(t[-2]), _bracket(t[-1]))
            for u, v, (x, y, xr, yr) in ((a, b, (ma, mb, ra, rb)), (mirror(b), mirror(a), (mb, ma, rb, ra))):
                g.add_edge(u, v, hinge_edge=int(t[5]), intersection=seen, length=int(t[2]), z=0,
                           read_a_match_start=x[0], read_a_match_end=x[1], read_b_match_start=y[0], read_b_match_end=y[1],
                           read_a_match_start_raw=xr[0], read_a_match_end_raw=xr[1], read_b_match_start_raw=yr[0], read_b_match_end_raw=yr[1])
    return g


def read_hinges(path: str) -> Tuple[Set[Node], Set[Node]]:
    """`.hinge.list` (hinging.cpp:1694-1704) -> (in-hinge vertices, out-hinge vertices), pruning_and_clipping.py:1402-1414:
    a hinge of type 1 makes read_0 an in-hinge and read_1 an out-hinge, type -1 the other way round."""
    ins: Set[Node] = set()
    outs: Set[Node] = set()
    with open(path) as f:
        for line in f:
            t = line.split()
            if len(t) < 3:
                continue
            r = int(t[0])
            if t[2] == "1":
                ins.add((r, 0)); outs.add((r, 1))
            elif t[2] == "-1":
                ins.add((r, 1)); outs.add((r, 0))
    return ins, outs


def annotate_hinges(g: StrandGraph, ins: Set[Node], outs: Set[Node]) -> None:
    """Vertex attribute `hinge`: 1 in-hinge, -1 out-hinge, else 0; an in-hinge wins (pruning_and_clipping.py:1040-1051)."""
    for v in g.attr:
        g.attr[v]["hinge"] = 1 if v in ins else -1 if v in outs else 0


def flag_bad_coverage(g: StrandGraph, path: str) -> int:
    """`<prefix>.cov.flag` (one read id per line): CFLAG on both strands' vertices (pruning_and_clipping.py:1056-1083)."""
    for v in g.attr:
        g.attr[v]["CFLAG"] = False
    n = 0
    with open(path) as f:
        for line in f:
            name = line.strip()
            try:
                r = int(name)
            except ValueError:
                continue           # (a name that is no read id names no vertex: the reference's string look-ups simply miss, :1072-1083)
            if ((r, 0) in g) != ((r, 1) in g):
                raise ValueError("%s is not symmetrically present in the graph input." % name)
            if (r, 0) in g:
                g.attr[(r, 0)]["CFLAG"] = g.attr[(r, 1)]["CFLAG"] = True
                n += 1
    return n


def mark_skipped(g: StrandGraph, path: str) -> None:
    """`.edges.skipped`: `skipped` = 1 on the edges of the graph it names, and on their mirror images (:1021-1034)."""
    with open(path) as f:
        for line in f:
            t = line.split()
            if len(t) < 5:
                continue
            a, b = (int(t[0]), int(t[3])), (int(t[1]), int(t[4]))
            if g.has_edge(a, b):
                g.out[a][b]["skipped"] = 1
                g.out[mirror(b)][mirror(a)]["skipped"] = 1


# ---- the three operations -------------------------------------------------------------------------------------------------------
def _chain(h: StrandGraph, first: Node, limit: int) -> Tuple[List[Node], Node]:
    """Follow the unbranched chain that starts at `first`: vertices with one way in and one way out, at most until the chain
    (counting what the caller already holds) has `limit` members.  Returns (the chain's vertices, the vertex it ends in front of)."""
    chain: List[Node] = []
    cur = first
    while h.in_degree(cur) == 1 and h.out_degree(cur) == 1 and len(chain) < limit:
        chain.append(cur)
        cur = h.successors(cur)[0]
    return chain, cur


def clip_dead_ends(g: StrandGraph, threshold: int, order: Optional[Iterable[Node]] = None) -> StrandGraph:
    """Remove short dead ends, both strands at once (pruning_and_clipping.py:197-262).  A vertex nothing points to starts a path
    along vertices with exactly one way in and out; if the path has at most `threshold` vertices and ends in front of a junction
    (something else also points there) or at a vertex with no way on, the path and its mirror image go.  The mirror images are
    the dead ENDS of the other strand, so sources are all that is looked at."""
    h = g.copy()
    starts = [v for v in (h.nodes() if order is None else order) if v in h and h.in_degree(v) == 0]
    for st in starts:
        if st not in h:
            continue
        path = [st]
        cur = st
        if h.out_degree(st) == 1:
            more, cur = _chain(h, h.successors(st)[0], threshold + 1)   # (the path is followed two vertices past the threshold at most)
            path += more
        if len(path) <= threshold and (h.in_degree(cur) > 1 or h.out_degree(cur) == 0):
            for v in path:
                h.remove_node(v)
                h.remove_node(mirror(v))
    return h


def clip_z_edges(g: StrandGraph, threshold: int, ins: Set[Node] = frozenset(), outs: Set[Node] = frozenset(),
                 order: Optional[Iterable[Node]] = None) -> Tuple[StrandGraph, StrandGraph]:
    """Remove short cross links between two well-supported paths (pruning_and_clipping.py:331-390): from a vertex with several
    ways out (and not an out-hinge), a branch that runs unbranched for fewer than `threshold` edges into a vertex that something
    else also points to (and that is not an in-hinge) is cut, with its mirror image, while the start still has another way out.
    Returns (the clipped graph, a copy of the input with `z` = 1 on what was cut)."""
    h = g.copy()
    marked = g.copy()
    starts = [v for v in (h.nodes() if order is None else order) if v in h and h.out_degree(v) > 1 and v not in outs]
    for st in starts:
        if st not in h:
            continue
        for sec in h.successors(st):
            if h.out_degree(st) == 1:
                break
            if sec not in h:      # (cut as the interior of an earlier branch of this vertex)
                continue
            path = [(st, sec)]
            cur = sec
            while h.in_degree(cur) == 1 and h.out_degree(cur) == 1:
                nxt = h.successors(cur)[0]
                path.append((cur, nxt))
                cur = nxt
                if len(path) > threshold + 1:
                    break
            if len(path) <= threshold and h.in_degree(cur) > 1 and h.out_degree(st) > 1 and cur not in ins:
                for u, v in path:
                    marked.out[u][v]["z"] = 1
                    marked.out[mirror(v)][mirror(u)]["z"] = 1
                    if h.remove_edge(u, v):
                        h.remove_edge(mirror(v), mirror(u))
                for _, v in path[:-1]:
                    marked.attr[v]["z"] = 1
                    marked.attr[mirror(v)]["z"] = 1
                    if h.remove_node(v):
                        h.remove_node(mirror(v))
    return h, marked


def burst_bubbles(h: StrandGraph, threshold: int, order: Optional[Iterable[Node]] = None) -> StrandGraph:
    """Remove one side of short bubbles, in place (pruning_and_clipping.py:561-622): a vertex with exactly two ways out whose two
    branches run unbranched (at most `threshold` edges each) into the same vertex loses its FIRST branch - edges and interior
    vertices, with their mirror images.  "First" is the first of the vertex's two successors in this graph's adjacency order
    (edge insertion order); in the reference it is whichever of the two names CPython 2 hashes first."""
    starts = [v for v in (h.nodes() if order is None else order) if v in h and h.out_degree(v) == 2]
    for st in starts:
        if st not in h or h.out_degree(st) < 2:
            continue
        branches = []
        for sec in h.successors(st)[:2]:
            path = [(st, sec)]
            cur = sec
            while h.in_degree(cur) == 1 and h.out_degree(cur) == 1:
                nxt = h.successors(cur)[0]
                path.append((cur, nxt))
                cur = nxt
                if len(path) > threshold + 1:
                    break
            branches.append((path, cur))
        (first, end0), (second, end1) = branches
        if len(first) <= threshold and len(second) <= threshold and end0 == end1:
            for u, v in first:
                h.remove_edge(u, v)
                h.remove_edge(mirror(v), mirror(u))
            for _, v in first[:-1]:
                h.remove_node(v)
                h.remove_node(mirror(v))
    return h


# ---- GraphML ------------------------------------------------------------------------------------------------------------------
_GRAPHML_TYPE = {bool: "boolean", int: "int", float: "double", str: "string"}


_COPIED = ("length", "read_a_match_start", "read_a_match_end", "read_b_match_start", "read_b_match_end", "read_a_match_start_raw",
           "read_a_match_end_raw", "read_b_match_start_raw", "read_b_match_end_raw")


def _dup(v: Node) -> Node:
    return tuple(v[:2]) + ("B",) + tuple(v[2:])


def resolve_repeat(g: StrandGraph, rep: List[Node], in_node: Node, out_node: Node) -> None:
    """pruning_and_clipping.py:625-700 (resolve_rep): the loop's way through the repeat gets its OWN copy of the repeat's vertices -
    in_node -> B rep[0] -> ... -> B rep[-1] -> out_node, with the nine coordinate attributes of the edges it replaces - and the two
    edges by which the loop entered and left the shared copy go; the same on the other strand."""
    def cp(u, v):
        return {k: g.out[u][v][k] for k in _COPIED if k in g.out[u][v]}
    a = cp(in_node, rep[0])
    g.add_edge(in_node, _dup(rep[0]), **a)
    g.remove_edge(in_node, rep[0])
    a = cp(rep[-1], out_node)
    g.add_edge(_dup(rep[-1]), out_node, **a)
    g.remove_edge(rep[-1], out_node)
    a = cp(mirror(rep[0]), mirror(in_node))
    g.add_edge(mirror(_dup(rep[0])), mirror(in_node), **a)
    g.remove_edge(mirror(rep[0]), mirror(in_node))
    a = cp(mirror(out_node), mirror(rep[-1]))
    g.add_edge(mirror(out_node), mirror(_dup(rep[-1])), **a)
    g.remove_edge(mirror(out_node), mirror(rep[-1]))
    for x, y in zip(rep, rep[1:]):
        g.add_edge(_dup(x), _dup(y), **cp(x, y))
        g.add_edge(mirror(_dup(y)), mirror(_dup(x)), **cp(mirror(y), mirror(x)))


def resolve_loops(g: StrandGraph, max_nodes: int = 500, flank: int = 50, max_plasmid_length: int = 500000,
                  order: Optional[Iterable[Node]] = None) -> List[List[Node]]:
    """pruning_and_clipping.py:705-839 (loop_resolution), IN PLACE.  From a vertex with two ways out: one way is followed along
    unbranched vertices (at most max_nodes) to the first vertex that is not - the repeat's first vertex; if another path enters
    there it must come out of at least `flank` unbranched vertices, and so must the start's other way out go on; the repeat is
    then followed (a vertex with two ways in and one out first, then unbranched vertices) and, if it ends at the START vertex after
    more than max_plasmid_length bases in all - a loop through a collapsed repeat that is too long to be a plasmid -, the loop gets
    its own copy of the repeat (resolve_repeat).  Lengths are |a_start of the next edge - b_start of the previous one| summed along
    the way (the reference's measure, kept as it is - including that, inside the repeat, `prev_edge` stays the repeat's first edge).
    Returns the resolved repeats of fewer than five vertices (what the reference lists in tandem.txt).  `order` = the starting
    vertices' order (default: the graph's insertion order; the reference walks a hash order - parity unpinned, module docstring)."""
    tandem: List[List[Node]] = []
    starts = [v for v in (g.nodes() if order is None else order) if v in g and g.out_degree(v) == 2]
    # `in_node` lives as long as the reference's variable does (function scope, pruning_and_clipping.py:745): it is only assigned
    # inside the unbranched walk, so when a way out of the start is branched right away the predecessor filter below and
    # resolve_repeat see the value the PREVIOUS walk left - and the reference dies with a NameError when there was none.
    in_node: Optional[Node] = None
    for st in starts:
        if st not in g or g.out_degree(st) != 2:
            continue
        for first in g.successors(st):
            if g.out_degree(st) != 2 or not g.has_edge(st, first):
                continue
            other = [x for x in g.successors(st) if x != first][0]
            nxt = first
            prev_edge = g.out[st][nxt]
            loop_len = cnt = 0
            while g.in_degree(nxt) == 1 and g.out_degree(nxt) == 1 and cnt < max_nodes:
                cnt += 1
                in_node = nxt
                nxt = g.successors(nxt)[0]
                loop_len += abs(g.out[in_node][nxt]["read_a_match_start"] - prev_edge["read_b_match_start"])
                prev_edge = g.out[in_node][nxt]
            if cnt >= max_nodes:
                continue
            first_of_repeat = nxt
            if g.in_degree(nxt) == 2:
                if in_node is None:
                    raise NameError("loop resolution: the first way out of the first start vertex is branched (the reference's "
                                    "in_node is unbound here, pruning_and_clipping.py:762)")
                prev = [x for x in g.predecessors(nxt) if x != in_node][0]
                cnt = 0
                while g.in_degree(prev) == 1 and g.out_degree(prev) == 1:
                    cnt += 1
                    prev = g.predecessors(prev)[0]
                    if cnt >= flank:
                        break
                if cnt < flank:
                    continue
            nxt, cnt = other, 0
            while g.in_degree(nxt) == 1 and g.out_degree(nxt) == 1:
                cnt += 1
                nxt = g.successors(nxt)[0]
                if cnt >= flank:
                    break
            if cnt < flank:
                continue
            rep = [first_of_repeat]
            nxt, cnt = first_of_repeat, 0
            if g.in_degree(nxt) == 2 and g.out_degree(nxt) == 1:
                dbl = g.successors(nxt)[0]
                rep.append(dbl)
                prev_edge = g.out[nxt][dbl]
            else:
                dbl = nxt
                if g.in_degree(dbl) == 1 and g.out_degree(dbl) == 1:
                    raise AssertionError("loop resolution: the walk stopped at an unbranched vertex (%s)" % node_name(dbl))
            while g.in_degree(dbl) == 1 and g.out_degree(dbl) == 1 and cnt < max_nodes:
                cnt += 1
                succ = g.successors(dbl)[0]
                loop_len += abs(g.out[dbl][succ]["read_a_match_start"] - prev_edge["read_b_match_start"])
                dbl = succ
                rep.append(dbl)
            if dbl == st and loop_len > max_plasmid_length:
                if in_node is None:
                    raise NameError("loop resolution: in_node is unbound (pruning_and_clipping.py:829)")
                resolve_repeat(g, rep, in_node, other)
                if cnt < 5:
                    tandem.append(rep)
    return tandem


def prune_ys(g: StrandGraph, flank: int = 10, order: Optional[Iterable[Node]] = None) -> StrandGraph:
    """pruning_and_clipping.py:841-893 (y_pruning): at a vertex with one way in and several out that is reached over at least `flank`
    unbranched vertices - a Y, not a collapsed repeat - the branches into vertices flagged for bad coverage (CFLAG) are cut, with
    their mirror images.  Returns a copy."""
    h = g.copy()
    ys = [v for v in (h.nodes() if order is None else order) if v in h and h.out_degree(v) > 1 and h.in_degree(v) == 1]
    for st in ys:
        if h.in_degree(st) < 1:
            continue
        prev, cnt = h.predecessors(st)[0], 0
        while h.in_degree(prev) == 1 and h.out_degree(prev) == 1:
            cnt += 1
            prev = h.predecessors(prev)[0]
            if cnt >= flank:
                break
        if cnt < flank:
            continue
        for v in h.successors(st):
            if h.attr[v].get("CFLAG") is True:
                if h.remove_edge(st, v):
                    h.remove_edge(mirror(v), mirror(st))
    return h


def sparsify(g: StrandGraph, max_nodes: int = 1000, rng=None, max_draws: int = 20000) -> StrandGraph:
    """A copy of g thinned to at most max_nodes vertices for drawing (the reference's `random_condensation_sym`,
    pruning_and_clipping.py:456-500): per draw one vertex chosen at random; if it has exactly one predecessor p and one successor q,
    p has no other way out and q no other way in, and p, v, q are three different vertices, v is spliced out - p -> q gets
    `hinge_edge` -1, `z` 0 and `intersection` 1 only if p -> v and v -> q both had it - and so is its mirror image on the other
    strand where that path exists.  Stops at max_nodes vertices or after max_draws draws (the reference then says so and goes on).
    A graph that is small enough comes back untouched: no draw is made."""
    import random
    h = g.copy()
    rng = rng or random.Random(0)
    draws = 0

    def splice(p: Node, v: Node, q: Node) -> bool:
        if not (h.has_edge(p, v) and h.has_edge(v, q)):
            return False
        both = h.out[p][v].get("intersection") == 1 and h.out[v][q].get("intersection") == 1
        h.add_edge(p, q, hinge_edge=-1, intersection=1 if both else 0, z=0)
        h.remove_node(v)
        return True

    while len(h) > max_nodes and draws < max_draws:
        draws += 1
        verts = h.nodes()
        v = verts[rng.randrange(len(verts))]
        if h.in_degree(v) != 1 or h.out_degree(v) != 1:
            continue
        p, q = h.predecessors(v)[0], h.successors(v)[0]
        if h.out_degree(p) != 1 or h.in_degree(q) != 1 or len({p, v, q}) != 3:
            continue
        if splice(p, v, q):
            splice(mirror(q), mirror(v), mirror(p))      # (missing on a graph that is not strand-symmetric there: left as it is)
    if draws >= max_draws:
        print("[clip] Sparsification ended with %d nodes." % len(h))
    return h


def overlay_strands(g: StrandGraph) -> StrandGraph:
    """g with every vertex joined to its mirror image, both ways (the reference's `connect_strands`, :1108-1116, which changes its
    argument; here a copy): the two strands of a contig then lie next to each other in a layout."""
    h = g.copy()
    for v in g.nodes():
        h.add_edge(v, mirror(v))
        h.add_edge(mirror(v), v)
    return h


def _clip_rng():
    import random
    seed = os.environ.get("HINGE_CLIP_SEED", "0")
    return random.Random() if seed == "random" else random.Random(int(seed))


def write_graphml(g: StrandGraph, path: str) -> None:
    """GraphML as graph libraries read it (one <key> per attribute name and domain, typed; vertices named `read_strand`).
    Vertices, edges and keys are written in this graph's own order - the reference's file has the same content in its graph
    library's hash order."""
    keys: Dict[Tuple[str, str], Tuple[str, str]] = {}

    def key_of(domain: str, name: str, value) -> str:
        k = keys.get((domain, name))
        if k is None:
            k = ("d%d" % len(keys), _GRAPHML_TYPE[type(value)])
            keys[(domain, name)] = k
        return k[0]

    body: List[str] = []
    for v, a in g.attr.items():
        body.append("    <node id=%s>" % quoteattr(node_name(v)) if a else "    <node id=%s />" % quoteattr(node_name(v)))
        for name, value in a.items():
            body.append("      <data key=\"%s\">%s</data>" % (key_of("node", name, value), escape(_text(value))))
        if a:
            body.append("    </node>")
    for u, v, a in g.edges():
        body.append("    <edge source=%s target=%s>" % (quoteattr(node_name(u)), quoteattr(node_name(v))))
        for name, value in a.items():
            body.append("      <data key=\"%s\">%s</data>" % (key_of("edge", name, value), escape(_text(value))))
        body.append("    </edge>")
    with open(path, "w") as f:
        f.write("<?xml version='1.0' encoding='utf-8'?>\n")
        f.write("<graphml xmlns=\"http://graphml.graphdrawing.org/xmlns\" xmlns:xsi=\"http://www.w3.org/2001/XMLSchema-instance\" "
                "xsi:schemaLocation=\"http://graphml.graphdrawing.org/xmlns http://graphml.graphdrawing.org/xmlns/1.0/graphml.xsd\">\n")
        for (domain, name), (kid, typ) in keys.items():
            f.write("  <key id=\"%s\" for=\"%s\" attr.name=%s attr.type=\"%s\" />\n" % (kid, domain, quoteattr(name), typ))
        f.write("  <graph edgedefault=\"directed\">\n")
        f.write("\n".join(body))
        f.write("\n  </graph>\n</graphml>\n")


def _text(value) -> str:
    return str(value)      # (booleans as "True" / "False": what the reference's graph library writes, and reads back)


# ---- the command ----------------------------------------------------------------------------------------------------------------
def layout_prefix(edges_path: str) -> str:
    """Everything before the FIRST dot of the path as given (pruning_and_clipping.py:1246: directories with dots included)."""
    return edges_path.split(".")[0]


def clip_settings(ini_path: Optional[str]) -> dict:
    """[layout] del_telomeres / aggressive_pruning / max_plasmid_length (pruning_and_clipping.py:1256-1277; only the first
    changes anything in the stages built here)."""
    out = {"del_telomeres": False, "aggressive_pruning": False, "max_plasmid_length": 500000}
    if ini_path:
        cp = configparser.ConfigParser()
        cp.read(ini_path)
        for name in ("del_telomeres", "aggressive_pruning"):
            try:
                out[name] = cp.getint("layout", name) == 1
            except Exception:
                pass
        try:
            out["max_plasmid_length"] = cp.getint("layout", "max_plasmid_length")
        except Exception:
            pass
    return out


def build_graph(edges_path: str, hinges_path: str) -> Tuple[StrandGraph, Set[Node], Set[Node]]:
    g = read_edges(edges_path)
    ins, outs = read_hinges(hinges_path)
    annotate_hinges(g, ins, outs)
    prefix = layout_prefix(edges_path)
    if os.path.isfile(prefix + ".cov.flag"):
        flag_bad_coverage(g, prefix + ".cov.flag")
    if os.path.isfile(prefix + ".edges.skipped"):
        mark_skipped(g, prefix + ".edges.skipped")
    return g, ins, outs


def clip(g: StrandGraph, del_telomeres: bool = False) -> Tuple[StrandGraph, StrandGraph]:
    """(G0, G1) as the reference's main body makes them (pruning_and_clipping.py:1436-1471): dead ends of up to 10 vertices,
    Z edges of fewer than 6 (the hinge sets play no part: the script passes empty ones), bubbles of up to 10 edges a side,
    then dead ends of up to 5 once more (20 and 20 with del_telomeres).  G0 = the graph after the first clipping with what
    the Z clipping cut marked `z` = 1; G1 = the end result."""
    g0 = clip_dead_ends(g, 10)
    g1, g0 = clip_z_edges(g0, 6)
    g1 = burst_bubbles(g1, 20 if del_telomeres else 10)
    g1 = clip_dead_ends(g1, 20 if del_telomeres else 5)
    return g0, g1


def main(argv: Optional[List[str]] = None) -> int:
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) < 3:
        sys.stderr.write("usage: hinge clip <prefix>.edges.hinges <prefix>.hinge.list <suffix> [nominal.ini]\n")
        return 1
    edges_path, hinges_path, suffix = argv[0], argv[1], argv[2]
    cfg = clip_settings(argv[3] if len(argv) >= 4 else None)
    if len(argv) >= 5:
        sys.stderr.write("[clip] ground-truth annotation (5th argument) is not part of this build; ignored\n")
    g, _, _ = build_graph(edges_path, hinges_path)
    print("[clip] Graph with %d nodes built" % len(g))
    g0, g1 = clip(g, cfg["del_telomeres"])
    print("[clip] Number of nodes remaining: %d" % len(g1))
    out = layout_prefix(edges_path) + suffix
    write_graphml(g0, out + ".G0.graphml")
    write_graphml(g1, out + ".G1.graphml")
    # G2 = G1 after loop resolution (pruning_and_clipping.py:1488-1497); with [layout] aggressive_pruning = 1 also G3 = G2 after Y
    # pruning and one more dead-end clipping (:1521-1527).  Everything else the script writes goes through random_condensation_sym.
    g2 = g1.copy()
    tandem = resolve_loops(g2, 500, 50, cfg["max_plasmid_length"])
    if tandem:
        with open("tandem.txt", "w") as f:
            for rep in tandem:
                f.write(str([node_name(v) for v in rep]))
    write_graphml(g2, out + ".G2.graphml")
    written = "G0, G1, G2"
    # the sparsified graphs and their strand overlays (pruning_and_clipping.py:1486, 1500-1515): Gs from G1, G2s from G2, in the
    # reference's order of draws (Gs first)
    rng = _clip_rng()
    gs = sparsify(g1, 1000, rng)
    g2s = sparsify(g2, 1000, rng)
    write_graphml(gs, out + ".Gs.graphml")
    write_graphml(g2s, out + ".G2s.graphml")
    write_graphml(overlay_strands(gs), out + ".Gc.graphml")
    write_graphml(overlay_strands(g2s), out + ".G2c.graphml")
    written += ", Gs, G2s, Gc, G2c"
    if cfg["aggressive_pruning"]:
        g3 = clip_dead_ends(prune_ys(g2, 10), 10)
        g3s = sparsify(g3, 1000, rng)
        write_graphml(g3, out + ".G3.graphml")
        write_graphml(g3s, out + ".G3s.graphml")
        write_graphml(overlay_strands(g3s), out + ".G3c.graphml")
        written += ", G3, G3s, G3c"
    if os.path.exists(out + ".PARTIAL"):      # (left by a run of an earlier build)
        os.remove(out + ".PARTIAL")
    if max(len(g1), len(g2)) > 1000:
        sys.stderr.write("[clip] more than 1000 vertices: the sparsified graphs are drawn at random - as the reference's are, which seeds nothing; "
                         "HINGE_CLIP_SEED=%s here\n" % os.environ.get("HINGE_CLIP_SEED", "0"))
    print("[clip] Done (%s)" % written)
    return 0

if __name__ == "__main__":
    sys.exit(main())
