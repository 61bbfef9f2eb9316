"""`hinge draft-path <dir> <db name> <graphml>` - from the clipped read graph to the path file `hinge draft` stitches
contigs from: `<dir>/<name>.edges.list` and `<dir>/<name>_draft.graphml`.

Restated from the BEHAVIOUR of the reference's scripts/get_draft_path.py (all 447 lines are one script body):
  :63-112   the graph is read, every vertex gets cut_start = 0 / cut_end = its read's length; at a vertex several edges enter
            (leave) the cut moves to the furthest entering (earliest leaving) match position - taken, for a strand-1 vertex,
            from the strand-0 vertex of the same read, mirrored;
  :120-152  unbranched chains are merged into their first vertex (`path`, `weightspath`, the last vertex's cut_end);
  :175-444  one record per contig AND its reverse complement, consecutively numbered: `O` a single read, `D` two reads, else
            `S` (first edge + where the contig starts), `T` ..., `E` (last edge + where it ends), the cuts of a contig that
            continues into another one taken from that neighbour.
The reference is a Python 2 script over networkx 1.x that shells out to DBshow for the read lengths; this module has its own
graph type (hinge_amd.clip.StrandGraph), reads the lengths from the DB's index, and visits vertices in the order the GraphML file
lists them.  **PARITY UNPINNED, order-dependent**: the reference walks `out_graph.nodes()` in CPython 2's hash order, which
decides the contigs' NUMBERS and their order in the file (not their content: a contig and its reverse complement always
come as a pair, and which of the two comes first depends on which vertex is met first).  A vertex named 'B' + name (the copy loop
resolution makes) is the read `name`, as in the reference (`lstrip('B')`).

    python -m hinge_amd.draft_path <dir> <db name> <graph.graphml>
"""
from __future__ import annotations

import os
import sys
import xml.etree.ElementTree as ET
from typing import Dict, List, Optional

from . import formats
from .clip import Node, StrandGraph, mirror, node_name, write_graphml

_NS = "{http://graphml.graphdrawing.org/xmlns}"


def parse_node(name: str) -> Node:
    head, strand = name.split("_")
    tag = head[:len(head) - len(head.lstrip("B"))]
    v: Node = (int(head.lstrip("B")), int(strand))
    return v + ((tag,) if tag else ())


def read_graphml(path: str) -> StrandGraph:
    """A GraphML file as hinge_amd.clip.write_graphml (or networkx) writes it: typed <key>s, <data> under nodes and edges."""
    root = ET.parse(path).getroot()
    conv = {"int": int, "long": int, "float": float, "double": float, "boolean": lambda s: s.strip().lower() == "true", "string": str}
    keys = {k.get("id"): (k.get("attr.name"), conv.get(k.get("attr.type"), str)) for k in root.iter(_NS + "key")}
    g = StrandGraph()
    graph = root.find(_NS + "graph")

    def data_of(el) -> dict:
        out = {}
        for d in el.findall(_NS + "data"):
            name, fn = keys[d.get("key")]
            out[name] = fn(d.text if d.text is not None else "")
        return out

    for el in graph.findall(_NS + "node"):
        v = parse_node(el.get("id"))
        g.add_node(v)
        g.attr[v].update(data_of(el))
    for el in graph.findall(_NS + "edge"):
        g.add_edge(parse_node(el.get("source")), parse_node(el.get("target")), **data_of(el))
    return g


def _fields(v: Node) -> str:
    return "%d %d" % (v[0], v[1])


def draft_path(g: StrandGraph, rlen) -> "tuple[StrandGraph, List[str]]":
    """(the merged graph, the lines of .edges.list)."""
    h = g.copy()
    plain = lambda v: (v[0], 0)                                    # the strand-0 vertex of the same read, WITHOUT the 'B' tag (:100, :108)
    for v in h.nodes():
        vlen = int(rlen[v[0]])
        a = h.attr[v]
        a["cut_start"], a["cut_end"] = 0, vlen
        if h.in_degree(v) > 1:
            if v[1] == 0:
                a["cut_start"] = max(h.out[x][v]["read_b_match_start"] for x in h.predecessors(v))
            else:
                a["cut_start"] = vlen - min(h.out[plain(v)][x]["read_a_match_start"] for x in h.successors(plain(v)))
        if h.out_degree(v) > 1:
            if v[1] == 0:
                a["cut_end"] = min(h.out[v][x]["read_a_match_start"] for x in h.successors(v))
            else:
                a["cut_end"] = vlen - max(h.out[x][plain(v)]["read_b_match_start"] for x in h.predecessors(plain(v)))
    # ---- unbranched chains into their first vertex (:120-152) -----------------------------------------------------------------
    todo = [v for v in h.nodes() if h.in_degree(v) == 1 and h.out_degree(h.predecessors(v)[0]) == 1]
    for cur in todo:
        prev = h.predecessors(cur)[0]
        if prev != cur:
            w = str(h.out[prev][cur]["length"])
            pa, ca = h.attr[prev], h.attr[cur]
            path1, wp1 = (pa["path"], pa["weightspath"]) if "path" in pa else ([prev], [])
            path2, wp2 = (ca["path"], ca["weightspath"]) if "path" in ca else ([cur], [])
            pa["path"], pa["weightspath"] = path1 + path2, wp1 + [w] + wp2
            for nb in h.successors(cur):
                h.add_edge(prev, nb, length=h.out[cur][nb]["length"])
            pa["cut_end"] = ca["cut_end"]
            h.remove_node(cur)
        else:                                                       # a cycle closed onto its own head
            a = h.attr[cur]
            if "path" not in a:
                raise KeyError("path")                              # (the reference's KeyError: a read that only overlaps itself)
            a["path"] = a["path"] + [cur]
            a["weightspath"] = a["weightspath"] + [str(h.out[prev][cur]["length"])]
            a["cut_end"] = int(rlen[cur[0]])
    # ---- the records ------------------------------------------------------------------------------------------------------------
    lines: List[str] = []
    printed: Dict[Node, int] = {}
    contig_no = 0
    L = lambda v: int(rlen[v[0]])
    for v in h.nodes():
        a = h.attr[v]
        if mirror(v) in printed:
            a["contig_id"] = printed[mirror(v)] + 1
            continue
        if "path" not in a:                                         # one read
            a["contig_id"] = contig_no + 1
            lines.append(">Unitig%d" % contig_no)
            printed[v] = contig_no
            contig_no += 1
            lines.append("O %s %s %d %d" % (_fields(v), _fields(v), a["cut_start"], a["cut_end"]))
            lines.append(">Unitig%d" % contig_no)
            contig_no += 1
            m = mirror(v)
            lines.append("O %s %s %d %d" % (_fields(m), _fields(m), h.attr[m]["cut_start"], h.attr[m]["cut_end"]))
            continue
        nodes, weights = a["path"], a["weightspath"]
        if h.in_degree(v) != 1 and h.out_degree(v) != 1 and len(nodes) == 2:     # two reads
            a["contig_id"] = contig_no
            lines.append(">Unitig%d" % contig_no)
            printed[nodes[0]] = printed[nodes[1]] = contig_no
            contig_no += 1
            lines.append("D %s %s %s %d %d" % (_fields(nodes[0]), _fields(nodes[1]), weights[0], a["cut_start"], a["cut_end"]))
            lines.append(">Unitig%d" % contig_no)
            contig_no += 1
            ra, rb = mirror(nodes[1]), mirror(nodes[0])
            lines.append("D %s %s %s %d %d" % (_fields(ra), _fields(rb), weights[0], L(ra) - a["cut_end"], L(rb) - a["cut_start"]))
            continue
        if len(nodes) != len(weights) + 1:
            print("Something went wrong with contig " + str(contig_no))
            continue
        for x in nodes:
            printed[x] = contig_no
        a["contig_id"] = contig_no
        lines.append(">Unitig%d" % contig_no)
        contig_no += 1
        nw = len(weights)
        from_other = h.in_degree(v) == 1 and h.predecessors(v)[0] != v
        into_other = h.out_degree(v) == 1 and h.successors(v)[0] != v
        last_of = lambda c: h.attr[c]["path"][-1] if "path" in h.attr[c] else c
        first_of = lambda c: h.attr[c]["path"][0] if "path" in h.attr[c] else c
        if from_other:                                              # the contig starts where the one in front of it ended
            pc = h.predecessors(v)[0]
            lines.append("S %s %s %s %d" % (_fields(last_of(pc)), _fields(nodes[0]), h.out[pc][v]["length"], h.attr[pc]["cut_end"]))
            if len(nodes) > 2:
                lines.append("T %s %s %s" % (_fields(nodes[0]), _fields(nodes[1]), weights[0]))
        else:
            lines.append("S %s %s %s %d" % (_fields(nodes[0]), _fields(nodes[1]), weights[0], a["cut_start"]))
        for i in range(1, nw - 1):
            lines.append("T %s %s %s" % (_fields(nodes[i]), _fields(nodes[i + 1]), weights[i]))
        if into_other:
            if len(nodes) > 2:
                lines.append("T %s %s %s" % (_fields(nodes[nw - 1]), _fields(nodes[nw]), weights[-1]))
            nc = h.successors(v)[0]
            lines.append("E %s %s %s %d" % (_fields(nodes[nw]), _fields(first_of(nc)), h.out[v][nc]["length"], h.attr[nc]["cut_start"]))
        else:
            lines.append("E %s %s %s %d" % (_fields(nodes[nw - 1]), _fields(nodes[nw]), weights[-1], a["cut_end"]))
        # the reverse complement of the same contig, right behind it
        lines.append(">Unitig%d" % contig_no)
        contig_no += 1
        if into_other:
            nc = h.successors(v)[0]
            na, nb = mirror(first_of(nc)), mirror(nodes[nw])
            lines.append("S %s %s %s %d" % (_fields(na), _fields(nb), h.out[v][nc]["length"], L(na) - h.attr[nc]["cut_start"]))
            if len(nodes) > 2:
                lines.append("T %s %s %s" % (_fields(mirror(nodes[nw])), _fields(mirror(nodes[nw - 1])), weights[-1]))
        else:
            na, nb = mirror(nodes[nw]), mirror(nodes[nw - 1])
            lines.append("S %s %s %s %d" % (_fields(na), _fields(nb), weights[-1], L(na) - a["cut_end"]))
        for i in range(nw - 1, 1, -1):
            lines.append("T %s %s %s" % (_fields(mirror(nodes[i])), _fields(mirror(nodes[i - 1])), weights[i - 1]))
        if from_other:
            if len(nodes) > 2:
                lines.append("T %s %s %s" % (_fields(mirror(nodes[1])), _fields(mirror(nodes[0])), weights[0]))
            pc = h.predecessors(v)[0]
            na, nb = mirror(nodes[0]), mirror(last_of(pc))
            lines.append("E %s %s %s %d" % (_fields(na), _fields(nb), h.out[pc][v]["length"], L(nb) - h.attr[pc]["cut_end"]))
        else:
            nb, na = mirror(nodes[0]), mirror(nodes[1])
            lines.append("E %s %s %s %d" % (_fields(na), _fields(nb), weights[0], L(nb) - a["cut_start"]))
    print("Number of contigs: " + str(contig_no))
    return h, lines


def main(argv: Optional[List[str]] = None) -> int:
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) < 3:
        sys.stderr.write("usage: hinge draft-path <dir> <db name> <graph.graphml>\n")
        return 1
    filedir, filename, graphml = argv[0], argv[1], argv[2]
    g = read_graphml(graphml)
    rlen = formats.read_db_index(os.path.join(filedir, filename))["rlen"]
    h, lines = draft_path(g, rlen)
    with open(os.path.join(filedir, filename + ".edges.list"), "w") as f:
        for ln in lines:
            f.write(ln + "\n")
    for v in h.nodes():                                             # lists as the reference's ';'-joined strings
        a = h.attr[v]
        if "path" in a:
            a["path"] = ";".join(node_name(x) for x in a["path"])
            a["weightspath"] = ";".join(a["weightspath"])
    write_graphml(h, os.path.join(filedir, filename + "_draft.graphml"))
    return 0


if __name__ == "__main__":
    sys.exit(main())
