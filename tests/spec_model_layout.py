"""A SECOND reading of `hinge layout`'s hinge bookkeeping (hinging.cpp): GetAlignment (:398-412 activity, :478-602 the
active x active pairs), the weight sort (:1066-1071), hinges / killed hinges (:1170-1208), the hinges a bridging match removes
(:1262-1321), the hinge graph with its matched and killed partners (:1365-1640), the connected-component filter (:1644-1675) and
`.hinge.list` (:1694-1704) - written from the reference's source, not from oracle/hinge_oracle.cpp and not from the product's
host code, followed by the greedy selection (:1911-2148) and PrintOverlapToFile (:188-248).  TEST INFRASTRUCTURE; it pins nothing
(the stage programs cannot be built in this image), it gives `.garbage.txt`, `.killed.hinges`, `.hgraph`, `.hinge.list`,
`.edges.hinges`, `.edges.skipped` and `.deadends.txt` a second, independently written reading (tests/test_spec_model_layout.py).

Pinned primitives used as they are (tests/test_oracle_pinned.py): ProcessAlignment, GetMatchingPosition, std::sort with
compare_overlap / compare_overlap_weight, the iteration order of libstdc++'s unordered_map<int, ...>.

Things in the source that are easy to read past (each changes a file):
  * `.repeat.txt` / `.hinges.txt` pairs are kept only if BOTH numbers are non-zero (`ss >> r1 >> r2`, :896-903);
  * a Hinge is (pos, type, active); `.killed.hinges` prints "type pos" (:1203-1205), `.hinge.list` prints the marked hinge;
  * the pair vectors are sorted ONCE here (twice in maximal), and `use_two_matches` comes from [layout];
  * a read found contained ("Should not happen", :590-600) goes inactive for everything after it;
  * hinge removal by a bridging match looks at ACTIVE matches of the right kind whose B read is active; FORWARD needs
    eff_start < pos - kill_hinge_overlap, FORWARD_INTERNAL eff_start < pos + kill_hinge_internal, and only in-hinges (+1) go;
    BACKWARD / BACKWARD_INTERNAL mirror it with eff_end and out-hinges (-1);
  * in the graph loop the hinge's OWN activity is never tested; a reverse-complement match flips the type it looks for on B;
    an edge is written "i b posI posB 1 rev" when the hinge is an in-hinge on a forward match (out-hinge on a backward match),
    else with the two reads swapped; matches against B's KILLED hinges write the same line with 0;
  * the forward block collects new_killed_hinges (a hinge of read i that meets a KILLED hinge of B through a plain FORWARD match)
    inside the type test, the backward block OUTSIDE it (position test only, :1617-1627): they poison plain FORWARD / BACKWARD
    matches of read i in the selection (:1931-1955, :2050-2072), one `.edges.skipped` line per poisoning hinge;
  * selection (:1911-2148): matches in weight order, only active ones with an active B; the FIRST unpoisoned FORWARD match is
    taken; a FORWARD_INTERNAL match is looked at only while none was taken before (forward_internal == 0) and B has hinges: the
    first hinge of B inside HINGE_TOLERANCE of the match's raw B start (raw B END for a reverse-complement match) with type
    1 - 2 comp that is active decides - taken if nothing is chosen yet or its weight > chosen weight - 2 HINGE_SLACK - and the
    search stops at that hinge either way; BACKWARD mirrors it (raw B end / start, type -1 + 2 comp);
  * PrintOverlapToFile (:188-248) swaps reads and the first four bracket pairs for BACKWARD types, not the raw pairs; the two
    direction columns are (0, comp) forward and (comp, 0) backward;
  * components are counted over ALL hinges of ALL reads (a hinge without an edge is a component of one), and a small component
    deactivates its hinges whatever their read's state.
"""
FORWARD, BACKWARD, BCOVERA, FORWARD_INTERNAL, BACKWARD_INTERNAL = 0, 1, 3, 12, 13


def parse_pairs(path, n_read):
    out = [[] for _ in range(n_read)]
    for line in open(path):
        tok = line.split()
        if not tok:
            continue
        num, rest = int(tok[0]), [int(t) for t in tok[1:]]
        out[num] = []
        for k in range(0, len(rest), 2):
            r1 = rest[k]
            r2 = rest[k + 1] if k + 1 < len(rest) else 0
            if r1 != 0 and r2 != 0:
                out[num].append((r1, r2))
    return out


def layout_hinges(n_read, eff, maximal, repeats, hinges, parts, P, process_alignment, matching_position, sort_perm, umap_order):
    """P: dict of the [filter] / [layout] values.  parts as in spec_model_maximal.  Returns dict of file name suffix -> lines."""
    active = [True] * n_read
    garbage = []
    for i in range(n_read):                                                   # :954-960
        if eff[i][1] - eff[i][0] < P["length_threshold"]:
            active[i] = False
            garbage.append("%d" % i)
    for i in range(n_read):                                                   # :409-411
        active[i] = active[i] and maximal[i]
    fwd = [[] for _ in range(n_read)]
    bwd = [[] for _ in range(n_read)]
    import numpy as np
    for p in parts:
        a, b = p["aread"], p["bread"]
        r_begin, r_end = int(a[0]), int(a[-1])
        first = np.searchsorted(a, np.arange(n_read + 1), side="left")
        keep = [active[int(a[j])] and active[int(b[j])] for j in range(len(a))]      # decided once per part, before the read loop
        for i in range(r_begin, r_end + 1):
            if not active[i]:
                continue
            recs = [j for j in range(int(first[i]), int(first[i + 1])) if keep[j]]
            groups = {}
            for j in recs:
                groups.setdefault(int(b[j]), []).append(j)
            contained = False
            for bid in umap_order([int(b[j]) for j in recs]):
                v = groups[bid]
                lens = [int(p["ae"][j]) - int(p["ab"][j]) + int(p["be"][j]) - int(p["bb"][j]) for j in v]
                v = [v[k] for k in sort_perm(lens)]
                for j in v[:2 if P["use_two_matches"] else 1]:
                    if int(a[j]) == int(b[j]):
                        continue                                              # inactive alignment: NOT_ACTIVE, in no list
                    m = process_alignment(int(p["ab"][j]), int(p["ae"][j]), int(p["bb"][j]), int(p["be"][j]), int(p["comp"][j]),
                                          eff[i], eff[bid], p["trace"][j], P["aln_threshold"], P["theta"], P["theta2"], True)
                    m.update(b=bid, comp=int(p["comp"][j]), raw=(int(p["ab"][j]), int(p["ae"][j]), int(p["bb"][j]), int(p["be"][j])), trace=p["trace"][j])
                    if m["active"] and m["type"] == BCOVERA and active[bid]:
                        contained = True
                    if m["type"] in (FORWARD, FORWARD_INTERNAL):
                        fwd[i].append(m)
                    elif m["type"] in (BACKWARD, BACKWARD_INTERNAL):
                        bwd[i].append(m)
            if contained:
                active[i] = False
    for i in range(n_read):                                                   # :1066-1071
        if active[i]:
            fwd[i] = [fwd[i][k] for k in sort_perm([m["weight"] for m in fwd[i]])]
            bwd[i] = [bwd[i][k] for k in sort_perm([m["weight"] for m in bwd[i]])]

    H = [[{"pos": pos, "type": typ, "active": True} for (pos, typ) in hinges[i]] for i in range(n_read)]     # :1180-1186
    K = []
    for i in range(n_read):                                                   # :1187-1195
        surviving = set(hinges[i])
        K.append([{"pos": pos, "type": typ} for (pos, typ) in repeats[i] if (pos, typ) not in surviving])
    killed_lines = ["%d " % i + "".join("%d %d " % (h["type"], h["pos"]) for h in K[i]) for i in range(n_read)]

    ko, ki = P["kill_hinge_overlap"], P["kill_hinge_internal"]
    for i in range(n_read):                                                   # :1262-1321
        if not active[i]:
            continue
        for m in fwd[i]:
            if m["active"] and m["type"] in (FORWARD, FORWARD_INTERNAL) and active[m["b"]]:
                for h in H[i]:
                    if h["type"] == 1 and ((m["type"] == FORWARD_INTERNAL and m["eff_ab"] < h["pos"] + ki) or
                                           (m["type"] == FORWARD and m["eff_ab"] < h["pos"] - ko)):
                        h["active"] = False
        for m in bwd[i]:
            if m["active"] and m["type"] in (BACKWARD, BACKWARD_INTERNAL) and active[m["b"]]:
                for h in H[i]:
                    if h["type"] == -1 and ((m["type"] == BACKWARD_INTERNAL and m["eff_ae"] > h["pos"] - ki) or
                                            (m["type"] == BACKWARD and m["eff_ae"] > h["pos"] + ko)):
                        h["active"] = False

    node = {}
    for i in range(n_read):
        for k in range(len(H[i])):
            node[(i, k)] = len(node)
    parent = list(range(len(node)))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    hgraph = []
    new_killed = [[] for _ in range(n_read)]
    slack = P["matching_hinge_slack"]
    for i in range(n_read):                                                   # :1365-1640
        if not active[i]:
            continue
        for k, h in enumerate(H[i]):
            for matches, kinds, straight in ((fwd[i], (FORWARD, FORWARD_INTERNAL), 1), (bwd[i], (BACKWARD, BACKWARD_INTERNAL), -1)):
                for m in matches:
                    if not (m["active"] and m["type"] in kinds and active[m["b"]]):
                        continue
                    pos_b = matching_position(m["raw"], m["comp"], m["trace"], h["pos"])
                    want = -h["type"] if m["comp"] else h["type"]
                    rev = 1 if m["comp"] else 0
                    bid = m["b"]
                    for l, hb in enumerate(H[bid]):
                        if pos_b - slack < hb["pos"] < pos_b + slack and want == hb["type"]:
                            parent[find(node[(i, k)])] = find(node[(bid, l)])
                            if h["type"] == straight:
                                hgraph.append("%d %d %d %d %d %d" % (i, bid, h["pos"], hb["pos"], 1, rev))
                            else:
                                hgraph.append("%d %d %d %d %d %d" % (bid, i, hb["pos"], h["pos"], 1, rev))
                    for hb in K[bid]:
                        if pos_b - slack < hb["pos"] < pos_b + slack:
                            if want == hb["type"]:
                                if h["type"] == straight:
                                    hgraph.append("%d %d %d %d %d %d" % (i, bid, h["pos"], hb["pos"], 0, rev))
                                else:
                                    hgraph.append("%d %d %d %d %d %d" % (bid, i, hb["pos"], h["pos"], 0, rev))
                            if (straight == 1 and want == hb["type"] and m["type"] == FORWARD) or (straight == -1 and m["type"] == BACKWARD):
                                new_killed[i].append((h["pos"], h["type"]))
    size = {}
    for x in range(len(node)):                                                # :1644-1675
        r = find(x)
        size[r] = size.get(r, 0) + 1
    for (i, k), x in node.items():
        if size[find(x)] < P["min_connected_component_size"]:
            H[i][k]["active"] = False
    hinge_list = ["%d %d %d" % (i, hinges[i][k][0], hinges[i][k][1]) for i in range(n_read) for k in range(len(H[i])) if active[i] and H[i][k]["active"]]
    # ---- selection and printing (:1911-2148, :188-248) ----
    def line(i, m):
        hinged = -1 if m["type"] in (FORWARD, BACKWARD) else 1          # UNHINGED_EDGE / HINGED_EDGE (:33-34)
        ra = (m["raw"][0], m["raw"][1])
        rb = (m["raw"][2], m["raw"][3])
        if m["type"] in (FORWARD, FORWARD_INTERNAL):
            return "%d %d %d %d %d %d [%d %d] [%d %d] [%d %d] [%d %d] [%d %d] [%d %d]" % (
                i, m["b"], m["length"], 0, m["comp"], hinged, m["eff_ab"], m["eff_ae"], m["eff_bb"], m["eff_be"],
                eff[i][0], eff[i][1], eff[m["b"]][0], eff[m["b"]][1], ra[0], ra[1], rb[0], rb[1])
        return "%d %d %d %d %d %d [%d %d] [%d %d] [%d %d] [%d %d] [%d %d] [%d %d]" % (
            m["b"], i, m["length"], m["comp"], 0, hinged, m["eff_bb"], m["eff_be"], m["eff_ab"], m["eff_ae"],
            eff[m["b"]][0], eff[m["b"]][1], eff[i][0], eff[i][1], ra[0], ra[1], rb[0], rb[1])

    tol, hslack = P["hinge_tolerance"], P["hinge_slack"]
    edges, skipped, deadends = [], [], []
    for i in range(n_read):
        if not active[i]:
            continue
        for matches, plain, internal, name in ((fwd[i], FORWARD, FORWARD_INTERNAL, "forward"), (bwd[i], BACKWARD, BACKWARD_INTERNAL, "backward")):
            chosen, taken, taken_internal = None, False, False
            for m in matches:
                if not (m["active"] and active[m["b"]]):
                    continue
                if m["type"] == plain and not taken:
                    poisoned = False
                    for (pos, typ) in new_killed[i]:
                        if plain == FORWARD:
                            hit = (m["comp"] != 1 and typ == -1 and pos > m["eff_be"]) or (m["comp"] == 1 and typ == 1 and pos < m["eff_bb"])
                        else:
                            hit = (m["comp"] != 1 and typ == 1 and pos < m["eff_bb"]) or (m["comp"] == 1 and typ == -1 and pos > m["eff_be"])
                        if hit:
                            skipped.append(line(i, m))
                            poisoned = True
                    if not poisoned:
                        chosen, taken = m, True
                elif m["type"] == internal and len(H[m["b"]]) > 0 and not taken_internal:
                    if plain == FORWARD:
                        anchor = m["raw"][3] if m["comp"] == 1 else m["raw"][2]
                        want = 1 - 2 * m["comp"]
                    else:
                        anchor = m["raw"][2] if m["comp"] == 1 else m["raw"][3]
                        want = -1 + 2 * m["comp"]
                    for hb in H[m["b"]]:
                        if hb["pos"] - tol < anchor < hb["pos"] + tol and hb["type"] == want and hb["active"]:
                            if not taken or m["weight"] > chosen["weight"] - 2 * hslack:
                                chosen, taken, taken_internal = m, True, True
                            break
            if chosen is not None:
                edges.append(line(i, chosen))
            else:
                deadends.append("%d\t matches_%s size: %d" % (i, name, len(matches)))
    return {".garbage.txt": garbage, ".killed.hinges": killed_lines, ".hgraph": hgraph, ".hinge.list": hinge_list,
            ".edges.hinges": edges, ".edges.skipped": skipped, ".deadends.txt": deadends}
