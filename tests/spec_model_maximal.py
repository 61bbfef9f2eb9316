"""A SECOND reading of `hinge maximal`'s main body (maximal.cpp:524-547 activity from the masks, :615-654 the (A, B) grouping,
:780-858 the order-dependent containment loop, :873-878 `.max`), written from the reference's source next to SURVEY 3.2 - not
from oracle/hinge_oracle.cpp and not from the product's host code.  TEST INFRASTRUCTURE.  tests/test_spec_model_maximal.py diffs
its `.max` / `.contained.txt` against the oracle's: the stage programs cannot be built in this image (spdlog / Boost.Graph), so
this does not pin anything - it gives the files a second, independently written reading.

What is NOT re-derived here, because it IS pinned to the reference's compiled code (tests/test_oracle_pinned.py): ProcessAlignment
(trim_overlap + AddTypesAsymmetric) through `process_alignment`, libstdc++'s std::sort with compare_overlap through `sort_perm`,
and the iteration order of libstdc++'s unordered_map<int, ...> through `umap_order`.

Points of the source that are easy to get wrong (all visible in the outputs):
  * the pair vectors are sorted TWICE (once for every read at :640-654, again at :790 for the read under test);
  * `containing_read` is assigned whenever the alignment says "B covers A" - whether or not B is still active - while `contained`
    only counts active B reads: .contained.txt names the LAST covering B in hash-map order, active or not (:813-818, :836-841);
  * a read's own activity is tested once, before its pairs are walked; B's activity is whatever the loop has left it at
    (lower ids: final, higher ids: initial);
  * A == B records are inactive alignments (:602-604): they stay in the map (the iteration order depends on the key being there)
    and ProcessAlignment turns them into NOT_ACTIVE;
  * the second overlap of a pair is only looked at with [filter] use_two_matches (default true).
"""
import numpy as np

BCOVERA = 3


def maximal(rlen, eff, parts, length_threshold, aln_threshold, theta, theta2, use_two_matches, trim,
            process_alignment, sort_perm, umap_order):
    """parts: list of dicts with per-record arrays aread, bread, comp, ab, ae, bb, be (B on its forward strand), trace (list of
    uint16 arrays).  Returns (lines of .max, lines of .contained.txt)."""
    n = len(rlen)
    active = [(int(eff[i][1]) - int(eff[i][0])) >= length_threshold for i in range(n)]        # :541-547
    out_max, out_contained = [], []
    for p in parts:
        a, b = p["aread"], p["bread"]
        r_begin, r_end = int(a[0]), int(a[-1])
        first = np.searchsorted(a, np.arange(n + 1), side="left")                               # records are sorted by A
        for i in range(r_begin, r_end + 1):
            if not active[i]:
                continue
            contained, containing = False, None
            recs = range(int(first[i]), int(first[i + 1]))
            keys = [int(b[j]) for j in recs]                                                    # insertion sequence of idx_ab[i]
            groups = {}
            for j in recs:
                groups.setdefault(int(b[j]), []).append(j)
            for bid in umap_order(keys):
                v = groups[bid]
                for _ in range(2):                                                              # :640-654 and :790
                    lens = [int(p["ae"][j]) - int(p["ab"][j]) + int(p["be"][j]) - int(p["bb"][j]) for j in v]
                    v = [v[k] for k in sort_perm(lens)]
                for j in v[:2 if use_two_matches else 1]:
                    if int(a[j]) == int(b[j]):
                        covers = False                                                          # inactive alignment -> NOT_ACTIVE
                    else:
                        res = process_alignment(int(p["ab"][j]), int(p["ae"][j]), int(p["bb"][j]), int(p["be"][j]), int(p["comp"][j]),
                                                eff[i], eff[bid], p["trace"][j], aln_threshold, theta, theta2, trim)
                        covers = res["active"] and res["type"] == BCOVERA
                    if covers:
                        containing = bid
                    if active[bid]:
                        contained = contained or covers
            if contained:
                active[i] = False
                out_contained.append("%d\t%d" % (i, containing))
        out_max += ["%d" % i for i in range(r_begin, r_end + 1) if active[i]]
    return out_max, out_contained
