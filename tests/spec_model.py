"""A SECOND, independent model of `hinge filter` (coverage mask, repeat annotation, merge, gate, hinge calling), written only from
the prose specification in SURVEY.md Appendix A (A1-A8) - not from the oracle's source and not from the kernels.  TEST
INFRASTRUCTURE.  tests/test_spec_model.py diffs it against the oracle's output files: the three `main()` bodies of the reference
cannot be built in this image (spdlog / Boost.Graph are absent), so for those loops parity stays "unpinned"; what this gives is a
three-way agreement reference-reading #1 (oracle, statement by statement) == reference-reading #2 (this file, from the spec) ==
HIP kernels (the GPU tests).

Only the order of equal keys needs help: std::sort is not stable, so the two sorts of A8 (pile-up by length, supporters by
coordinate) go through `sort_perm`, the oracle's libstdc++ replay that IS pinned to the reference's comparators."""
import numpy as np


def c_div(a, b):
    """C integer division (truncation toward zero)."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b > 0) else -q


def coverage(ab, ae, c):
    """A1: bins of one pile-up at cutoff c."""
    if len(ab) == 0:
        return np.zeros(0, np.int64)
    ev_b, ev_e = ab.astype(np.int64) + c, ae.astype(np.int64) - c
    K = int(max(ev_b.max(), ev_e.max())) // 40 + 2
    edges = 40 * np.arange(K, dtype=np.int64)
    return (ev_b[None, :] < edges[:, None]).sum(1) - (ev_e[None, :] < edges[:, None]).sum(1)


def qv_mask(q, tspace):
    """A5."""
    s = e = mx = maxs = maxe = 0
    n = len(q)
    for j in range(n):
        if q[j] < 40 and j < n - 1:
            e += 1
        else:
            if e - s > mx:
                maxs, maxe, mx = s, e, e - s
            s = e = j + 1
    return maxs * tspace, maxe * tspace


def coverage_mask(covc, min_cov):
    """A4: (maxstart, maxend), (msc, mec)."""
    start = end = maxlen = maxstart = maxend = sc = ec = msc = mec = 0
    for j in range(len(covc)):
        if max(0, int(covc[j]) - min_cov) > 0:
            end, ec = 40 * j, j
        else:
            if end > start and end - start - 40 > maxlen:
                maxlen, maxstart, maxend, msc, mec = end - start - 40, start + 40, end, sc + 1, ec
            start = end = 40 * j
            sc = ec = j
    return (maxstart, maxend), (msc, mec)


def annotate(cov0, mask, min_cov, P):
    """A2 + A6 + A7: merged list of (position, type)."""
    K = len(cov0)
    out = []
    for j in range(0, K - 2):
        if not (mask[0] + P["NHR"] <= 40 * j <= mask[1] - P["NHR"]):
            continue
        thr = min(max(c_div(int(cov0[j]) + min_cov, P["CF"]), P["MINRA"]), P["MAXRA"])
        g = int(cov0[j + 1]) - int(cov0[j])
        if g > thr:
            out.append((40 * j, 1))
        elif g < -thr:
            out.append((40 * j, -1))
    it = 0
    while it + 1 < len(out):
        a, b = out[it], out[it + 1]
        if a[1] == 1 and b[1] == 1 and b[0] - a[0] < P["GAP"]:
            del out[it + 1]
        elif a[1] == -1 and b[1] == -1 and b[0] - a[0] < P["GAP"]:
            del out[it]
        else:
            it += 1
    return out


def gate_skips(cov0, mask, P):
    """A8 gate, float32 like the reference (a 0/0 is NaN and NaN < 10 is false: the read is NOT skipped)."""
    ks = [k for k in range(len(cov0)) if mask[0] <= 40 * k <= mask[0] + P["NHR"]]
    ke = [k for k in range(len(cov0)) if mask[1] - P["NHR"] <= 40 * k <= mask[1]]
    S, E = np.float32(sum(int(cov0[k]) for k in ks)), np.float32(sum(int(cov0[k]) for k in ke))
    with np.errstate(all="ignore"):
        d = np.float32(np.float32(E) / np.float32(len(ke))) - np.float32(np.float32(S) / np.float32(len(ks)))
        return bool(np.abs(d) < np.float32(10))


def hinge_scan(x, m_edge, sign, P):
    """A8 scan over the sorted supporters x = [(first, second)]: sign = +1 for type -1 annotations (ascending abpos, distances
    first - edge), -1 for type +1 (descending aepos, distances edge - first).  Returns bridged."""
    considered = to_end = 0
    s = len(x)
    for idx in range(s):
        f, sec = x[idx]
        if sign * (f - m_edge) < P["BIN"]:
            considered += 1
            to_end += 1
            if to_end > P["UNB"] or (considered > P["UNB"] and sign * (f - x[0][0]) > P["BIN"]):
                return False
        elif sec < P["TH"]:
            considered += 1
            if to_end > P["UNB"] or (considered > P["UNB"] and sign * (f - x[0][0]) > P["BIN"]):
                return False
        elif sec > P["TH"]:
            considered += 1
            pile, j = 1, idx + 1
            while j < s and sign * (x[j][0] - f) < P["BIN"]:
                pile += 1
                j += 1
            if pile > P["PIL"]:
                return True
    return True


def filter_part(rlen, qvm, row_ptr, a_span, b_span, b_flag, r_begin, r_end, maskvec, min_cov, P, sort_perm):
    """One .las part (A3 + A4 + A6-A8).  maskvec is the running table of all reads (later parts still (0, 0)); returns the new
    MIN_COV, per-read cmask, annotations and hinges for reads r_begin..r_end, and the cutoff-0 bins."""
    reads = range(r_begin, r_end + 1)
    cov0, covc = {}, {}
    means = []
    for i in reads:
        s, e = int(row_ptr[i]), int(row_ptr[i + 1])
        cov0[i] = coverage(a_span[s:e, 0], a_span[s:e, 1], 0)
        covc[i] = coverage(a_span[s:e, 0], a_span[s:e, 1], P["CUT"])
        if rlen[i] >= 5000:
            means.append(c_div(int(cov0[i].sum()), max(1, len(cov0[i]))))
    cov_est = sorted(means)[len(means) // 2]
    if P["EC"] != 0:
        cov_est = P["EC"]
    min_cov = max(min_cov, c_div(cov_est, 3))
    cmask = {}
    for i in reads:
        (ms, me), cmask[i] = coverage_mask(covc[i], min_cov)
        q = qvm[i] if qvm is not None else (0, 0)
        if P["USE_QV"] and P["USE_COV"]:
            maskvec[i] = (max(ms, q[0]), min(me, q[1]))
        elif P["USE_COV"]:
            maskvec[i] = (ms, me)
        else:
            maskvec[i] = tuple(q)
    annos, hinges = {}, {}
    for i in reads:
        annos[i] = annotate(cov0[i], maskvec[i], min_cov, P)
        hinges[i] = []
        if not annos[i] or gate_skips(cov0[i], maskvec[i], P):
            continue
        s, e = int(row_ptr[i]), int(row_ptr[i + 1])
        n = e - s
        length = (a_span[s:e, 1] - a_span[s:e, 0] + b_span[s:e, 1] - b_span[s:e, 0]).astype(np.int32)
        order = sort_perm(length, descending=True)            # the pile-up as filter.cpp:565-567 leaves it
        for (p, t) in annos[i]:
            sup = []
            for k in order:
                ab, ae = int(a_span[s + k, 0]), int(a_span[s + k, 1])
                bb, be = int(b_span[s + k, 0]), int(b_span[s + k, 1])
                b, comp = int(b_flag[s + k] & 0x7FFFFFFF), int(b_flag[s + k] >> 31)
                mf, msd = maskvec[b]
                if comp == 0:
                    R, L = max(msd - be, 0), max(bb - mf, 0)
                else:
                    R, L = max(bb - mf, 0), max(msd - be, 0)
                if t == -1 and R > P["TH"] and p - P["TOL"] < ae < p + P["TOL"]:
                    sup.append((ab, L))
                if t == 1 and L > P["TH"] and p - P["TOL"] < ab < p + P["TOL"]:
                    sup.append((ae, R))
            if len(sup) < P["SUP"]:
                continue
            keys = np.array([x[0] for x in sup], np.int32)
            sup = [sup[j] for j in sort_perm(keys, descending=(t == 1))]
            bridged = hinge_scan(sup, maskvec[i][0] if t == -1 else maskvec[i][1], 1 if t == -1 else -1, P)
            if not bridged and len(sup) > P["SUP"]:
                hinges[i].append((p, t))
    return min_cov, cmask, annos, hinges, cov0
