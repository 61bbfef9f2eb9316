"""K5 (hinge_select_edges -> k_select_edges): the greedy best-overlap selection of `hinge layout` against a statement-level
Python restatement of hinging.cpp:1911-2148 on random match lists that hit every branch (poisoned FORWARD matches, internal
matches that override a plain one by weight, inactive partners, hinges outside the tolerance, the first-hinge `break`).
The executables' layout outputs (tests/test_cli_gpu.py: .edges.hinges, .edges.hinges2, .edges.skipped, .deadends.txt) hold the
same kernel against the CPU oracle on real pipelines."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FORWARD, BACKWARD, FORWARD_INTERNAL, BACKWARD_INTERNAL = 0, 1, 12, 13


def reference_selection(active, fwd, bwd, hinges, killed, tol, slack):
    """hinging.cpp:1911-2148, variable for variable.  fwd / bwd: per read a list of dicts (b, comp, type, active, weight, eff_bb,
    eff_be, bb, be, gid).  Returns chosen[2][n] (gid or -1), hinge_pos printed with it, and the skipped prints (gid list)."""
    n = len(active)
    chosen = -np.ones((2, n), np.int64)
    hp = -np.ones((2, n), np.int64)
    skipped = []
    hinge_pos = -1
    for i in range(n):
        if not active[i]:
            continue
        forward = forward_internal = backward = backward_internal = 0
        chosen_match = None
        for m in fwd[i]:
            if m["active"] and active[m["b"]]:
                if m["type"] == FORWARD and forward == 0:
                    poisoned = False
                    for (kpos, ktype) in killed[i]:
                        if m["comp"] != 1 and ktype == -1 and kpos > m["eff_be"]:
                            skipped.append(m["gid"]); poisoned = True
                        elif m["comp"] == 1 and ktype == 1 and kpos < m["eff_bb"]:
                            skipped.append(m["gid"]); poisoned = True
                    if not poisoned:
                        chosen_match = m; hinge_pos = -1; forward = 1
                elif m["type"] == FORWARD_INTERNAL and len(hinges[m["b"]]) > 0 and forward_internal == 0:
                    start = m["be"] if m["comp"] == 1 else m["bb"]
                    for (hpos, htype, hact) in hinges[m["b"]]:
                        if start > hpos - tol and start < hpos + tol and htype == 1 - 2 * m["comp"] and hact:
                            if forward == 0 or m["weight"] > chosen_match["weight"] - 2 * slack:
                                chosen_match = m; forward = 1; forward_internal = 1; hinge_pos = hpos
                            break
        if chosen_match is not None:
            chosen[0, i] = chosen_match["gid"]; hp[0, i] = hinge_pos
            chosen_match = None
        for m in bwd[i]:
            if m["active"] and active[m["b"]]:
                if m["type"] == BACKWARD and backward == 0:
                    poisoned = False
                    for (kpos, ktype) in killed[i]:
                        if m["comp"] != 1 and ktype == 1 and kpos < m["eff_bb"]:
                            skipped.append(m["gid"]); poisoned = True
                        elif m["comp"] == 1 and ktype == -1 and kpos > m["eff_be"]:
                            skipped.append(m["gid"]); poisoned = True
                    if not poisoned:
                        chosen_match = m; backward = 1; hinge_pos = -1
                elif m["type"] == BACKWARD_INTERNAL and len(hinges[m["b"]]) > 0 and backward_internal == 0:
                    end = m["bb"] if m["comp"] == 1 else m["be"]
                    for (hpos, htype, hact) in hinges[m["b"]]:
                        if end > hpos - tol and end < hpos + tol and htype == -1 + 2 * m["comp"] and hact:
                            if backward == 0 or m["weight"] > chosen_match["weight"] - 2 * slack:
                                chosen_match = m; backward = 1; backward_internal = 1; hinge_pos = hpos
                            break
        if chosen_match is not None:
            chosen[1, i] = chosen_match["gid"]; hp[1, i] = hinge_pos
    return chosen, hp, skipped


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_select_edges_matches_the_reference_loop(seed):
    from hinge_amd import capi
    rng = np.random.default_rng(seed)
    n = 400
    tol, slack = int(rng.choice([50, 150])), int(rng.choice([10, 400, 1000]))
    active = (rng.random(n) < 0.85).astype(np.uint8)
    hinges, killed = [], []
    for i in range(n):
        hinges.append([(int(rng.integers(0, 9000)), int(rng.choice([-1, 1])), int(rng.random() < 0.7)) for _ in range(int(rng.integers(0, 4)) if rng.random() < 0.4 else 0)])
        killed.append([(int(rng.integers(0, 9000)), int(rng.choice([-1, 1]))) for _ in range(int(rng.integers(1, 3)) if rng.random() < 0.25 else 0)])
    recs, fwd, bwd = [], [], []
    off = {0: [], 1: []}
    lists = {0: fwd, 1: bwd}
    for dirn in (0, 1):                  # the kernel takes one record array: all forward lists, then all backward lists
        off[dirn].append(len(recs))
        for i in range(n):
            ms = []
            w = int(rng.integers(8000, 20000))
            for _ in range(int(rng.integers(0, 9))):
                w -= int(rng.integers(0, 900))           # compare_overlap_weight order: descending weight
                b = int(rng.integers(0, n))
                comp = int(rng.random() < 0.5)
                t = int(rng.choice([FORWARD, FORWARD_INTERNAL, BACKWARD, BACKWARD_INTERNAL] if rng.random() < 0.15 else
                                   ([FORWARD, FORWARD_INTERNAL] if dirn == 0 else [BACKWARD, BACKWARD_INTERNAL])))
                bb = int(rng.integers(0, 8000)); be = bb + int(rng.integers(500, 4000))
                if hinges[b] and rng.random() < 0.6:     # land the anchor near one of B's hinges
                    hpos = hinges[b][int(rng.integers(0, len(hinges[b])))][0]
                    d = int(rng.integers(-tol - 20, tol + 20))
                    if (dirn == 0) == (comp == 0): bb = hpos + d
                    else: be = hpos + d
                m = dict(b=b, comp=comp, type=t, active=int(rng.random() < 0.9), weight=w, eff_bb=bb + int(rng.integers(0, 200)),
                         eff_be=be - int(rng.integers(0, 200)), bb=bb, be=be, gid=len(recs))
                recs.append([m[k] for k in ("b", "comp", "type", "active", "weight", "eff_bb", "eff_be", "bb", "be")])
                ms.append(m)
            lists[dirn].append(ms)
            off[dirn].append(len(recs))
    h_off = np.concatenate([[0], np.cumsum([len(h) for h in hinges])])
    k_off = np.concatenate([[0], np.cumsum([len(k) for k in killed])])
    h_rec = np.array([x for h in hinges for x in h], np.int32).reshape(-1, 3)
    k_rec = np.array([x for k in killed for x in k], np.int32).reshape(-1, 2)
    want_c, want_h, want_skipped = reference_selection(active, fwd, bwd, hinges, killed, tol, slack)
    ctx = capi.Context(0)
    got_c, got_h, poison = ctx.select_edges(active, off[0], off[1], np.array(recs, np.int32).reshape(-1, 9), h_off, h_rec, k_off, k_rec, tol, slack)
    ctx.close()
    assert np.array_equal(got_c, want_c)
    assert np.array_equal(got_h[got_c >= 0], want_h[want_c >= 0])
    assert np.array_equal(poison, np.bincount(np.array(want_skipped, np.int64), minlength=len(recs)))
    # every branch was taken somewhere
    types = np.array(recs)[:, 2]
    assert (want_c >= 0).any() and (want_c < 0).any() and len(want_skipped) > 0
    assert np.isin(types[want_c[0][want_c[0] >= 0]], [FORWARD_INTERNAL]).any() and np.isin(types[want_c[1][want_c[1] >= 0]], [BACKWARD_INTERNAL]).any()
