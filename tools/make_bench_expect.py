#!/usr/bin/env python
"""Expected results of bench.py's read sets from the CPU oracle -> tests/golden/bench_expect.json.

bench.py asserts, at every N, the number of hinges AND a digest of the (read, position, type) rows of every part of every rank
against this file.  Run here (CPU only, ~20 min, <= 20 GB):  python tools/make_bench_expect.py [--parts 4] [--worlds 1,2,4,8]

How the N > 2 rows are derived without an oracle run over N blocks (8 blocks = 2.1e8 overlaps = ~70 GB in the oracle):
the data sets of different teams (hinge_amd/benchsets.py) share nothing but MIN_COV = max(min_cov, cov_est / 3) with cov_est the
median over ALL blocks (filter.cpp:660-678; cov_est is used nowhere else).  So
  1. every team's 2-block data set runs through the oracle on its own (which IS the N = 2 world for team 0);
  2. the oracle hands out the per-read mean coverages it took its median from (oracle_probe_means: its own values, not a
     restatement) - the world's cov_est is the median of the union, taken exactly as filter.cpp:660-664 does
     (nth_element at size / 2);
  3. a team whose own MIN_COV differs from the world's is run again with `ec` = the world's cov_est (filter.cpp:671).
One more thing the oracle cannot show by itself: `.hinges.txt` stops before the LAST A read of a run (filter.cpp:1091), which in
the N-block world is the last rank's last read only.  Team runs therefore get one sentinel read behind the last one (999 bp -
outside the median, which skips reads < 5000 bp - with a single overlap record of its own), so every real read's hinges are
printed; the rule is then applied for the world's last rank here, and checked against a plain (sentinel-free) oracle run
of team 0 at N = 2.
"""
import argparse
import ctypes
import json
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def hinge_rows(path):
    rows = []
    for line in open(path):
        tok = line.split()
        for j in range(1, len(tok) - 1, 2):
            rows.append((int(tok[0]), int(tok[j]), int(tok[j + 1])))
    return np.array(rows, np.int64).reshape(-1, 3)


def with_sentinel(d):
    """d plus one 999-bp read behind the last one whose only overlap record has B = read 0."""
    import dataclasses
    n = d.n_reads
    app = lambda a, v: np.concatenate([a, np.asarray([v], dtype=a.dtype)])
    bf = list(d.block_first)
    bf[-1] = n + 1
    return dataclasses.replace(d, rlen=app(d.rlen, 999), aread=app(d.aread, n), bread=app(d.bread, 0), comp=app(d.comp, 0), ab=app(d.ab, 0),
                               ae=app(d.ae, 900), bb=app(d.bb, 0), be=app(d.be, 900), block_first=bf)


def run_oracle(lib, d, ini_text):
    """(hinge rows, per-read means the oracle fed into its median, its natural cov_est)."""
    from hinge_amd import synth
    tmp = tempfile.mkdtemp(prefix="hinge_expect_")
    try:
        synth.formats.write_db(os.path.join(tmp, "G"), d.rlen, block_first=[0, d.n_reads], write_bases=False)
        synth.write_las_file(d, os.path.join(tmp, "G.las"))
        open(os.path.join(tmp, "nominal.ini"), "w").write(ini_text)
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            rc = lib.oracle_filter(b"G", b"G.las", 0, b"G", b"nominal.ini", b"")
        finally:
            os.chdir(cwd)
        assert rc == 0, rc
        est = ctypes.c_int()
        n = lib.oracle_probe_means(None, 0, ctypes.byref(est))
        means = np.zeros(n, np.int32)
        lib.oracle_probe_means(means.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), n, None)
        return hinge_rows(os.path.join(tmp, "G.hinges.txt")), means, int(est.value)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def min_cov_of(ini_min_cov, cov_est):
    return max(ini_min_cov, int(cov_est / 3))


def entry(rows, lo, hi, drop_read=None):
    from hinge_amd import benchsets
    sel = (rows[:, 0] >= lo) & (rows[:, 0] < hi)
    if drop_read is not None:
        sel &= rows[:, 0] != drop_read
    r = rows[sel].copy()
    r[:, 0] -= lo
    return {"hinges": int(len(r)), "digest": benchsets.digest(r)}


def team_runs(lib, base, n_teams, p, ini_text, log=print):
    """Step 1: every team's data set of part p through the oracle (sentinel read appended)."""
    from hinge_amd import benchsets, synth
    team = []
    for q in range(n_teams):
        spec, _ = benchsets.part_spec(base, benchsets.TEAM * n_teams, benchsets.TEAM * q, p)
        d = synth.generate(spec)
        ds = with_sentinel(d)
        rows, means, est = run_oracle(lib, ds, ini_text)
        assert len(means) == int(np.sum(d.rlen >= 5000))
        chk = None
        if q == 0:      # the `i < r_end` rule, emulated in world_entries, against the oracle itself on the sentinel-free N = 2 world
            chk, means0, est0 = run_oracle(lib, d, ini_text)
            assert est0 == est and np.array_equal(means0, means)
        team.append({"d": d, "ds": ds, "rows": rows, "means": means, "est": est, "plain_rows": chk})
        log("part", p, "team", q, "reads", d.n_reads, "records", d.novl, "cov_est", est, "hinge rows", len(rows))
    return team


def world_entries(lib, team, N, ini_text, ini_min_cov, log=print):
    """Steps 2 and 3 for the world of N ranks: its cov_est and the per-rank entries."""
    from hinge_amd import benchsets
    teams = team[:N // benchsets.TEAM]
    allm = np.sort(np.concatenate([t["means"] for t in teams]))
    est = int(allm[len(allm) // 2])
    ranks, reads, records = [], [], []
    for q, t in enumerate(teams):
        rows = t["rows"]
        if min_cov_of(ini_min_cov, est) != min_cov_of(ini_min_cov, t["est"]):
            key = "rows_ec_%d" % est
            if key not in t:
                t[key] = run_oracle(lib, t["ds"], ini_text.replace("[filter]\n", "[filter]\nec = %d\n" % est, 1))[0]
                log("  N=%d: team %d rerun with ec = %d (own cov_est %d)" % (N, q, est, t["est"]))
            rows = t[key]
        d = t["d"]
        for k in range(benchsets.TEAM):
            lo, hi = d.block_first[k], d.block_first[k + 1]
            last = benchsets.TEAM * q + k == N - 1
            ranks.append(entry(rows, lo, hi, drop_read=int(d.aread[-1]) if last else None))
            reads.append(hi - lo)
            records.append(int(np.sum((d.aread >= lo) & (d.aread < hi))))
    if N == 2:
        t = teams[0]
        d = t["d"]
        plain = [entry(t["plain_rows"], d.block_first[k], d.block_first[k + 1]) for k in range(2)]
        assert plain == ranks, ("the emulated `i < r_end` drop differs from the oracle's own run", plain, ranks)
    return {"cov_est": est, "ranks": ranks, "reads": reads, "records": records}


def merged_world(team, N):
    """The N-block world as ONE data set (teams concatenated, ids shifted): what the derivation above stands for.  Only
    small workloads fit the oracle this way (tests/test_bench_expect.py does it for one)."""
    import dataclasses
    from hinge_amd import benchsets
    teams = [t["d"] for t in team[:N // benchsets.TEAM]]
    off = np.concatenate([[0], np.cumsum([d.n_reads for d in teams])]).astype(np.int64)
    cat = lambda f, shift=False: np.concatenate([(getattr(d, f) + (off[q] if shift else 0)).astype(getattr(d, f).dtype) for q, d in enumerate(teams)])
    bf = [int(off[q] + b) for q, d in enumerate(teams) for b in d.block_first[:-1]] + [int(off[-1])]
    return dataclasses.replace(teams[0], rlen=cat("rlen"), aread=cat("aread", True), bread=cat("bread", True), comp=cat("comp"), ab=cat("ab"),
                               ae=cat("ae"), bb=cat("bb"), be=cat("be"), block_first=bf)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2_ecoli160")
    ap.add_argument("--parts", type=int, default=4)
    ap.add_argument("--worlds", default="1,2,4,8")
    args = ap.parse_args()
    import oracle
    from hinge_amd import benchsets, synth
    from bench import INI
    lib = oracle.oracle_lib()
    ini_min_cov = 5
    assert "min_cov = 5;" in INI
    worlds = sorted(int(w) for w in args.worlds.split(","))
    path = os.path.join(ROOT, "tests", "golden", "bench_expect.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    base = synth.CONFIGS[args.workload]
    W = out.setdefault(args.workload, {}).setdefault("worlds", {})
    log = lambda *a: print(*a, flush=True)
    for p in range(args.parts):
        if 1 in worlds:
            spec, _ = benchsets.part_spec(base, 1, 0, p)
            d = synth.generate(spec)
            rows, means, est = run_oracle(lib, d, INI)
            W.setdefault("1", {})[str(p)] = {"cov_est": est, "ranks": [entry(rows, 0, d.n_reads)], "reads": [d.n_reads], "records": [d.novl]}
            log("N=1 part", p, W["1"][str(p)])
            del d
        n_teams = max(worlds) // benchsets.TEAM
        if n_teams == 0:
            continue
        team = team_runs(lib, base, n_teams, p, INI, log)
        for N in worlds:
            if N == 1:
                continue
            W.setdefault(str(N), {})[str(p)] = world_entries(lib, team, N, INI, ini_min_cov, log)
            log("N=%d part %d cov_est %d" % (N, p, W[str(N)][str(p)]["cov_est"]), [r["hinges"] for r in W[str(N)][str(p)]["ranks"]])
        json.dump(out, open(path, "w"), indent=1, sort_keys=True)
        del team


if __name__ == "__main__":
    main()
