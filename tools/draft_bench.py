#!/usr/bin/env python
"""`hinge draft`'s ladder step at E. coli size: what draft.cpp:597-691 does once per backbone
window - falcon's banded O(ND) alignment of every ladder member against the ladder's template (DW_banded.c:97-311) and falcon's
consensus over their alignment tags (falcon.c:246-517).

    python tools/draft_bench.py [--ladders 5000] [--members 25] [--length 900] [--err 0.12] [--steps 3] [--cpu-seconds 10] [--no-cpu]

Workload: a 4.6 Mb backbone cut every `[draft] tspace` = 900 bases is ~5 100 ladders; at 25x each has ~25 members of ~900 bases, each
a noisy copy (substitutions, insertions, deletions: --err in all) of the window's true sequence, on either strand inside a longer
stored read.  The members go into a read DB; `hinge_draft_ladders` (k_draft_align: one wavefront per member, k_draft_cns: one
wavefront per ladder) gets (read, strand, start, end) per member, as `draft_assembly` calls it.

CPU baseline: the REFERENCE'S OWN falcon (`ref_falcon_ladder` of oracle/ref_shim.cpp over lib/falcon.c + DW_banded.c + kmer_lookup.c
compiled unmodified, `kind: "reference"`), one thread, on as many of the same ladders as fit --cpu-seconds; every ladder it ran must
give the GPU's string byte for byte or the run fails.  Prints one JSON line."""
import argparse
import ctypes
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def noisy(rng, t, err):
    """A copy of base array t with substitutions / insertions / deletions at a total rate of err (vectorised tests/draft_common.noisy)."""
    import numpy as np
    ps, pi, pd = err * 0.3, err * 0.45, err * 0.25
    L = len(t)
    r = rng.random(L)
    ins = rng.random(L) < pi
    keep = r >= pd
    base = np.where(r < pd + ps, rng.integers(0, 4, L).astype(np.uint8), t)
    cnt = ins.astype(np.int64) + keep.astype(np.int64)
    start = np.concatenate([[0], np.cumsum(cnt)])
    out = np.empty(int(start[-1]), np.uint8)
    out[start[:-1][ins]] = rng.integers(0, 4, int(ins.sum())).astype(np.uint8)
    out[(start[:-1] + ins)[keep]] = base[keep]
    return out if len(out) else np.zeros(1, np.uint8)


def make(args, wd):
    import numpy as np
    from hinge_amd import formats
    rng = np.random.default_rng(args.seed)
    reads, rungs, templates, members = [], [], [], []
    k = 0
    for l in range(args.ladders):
        L = int(rng.integers(args.length * 9 // 10, args.length * 11 // 10 + 1))
        truth = rng.integers(0, 4, L).astype(np.uint8)
        n = max(2, int(rng.poisson(args.members)))
        n = min(n, 64)
        rg, ms = [], []
        for m in range(n):
            fwd = noisy(rng, truth, args.err)
            pad_l, pad_r = int(rng.integers(0, 40)), int(rng.integers(0, 40))      # the member sits inside a longer read
            body = np.concatenate([rng.integers(0, 4, pad_l).astype(np.uint8), fwd, rng.integers(0, 4, pad_r).astype(np.uint8)])
            strand = int(rng.integers(0, 2))
            reads.append((3 - body[::-1]).astype(np.uint8) if strand else body)
            rg.append((k, strand, pad_l, pad_l + len(fwd)))
            ms.append(fwd)
            k += 1
        rungs.append(rg)
        members.append(ms)
        templates.append(int(rng.integers(0, n)))
    db = os.path.join(wd, "L")
    formats.write_db(db, np.asarray([len(r) for r in reads], np.int32), bases=reads)
    return db, rungs, templates, members


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ladders", type=int, default=5000)
    ap.add_argument("--members", type=int, default=25)
    ap.add_argument("--length", type=int, default=900)
    ap.add_argument("--err", type=float, default=0.12)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    import numpy as np
    from hinge_amd import capi
    wd = tempfile.mkdtemp(prefix="hinge_draft_bench_")
    t0 = time.time()
    db, rungs, templates, members = make(args, wd)
    gen_s = time.time() - t0
    n_jobs = sum(len(r) for r in rungs)
    member_bases = int(sum(len(m) for ms in members for m in ms))
    # what the aligner has to touch at least: every member against its template, O(N D) diagonal steps ~ (q + t) * edit distance
    out = {"ladders": args.ladders, "member_alignments": n_jobs, "member_bases": member_bases, "mean_members": n_jobs / args.ladders,
           "length": args.length, "error_rate": args.err, "generate_s": round(gen_s, 2)}
    ctx = capi.Context(0)
    dr = capi.Draft(ctx, db)
    got = dr.ladders(rungs, templates)            # warm-up (allocations)
    ctx.profile_enable(16 * args.steps)
    t = time.time()
    for _ in range(args.steps):
        got = dr.ladders(rungs, templates)
    call_ms = (time.time() - t) * 1e3 / args.steps
    rep = ctx.profile_report()
    out["call_ms"] = round(call_ms, 3)            # host job tables + H2D + kernels + D2H of the strings
    out["kernels_ms"] = {k: round(v[0] / args.steps, 4) for k, v in rep.items() if k.startswith("k_draft") and v[1]}
    ksum = sum(out["kernels_ms"].values())
    out["kernels_ms_sum"] = round(ksum, 4)
    out["member_bases_per_s_kernels"] = member_bases / (ksum * 1e-3) if ksum else None
    out["ladders_per_s_kernels"] = args.ladders / (ksum * 1e-3) if ksum else None
    out["consensus_bases"] = int(sum(len(g) for g in got))
    if not args.no_cpu:
        import oracle
        import draft_common as dc
        ref = oracle.ref_lib()
        kind = "reference"
        if ref is not None and hasattr(ref, "ref_falcon_ladder"):
            fn = dc.bind_ref(ref).ref_falcon_ladder
        else:
            fn = dc.bind(oracle.oracle_lib()).oracle_falcon_ladder
            kind = "port"
        code = "acgt"
        t = time.time()
        done = 0
        bases = 0
        for l in range(args.ladders):
            mem = ["".join(code[x] for x in m) for m in members[l]]
            want = dc.ladder_call(fn, mem, templates[l])[1]
            assert want == got[l], "ladder %d: the GPU's consensus differs from the CPU side's (%d vs %d bases)" % (l, len(got[l]), len(want))
            done += 1
            bases += sum(len(m) for m in mem)
            if time.time() - t > args.cpu_seconds:
                break
        wall = time.time() - t
        out["cpu"] = {"kind": kind, "cores": 1, "ladders": done, "member_bases": bases, "wall_s": round(wall, 3),
                      "ladders_per_s": done / wall, "member_bases_per_s": bases / wall,
                      "sample": "the first %d of the %d ladders (python string marshalling included: ~2 %% of the time)" % (done, args.ladders)}
        out["byte_identical"] = True
        out["speedup_kernels_vs_cpu"] = round(out["ladders_per_s_kernels"] / out["cpu"]["ladders_per_s"], 1) if ksum else None
        out["speedup_call_vs_cpu"] = round((args.ladders / (call_ms * 1e-3)) / out["cpu"]["ladders_per_s"], 1)
    out["roofline"] = draft_roofline(out["kernels_ms"])
    ctx.close()
    print(json.dumps(out))


def draft_roofline(kernels_ms):
    """k_draft_align / k_draft_cns against vector-instruction ISSUE (integer compares on 2-bit bases along one alignment path per
    wavefront: no HBM stream to price - the whole working set of a ladder is a few hundred KB).  achieved = SQ_INSTS_VALU of one launch
    (newest profiles/*_draft_rocprofv3_sq_summary.csv: a property of the instruction stream and this seeded data set) / the launch
    time measured HERE; peak = one wave64 vector instruction per SIMD per 4 cycles: 256 CUs x 4 SIMDs x 2.4 GHz / 4."""
    import csv
    import glob
    peak = 256 * 4 * 2.4e9 / 4
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_draft_rocprofv3_sq_summary.csv")), key=lambda f: (os.path.basename(f)[:2], os.path.basename(f)))
    if not files:
        return None
    res = {}
    for kname in ("k_draft_align", "k_draft_cns"):
        ms = kernels_ms.get(kname)
        valu = None
        for row in csv.DictReader(open(files[-1])):
            if kname in row["kernel"] and row["counter"] == "SQ_INSTS_VALU":
                valu = float(row["mean_value"])
        if ms and valu:
            res[kname] = {"bound": "valu-issue", "achieved": valu / (ms * 1e-3), "peak": peak, "unit": "wave-instructions/s", "frac": valu / (ms * 1e-3) / peak,
                          "valu_instructions_per_launch": valu, "avg_launch_ms": ms}
    res["counters_source"] = os.path.relpath(files[-1], ROOT)
    return res


if __name__ == "__main__":
    main()
