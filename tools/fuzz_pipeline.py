#!/usr/bin/env python
"""Randomised differential test: random generator settings and random nominal.ini values, the three executables against
the oracle, every output file byte for byte.   tools/fuzz_pipeline.py [--cases 40] [--seed 1]   (needs a GPU)"""
import argparse
import filecmp
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FILES = ["G.mas", "G.cmas", "G.repeat.txt", "G.hinges.txt", "G.coverage.txt", "G.cov.flag", "G.self.flag", "G.max", "G.contained.txt",
         "G.killed.hinges", "G.edges.hinges", "G.edges.hinges2", "G.hinge.list", "G.deadends.txt", "G.hgraph", "G.edges.greedy",
         "G.edges.1", "G.edges.2", "G.edges.skipped", "edges.g_out.txt"]


def random_case(rng):
    from hinge_amd import synth
    lmin = int(rng.integers(1500, 6000))
    shape = rng.random()
    if shape < 0.12:      # deep: pile-ups of thousands of overlaps (both k_hinge_call instances, the serial exact kernel beyond 4096)
        genome, cov, lmax, lmean = int(rng.integers(25_000, 50_000)), float(rng.uniform(250, 600)), int(lmin + rng.integers(4000, 9000)), float(rng.uniform(5000, 8000))
    elif shape < 0.24:    # long reads: two and four LDS slots per read in K2, reads beyond a workgroup's LDS
        genome, cov, lmax, lmean = int(rng.integers(200_000, 400_000)), float(rng.uniform(25, 45)), int(rng.integers(40_000, 130_000)), float(rng.uniform(15_000, 30_000))
    else:
        genome, cov, lmax, lmean = int(rng.integers(60_000, 220_000)), float(rng.uniform(22, 70)), int(lmin + rng.integers(4000, 16000)), float(rng.uniform(6000, 11000))
    spec = synth.SynthSpec(
        genome_len=genome, coverage=cov, len_dist=("lognormal" if shape < 0.24 and shape >= 0.12 else str(rng.choice(["uniform", "lognormal"]))),
        len_min=lmin, len_max=lmax, len_mean=lmean, len_sigma=float(rng.uniform(0.2, 0.7)),
        n_repeat_families=int(rng.integers(0, 4)), repeat_len=(int(rng.integers(1500, 4000)), int(rng.integers(4000, 9000))),
        repeat_copies=(2, int(rng.integers(2, 5))), inverted_copies=bool(rng.integers(0, 2)), chimera_frac=float(rng.choice([0.0, 0.0, 0.02, 0.05])),
        min_ovl=int(rng.choice([500, 1000, 1500])), end_jitter=int(rng.choice([0, 10, 25, 60])), indel_max=int(rng.choice([0, 3, 6, 12])),
        tspace=int(rng.choice([100, 100, 100, 50, 200])), seed=int(rng.integers(1, 1 << 30)), n_blocks=int(rng.choice([1, 1, 2, 3])),
        with_qv=bool(rng.integers(0, 2)), tie_quantum=int(rng.choice([0, 0, 0, 40, 100])), short_reads=int(rng.choice([0, 0, 10])),
        orphan_reads=int(rng.choice([0, 0, 4])), self_overlap_reads=int(rng.choice([0, 0, 4])))
    # round 5: three cases in four carry jittered traces (per-segment B advances of tspace +- 15 / 30 %); taken from the seed,
    # so the draws below are those of the earlier rounds' cases
    spec.trace_jitter = [0, 15, 15, 30][spec.seed % 4] * spec.tspace // 100
    filt, lay = [], []
    if rng.random() < 0.5:
        filt.append("cut_off = %d" % int(rng.choice([0, 100, 200, 300, 300, 400, 310])))
    if rng.random() < 0.4:
        filt.append("theta = %d" % int(rng.choice([100, 200, 300, 500])))
    if rng.random() < 0.3:
        filt.append("aln_threshold = %d" % int(rng.choice([500, 1000, 2500])))
    if rng.random() < 0.3:
        filt.append("min_cov = %d" % int(rng.choice([0, 3, 5, 12])))
    if rng.random() < 0.3:
        filt.append("ec = %d" % int(rng.choice([20, 45, 90])))
    if rng.random() < 0.3:
        filt.append("hinge_min_support = %d\nhinge_unbridged = %d\nhinge_min_pileup = %d" % (int(rng.integers(2, 9)), int(rng.integers(1, 8)), int(rng.integers(2, 9))))
    if rng.random() < 0.2:
        filt.append("no_hinge_region = %d\nrepeat_annotation_gap_threshold = %d" % (int(rng.choice([200, 500, 800])), int(rng.choice([100, 300, 600]))))
    if rng.random() < 0.3:
        lay.append("del_telomere = 1\ndel_telomeres = 1")
    if rng.random() < 0.3:
        lay.append("use_two_matches = 0")
    if rng.random() < 0.4:
        lay.append("min_connected_component_size = %d" % int(rng.choice([1, 2, 8])))
    if rng.random() < 0.3:
        lay.append("hinge_slack = %d\nhinge_tolerance = %d\nmatching_hinge_slack = %d" % (int(rng.choice([10, 500, 1000])), int(rng.choice([50, 150, 400])), int(rng.choice([100, 200, 500]))))
    return spec, "\n".join(filt) + ("\n" if filt else ""), "\n".join(lay) + ("\n" if lay else "")


def random_paths(rng, spec):
    """Which of the library's alternative code paths the executables are pushed onto, and whether the case runs on
    FASTA + PAF input instead of DB + .las (one block only: --mlas needs a DB)."""
    env = {}
    if rng.random() < 0.25:
        env["HINGE_NO_SPAN16"] = "1"                 # int32 spans instead of the 16|16 copy
    if rng.random() < 0.25:
        env["HINGE_DEBUG_GENERAL_MASK"] = "1"        # general two-histogram K2 instead of the 20-bp kernel
    r = rng.random()
    if r < 0.15:
        env["HINGE_DEBUG_FORCE_EXACT"] = "1"         # serial exact hinge kernel
    elif r < 0.35:
        env["HINGE_DEBUG_FORCE_EXACT"] = "2"         # exact std::sort replay in LDS
    if rng.random() < 0.3:
        env["HINGE_THREADS"] = str(int(rng.choice([1, 3, 16])))
    if rng.random() < 0.3:
        env["HINGE_K2_WGS"] = str(int(rng.choice([1, 2, 5, 8, 16])))   # few persistent workgroups: every wavefront of k_mask_annotate_q20 takes several reads
                                                                          # (8, 16: a multiple of the 8 XCDs, so the XCD-contiguous deal of the reads is on)
    if rng.random() < 0.2:
        env["HINGE_CALL_LIGHT"] = "0"                # open annotations straight to k_hinge_call<CAP> (no light kernel in front)
    if rng.random() < 0.2:
        env["HINGE_CALL_MINI"] = "1"                 # a quarter-size instance k_hinge_call<1024> in front of the second tier
    if rng.random() < 0.2:
        env["HINGE_K2_DEAL"] = "0"                   # round 2's longest-first order of the drawn reads
    if rng.random() < 0.2:
        env["HINGE_K2_HEAVY"] = str(int(rng.choice([0, 1])))   # the deep pile-ups first / left in storage order (default: spread over the first 60 %)
    paf = spec.n_blocks == 1 and rng.random() < 0.2
    # round 5 (drawn after everything above, so the earlier rounds' cases keep their draws)
    if rng.random() < 0.25:
        env["HINGE_K4_SOA"] = "1"                    # `hinge maximal` through the column form (k_trim_classify_stream) instead of the .las image
    elif rng.random() < 0.3:
        env["HINGE_K4_WAVES_PER_CU"] = str(int(rng.choice([1, 3, 7])))   # few persistent wavefronts in k_trim_classify_image: long window sequences per wavefront
        env["HINGE_K4_CAP"] = str(int(rng.choice([8192, 10240, 12288])))
    if rng.random() < 0.2:
        env["HINGE_COUNT_INT32"] = "1"               # the hinge kernels on the int32 span columns instead of the 16|16 copies
    # round 6 (drawn after everything above)
    if rng.random() < 0.25:
        env["HINGE_CALL_GROUP"] = "0"                # k_hinge_call<CAP> draws items one by one instead of reads with their item chains
    if rng.random() < 0.3:
        env["HINGE_COMM_ONE_RANK"] = "1"             # maximal / layout send their rows through a one-rank RCCL communicator (hinge_comm_allgather_rows)
    if rng.random() < 0.25:
        env["HINGE_CALL_LEAN"] = "0"                 # the full-size replay instance behind the light kernel instead of the 52-KiB one
    return env, paf


def run_case(k, spec, filt, lay, lib, keep_failures, env=None, paf=False):
    import conftest
    from hinge_amd import synth
    d = synth.generate(spec)
    if d.novl == 0 or int(d.rlen.max()) < 5000:
        return "skipped (no overlaps / no 5 kb read)"
    tmp = tempfile.mkdtemp(prefix="hinge_fuzz_")
    mlas = spec.n_blocks > 1
    try:
        src = os.path.join(tmp, "src")
        try:
            synth.write_dataset(d, src, "G", write_bases=False)
        except AssertionError:
            return "skipped (trace generator cannot express this indel / trace-spacing combination)"
        conftest.write_ini(os.path.join(src, "v.ini"), extra_filter=filt, extra_layout=lay)
        wd_o = conftest.clone_dataset(src, os.path.join(tmp, "oracle"))
        wd_h = conftest.clone_dataset(src, os.path.join(tmp, "hip"))
        hinge = os.path.join(ROOT, "hinge_amd", "bin", "hinge")
        penv = dict(os.environ, **(env or {}))
        got = []
        if paf:
            from hinge_amd import formats
            for wd in (wd_o, wd_h):
                formats.write_fasta(os.path.join(wd, "G.fasta"), d.rlen, seed=3)
                formats.write_paf(os.path.join(wd, "G.paf"), d.rlen, d.aread, d.bread, d.comp, d.ab, d.ae, d.bb, d.be)
            rcs = [conftest.run_in(wd_o, lib.oracle_filter_paf, b"G.fasta", b"G.paf", b"G", b"v.ini"),
                   conftest.run_in(wd_o, lib.oracle_maximal_paf, b"G.fasta", b"G.paf", b"G", b"v.ini"),
                   conftest.run_in(wd_o, lib.oracle_layout_paf, b"G.fasta", b"G.paf", b"G", b"G", b"v.ini")]
            for sub, extra in (("filter", []), ("maximal", []), ("layout", ["-o", "G"])):
                argv = [hinge, sub, "--fasta", "G.fasta", "--paf", "G.paf", "-x", "G", "--config", "v.ini"] + extra
                got.append(subprocess.run(argv, cwd=wd_h, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=penv).returncode)
        else:
            las = b"G" if mlas else b"G.las"
            rcs = [conftest.run_in(wd_o, lib.oracle_filter, b"G", las, int(mlas), b"G", b"v.ini", b""),
                   conftest.run_in(wd_o, lib.oracle_maximal, b"G", las, int(mlas), b"G", b"v.ini"),
                   conftest.run_in(wd_o, lib.oracle_layout, b"G", las, int(mlas), b"G", b"G", b"v.ini")]
            for sub, extra in (("filter", []), ("maximal", []), ("layout", ["-o", "G"])):
                argv = [hinge, sub, "--db", "G", "--las", "G" if mlas else "G.las"] + (["--mlas"] if mlas else []) + ["-x", "G", "--config", "v.ini"] + extra
                got.append(subprocess.run(argv, cwd=wd_h, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=penv).returncode)
        if rcs[0] != 0:          # undefined in the reference (e.g. a part without a 5 kb read): both sides must refuse
            ok = got[0] != 0
            return "undefined input, both refuse" if ok else "FAIL: oracle rc %s, executables rc %s" % (rcs, got)
        if rcs != [0, 0, 0] or got != [0, 0, 0]:
            same_refusal = all((a == 0) == (b == 0) for a, b in zip(rcs, got))
            return ("both refuse a later stage %s %s" % (rcs, got)) if same_refusal else "FAIL: oracle rc %s, executables rc %s" % (rcs, got)
        bad = [f for f in FILES if not filecmp.cmp(os.path.join(wd_o, f), os.path.join(wd_h, f), shallow=False)]
        if bad:
            if keep_failures:
                shutil.copytree(tmp, os.path.join(keep_failures, "case%03d" % k))
            return "FAIL: differs in %s" % bad
        return "ok (%d reads, %d overlaps, %d hinges)" % (d.n_reads, d.novl, sum((len(l.split()) - 1) // 2 for l in open(os.path.join(wd_h, "G.hinges.txt"))))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--keep-failures", default="")
    ap.add_argument("--paths", action="store_true", help="also randomise the library's alternative kernel paths, the thread count and FASTA + PAF input")
    args = ap.parse_args()
    import oracle
    lib = oracle.oracle_lib()
    rng = np.random.default_rng(args.seed)
    fails = 0
    for k in range(args.cases):
        spec, filt, lay = random_case(rng)
        env, paf = random_paths(rng, spec) if args.paths else ({}, False)
        res = run_case(k, spec, filt, lay, lib, args.keep_failures, env, paf)
        fails += res.startswith("FAIL")
        print("case %3d: %s%s%s" % (k, res, "  [PAF]" if paf else "", ("  " + " ".join("%s=%s" % kv for kv in sorted(env.items()))) if env else ""), flush=True)
        if res.startswith("FAIL"):
            print("   spec = %r\n   filter ini = %r\n   layout ini = %r\n   env = %r paf = %r" % (spec, filt, lay, env, paf), flush=True)
    print("%d cases, %d failures" % (args.cases, fails))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
