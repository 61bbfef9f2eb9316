#!/usr/bin/env python
"""Randomised differential test of `hinge consensus`: random generator settings (contig count and lengths, coverage, read lengths,
substitution / insertion / deletion rates, planted draft errors, low-coverage windows, unaligned flanks, short alignments,
duplicate B reads, empty contigs, trace spacing 50 / 100 / 200 = one- and two-byte traces, min_length) - the GPU executable
against the reference's own program (oracle/_ref/consensus) where it was built, else the oracle restatement pinned to it:
FASTA and stdout byte for byte.      python tools/fuzz_consensus.py --cases 60 --seed 1
"""
import argparse
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def random_spec(rng):
    from hinge_amd import synth_consensus as sc
    lo = int(rng.integers(1_500, 20_000))
    rl = int(rng.integers(500, 3_000))
    noise = float(rng.choice([0.0, 0.02, 0.05, 0.08]))
    return sc.ConsensusSpec(
        n_contigs=int(rng.integers(1, 5)), contig_len=(lo, lo + int(rng.integers(0, 20_000))), coverage=float(rng.choice([2.5, 6, 12, 25, 45])),
        read_len=(rl, rl + int(rng.integers(200, 9_000))), p_sub=noise * float(rng.random()), p_ins=noise * float(rng.random()) * 1.5,
        p_del=noise * float(rng.random()), draft_errors_per_kb=float(rng.choice([0.0, 1.0, 4.0])), p_carry=float(rng.choice([0.6, 0.92, 1.0])),
        low_cov_windows=int(rng.integers(0, 4)), flank_max=int(rng.choice([0, 15, 120])), short_alignments=int(rng.integers(0, 4)),
        duplicate_b=int(rng.integers(0, 3)), empty_contigs=int(rng.integers(0, 2)), tspace=int(rng.choice([50, 100, 100, 200])),
        seed=int(rng.integers(1, 1 << 30)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import consensus_common as cc
    import oracle
    from hinge_amd import synth_consensus as sc
    lib = oracle.oracle_lib()
    rng = np.random.default_rng(args.seed)
    fails = 0
    for k in range(args.cases):
        spec = random_spec(rng)
        min_len = int(rng.choice([0, 300, 500, 2000]))
        tmp = tempfile.mkdtemp(prefix="hinge_cfuzz_")
        try:
            try:
                d = sc.generate(spec)
                sc.write_dataset(d, tmp, min_length=min_len)
            except (AssertionError, ValueError) as ex:
                print("case %3d: skipped (generator: %s)" % (k, str(ex)[:60]), flush=True)
                continue
            if d.n_alignments == 0:
                print("case %3d: skipped (no alignment)" % k, flush=True)
                continue
            ref = cc.run_reference(tmp) or cc.run_oracle(lib, tmp)
            got = cc.run_product(tmp)
            ok = ref[0] == got[0] and ref[1] == got[1]
            fails += not ok
            print("case %3d: %s (%d contigs, %d alignments, tspace %d, min_length %d)" % (k, "ok" if ok else "FAIL fasta=%s stdout=%s" % (ref[0] == got[0], ref[1] == got[1]),
                                                                                        len(d.contigs), d.n_alignments, spec.tspace, min_len), flush=True)
            if not ok:
                print("   spec = %r" % (spec,), flush=True)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    print("%d cases, %d failures" % (args.cases, fails))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
