#!/usr/bin/env python
"""A/B timing of the K2 kernels (coverage mask + repeat annotation) on the bench workload: every variant runs on its own
context over the same resident part, its masks / annotations are compared with the first variant's, and K2 alone is timed
with HIP events (hinge_profile_*).  Variants are environment settings read when a context is created
(HINGE_K2_WGS, HINGE_NO_SPAN16 ...).    python tools/k2_bench.py [--genome 4600000] [--cov-out]"""
import argparse
import dataclasses
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = [
    ("default", {}),
    ("heavy=0", {"HINGE_K2_HEAVY": "0"}),
    ("heavy=1", {"HINGE_K2_HEAVY": "1"}),
    ("wgs=5/cu", {"HINGE_K2_WGS": "1280"}),
    ("wgs=6/cu", {"HINGE_K2_WGS": "1536"}),
    ("wgs=7/cu", {"HINGE_K2_WGS": "1792"}),
    ("order=64", {"HINGE_K2_ORDER_BP": "64"}),
    ("order=4096", {"HINGE_K2_ORDER_BP": "4096"}),
    ("order=none", {"HINGE_K2_ORDER_BP": "1000000"}),
    ("int32 spans", {"HINGE_NO_SPAN16": "1"}),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=int, default=4_600_000)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--cov-out", action="store_true")
    ap.add_argument("--only", default="", help="comma-separated variant names (default: all)")
    ap.add_argument("--no-check", action="store_true", help="do not compare results between variants (ablation builds produce garbage)")
    ap.add_argument("--parts", type=int, default=3, help="distinct parts rotated between timed launches (cold Infinity Cache)")
    args = ap.parse_args()
    import torch
    from hinge_amd import capi, synth
    from hinge_amd.config import default_filter_params
    P = default_filter_params()
    dev = torch.device("cuda", 0)
    parts = []
    for k in range(args.parts):
        spec = dataclasses.replace(synth.CONFIGS["cfg2_ecoli160"], genome_len=args.genome, n_blocks=1, seed=2 + 17 * k)
        d = synth.generate(spec)
        pile = synth.to_pileups(d)
        span16, max_pile, in_range = capi.pack_spans(pile.row_ptr, pile.a_span, d.rlen)
        tens = [torch.from_numpy(x).to(dev) for x in (pile.row_ptr, pile.a_span, pile.b_span, pile.b_flag.view(np.int32), span16.view(np.int32))]
        parts.append((d.rlen.copy(), d.n_reads, pile.n_ovl, tens, max_pile, in_range))
    base = None
    only = [x.strip() for x in args.only.split(",") if x.strip()]
    for name, env in VARIANTS:
        if only and name not in only:
            continue
        for k in ("HINGE_K2_ORDER_BP", "HINGE_K2_WGS", "HINGE_K2_HEAVY", "HINGE_K2_DEAL", "HINGE_NO_SPAN16", "HINGE_K2_ABLATE", "HINGE_K1_W8"):
            os.environ.pop(k, None)
        os.environ.update(env)
        ctxs = []
        for rlen, n, m, tens, max_pile, in_range in parts:
            ctx = capi.Context(0)
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            ctx.set_reads(rlen, None)
            ctx.set_pileups_packed(0, n - 1, tens[0], tens[1], tens[2], tens[3], tens[4], max_pile, in_range, n_ovl=m, on_device=True)
            ctx.coverage_out(args.cov_out)
            ctx.set_min_cov(P.min_cov)
            ctx.filter_stats(P)
            ctx.filter_median(P, 0, n - 1, fetch=True)
            ctx.filter_mask_annotate(P)      # synchronous: sizes the annotation buffer
            ctxs.append(ctx)
        ablated = args.no_check
        res = None if ablated else [(c.get_masks(), c.get_annotations()[:3]) for c in ctxs]
        if ablated:
            pass
        elif base is None:
            base = res
        else:
            for (m0, a0), (m1, a1) in zip(base, res):
                assert all(np.array_equal(x, y) for x, y in zip(m0, m1)), name + ": masks differ"
                assert all(np.array_equal(x, y) for x, y in zip(a0, a1)), name + ": annotations differ"
        for c in ctxs:
            c.profile_select(["k_mask_annotate", "k_cov_stats"])
            c.profile_enable(4 * args.reps + 8)
        for _ in range(args.reps):
            for c in ctxs:
                c.filter_stats(P)
                c.filter_mask_annotate_async(P)
        tot = {}
        for c in ctxs:
            for k, (ms, cnt) in c.profile_report().items():
                if cnt:
                    a = tot.setdefault(k, [0.0, 0]); a[0] += ms; a[1] += cnt
            c.check()
            c.close()
        n_ovl = sum(p[2] for p in parts) / len(parts)
        line = "%-12s" % name + "  ".join("%s %.1f us (%.2f TB/s at 8 B/ovl)" % (k, 1e3 * v[0] / v[1], 8 * n_ovl / (v[0] / v[1] * 1e-3) / 1e12) for k, v in sorted(tot.items()))
        print(line, flush=True)


if __name__ == "__main__":
    main()
