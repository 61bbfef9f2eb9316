#!/usr/bin/env python
"""Timing of the statistics sweep with the median inside it (hinge_filter_stats_median) against the two-launch form
(hinge_filter_stats + hinge_filter_median), HIP events around each kernel, three cold parts rotating.
HINGE_K1M_ABLATE variants switch pieces of the fused epilogue off (measurement only: results are garbage then).
python tools/k1_bench.py [--reps 20] [--ablate 0,1,2,4,8]   (through gpurun)"""
import argparse
import dataclasses
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--parts", type=int, default=3)
    ap.add_argument("--ablate", default="0")
    args = ap.parse_args()
    import torch
    from hinge_amd import capi, synth
    from hinge_amd.config import default_filter_params
    P = default_filter_params()
    dev = torch.device("cuda", 0)
    parts = []
    for k in range(args.parts):
        d = synth.generate(dataclasses.replace(synth.CONFIGS["cfg2_ecoli160"], n_blocks=1, seed=2 + 17 * k))
        pile = synth.to_pileups(d)
        span16, max_pile, in_range = capi.pack_spans(pile.row_ptr, pile.a_span, d.rlen)
        tens = [torch.from_numpy(x).to(dev) for x in (pile.row_ptr, pile.a_span, pile.b_span, pile.b_flag.view(np.int32), span16.view(np.int32))]
        parts.append((d.rlen.copy(), d.n_reads, pile.n_ovl, tens, max_pile, in_range))

    def run(label, fused, hist):
        ctxs = []
        for rlen, n, m, tens, max_pile, in_range in parts:
            ctx = capi.Context(0)
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            ctx.set_reads(rlen, None)
            ctx.set_pileups_packed(0, n - 1, tens[0], tens[1], tens[2], tens[3], tens[4], max_pile, in_range, n_ovl=m, on_device=True)
            ctx.set_min_cov(P.min_cov)
            ctxs.append((ctx, n, torch.zeros(4098, dtype=torch.int32, device=dev)))
        for c, n, h in ctxs:
            c.profile_select(None)
            c.profile_enable(4 * args.reps + 8)
        for _ in range(args.reps):
            for c, n, h in ctxs:
                if fused:
                    c.filter_stats_median(P, hist_dev=h if hist else None)
                else:
                    c.filter_stats(P)
                    c.filter_median(P, 0, n - 1, fetch=False)
        torch.cuda.synchronize()
        tot = {}
        for c, n, h in ctxs:
            for k, (ms, cnt) in c.profile_report().items():
                if cnt:
                    a = tot.setdefault(k, [0.0, 0]); a[0] += ms; a[1] += cnt
            c.close()
        print("%-28s" % label + "  ".join("%s %.1f us" % (k, 1e3 * v[0] / v[1]) for k, v in sorted(tot.items())), flush=True)

    run("two launches", False, False)
    for a in [int(x) for x in args.ablate.split(",")]:
        os.environ["HINGE_K1M_ABLATE"] = str(a)
        run("fused, ablate=%d" % a, True, False)
        if a == 0:
            run("fused, histogram out", True, True)
    os.environ.pop("HINGE_K1M_ABLATE", None)


if __name__ == "__main__":
    main()
