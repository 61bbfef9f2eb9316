#!/usr/bin/env python
"""Kernel times of one pass over BASELINE config 5's share (1.3e8 overlaps generated on the device, hinge_amd/synth_device.py):
the same kernels as bench.py on a part five times as large.    python tools/cfg5_bench.py [--genome 52000000]"""
import argparse
import dataclasses
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=int, default=52_000_000)
    ap.add_argument("--reps", type=int, default=6)
    args = ap.parse_args()
    import torch
    from hinge_amd import capi, synth, synth_device
    from hinge_amd.config import default_filter_params
    dev = torch.device("cuda", 0)
    spec = dataclasses.replace(synth.CONFIGS["cfg5_share"], genome_len=args.genome)
    p = synth_device.generate_pileups(spec, dev, span16_pad=capi.span16_pad())
    n, m = p.n_reads, p.n_ovl
    P = default_filter_params()
    ctx = capi.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_reads(p.rlen, None)
    ctx.set_pileups_packed(0, n - 1, p.row_ptr, p.a_span, p.b_span, p.b_flag, p.span16, p.max_pile, p.spans_in_range, n_ovl=m, on_device=True)
    ctx.coverage_out(True)
    ctx.set_min_cov(P.min_cov)
    ctx.filter_stats(P)
    ctx.filter_median(P, 0, n - 1, fetch=True)
    ctx.filter_mask_annotate(P)
    ctx.filter_hinges(P)
    junk = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    ctx.profile_select(None)
    ctx.profile_enable(16 * args.reps + 16)
    for _ in range(args.reps):
        junk.fill_(1)                      # 1 GiB through the caches between two passes
        ctx.filter_stats(P)
        ctx.filter_median(P, 0, n - 1, fetch=False)
        ctx.filter_mask_annotate_async(P)
        ctx.filter_hinges_async(P)
    torch.cuda.synchronize()
    ctx.check()
    print("config 5 share: %d reads, %d overlaps (%.0f MB span copy)" % (n, m, 4 * m / 1e6))
    for k, (ms, cnt) in sorted(ctx.profile_report().items()):
        if cnt:
            us = 1e3 * ms / cnt
            extra = ""
            if k == "k_mask_annotate":
                extra = "  -> %.2f TB/s at 8 B per overlap + 37 B per read = %.2f of the HBM peak" % ((8 * m + 37 * n) / (us * 1e-6) / 1e12, (8 * m + 37 * n) / (us * 1e-6) / 8e12)
            if k == "k_cov_stats":
                extra = "  -> %.2f TB/s real (4 B per overlap), %.2f TB/s at 8 B" % (4 * m / (us * 1e-6) / 1e12, 8 * m / (us * 1e-6) / 1e12)
            print("  %-18s %9.1f us%s" % (k, us, extra))
    ctx.close()


if __name__ == "__main__":
    main()
