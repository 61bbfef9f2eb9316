#!/usr/bin/env python
"""Per-read time stamps of k_mask_annotate_q20 (an ablation build of the library, -DHINGE_ABLATE, given by HINGE_LIB): how long a
wavefront spends on one read and in which phase, and how many reads are in flight over the life of the launch (ramp, steady state,
tail).    HINGE_LIB=.../libhinge_hip_ablate.so python tools/k2_trace.py [--wgs 1792]"""
import argparse
import ctypes
import dataclasses
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=int, default=4_600_000)
    ap.add_argument("--wgs", default="")
    ap.add_argument("--cov-out", action="store_true", default=True)
    args = ap.parse_args()
    if args.wgs:
        os.environ["HINGE_K2_WGS"] = args.wgs
    import torch
    from hinge_amd import capi, synth
    from hinge_amd.config import default_filter_params
    lib = ctypes.CDLL(capi.LIB_PATH)
    P = default_filter_params()
    dev = torch.device("cuda", 0)
    spec = dataclasses.replace(synth.CONFIGS["cfg2_ecoli160"], genome_len=args.genome, n_blocks=1, seed=2)
    d = synth.generate(spec)
    pile = synth.to_pileups(d)
    span16, max_pile, in_range = capi.pack_spans(pile.row_ptr, pile.a_span, d.rlen)
    tens = [torch.from_numpy(x).to(dev) for x in (pile.row_ptr, pile.a_span, pile.b_span, pile.b_flag.view(np.int32), span16.view(np.int32))]
    n = d.n_reads
    ctx = capi.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_reads(d.rlen.copy(), None)
    ctx.set_pileups_packed(0, n - 1, tens[0], tens[1], tens[2], tens[3], tens[4], max_pile, in_range, n_ovl=pile.n_ovl, on_device=True)
    ctx.coverage_out(args.cov_out)
    ctx.set_min_cov(P.min_cov)
    ctx.filter_stats(P)
    ctx.filter_median(P, 0, n - 1, fetch=True)
    ctx.filter_mask_annotate(P)
    for _ in range(3):
        ctx.filter_stats(P)
        ctx.filter_mask_annotate_async(P)
    torch.cuda.synchronize()
    # evict: stream something else through the caches
    junk = torch.empty(1 << 30, dtype=torch.uint8, device=dev); junk.fill_(1); torch.cuda.synchronize()
    ctx.filter_stats(P)
    torch.cuda.synchronize()
    lib.hinge_debug_k2_trace_begin.argtypes = [ctypes.c_longlong]
    lib.hinge_debug_k2_trace_end.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
    assert lib.hinge_debug_k2_trace_begin(n) == 0
    ctx.filter_mask_annotate_async(P)
    torch.cuda.synchronize()
    out = np.zeros((n, 5), np.uint64)
    assert lib.hinge_debug_k2_trace_end(out.ctypes.data, n) == 0
    ctx.check(); ctx.close()
    items = np.nonzero(out[:, 4] > 0)[0]
    t = out[out[:, 4] > 0].astype(np.int64)
    t0 = t[:, 0].min()
    t = (t - t0) * 10e-3          # us (100 MHz)
    span = t[:, 4].max()
    dur = t[:, 4] - t[:, 0]
    print("reads traced %d of %d; launch span (first read start to last read end) %.1f us" % (len(t), n, span))
    print("per read: %.2f us mean, %.2f median, %.2f p90, %.2f max" % (dur.mean(), np.median(dur), np.percentile(dur, 90), dur.max()))
    names = ["rows + histogram", "fold + scan", "mask pass", "gate / annotate / outputs"]
    for k in range(4):
        ph = t[:, k + 1] - t[:, k]
        print("  %-28s %.2f us mean (%.0f %%), p90 %.2f" % (names[k], ph.mean(), 100 * ph.sum() / dur.sum(), np.percentile(ph, 90)))
    grid = np.linspace(0, span, 41)
    infl = [(int(((t[:, 0] <= x) & (t[:, 4] > x)).sum())) for x in grid]
    print("reads in flight over the launch (40 steps of %.1f us):" % (span / 40))
    print("  " + " ".join(str(v) for v in infl))
    done = np.sort(t[:, 4])
    for q in (50, 90, 95, 99, 100):
        print("  %3d %% of the reads done at %.1f us" % (q, done[min(len(done) - 1, int(len(done) * q / 100))] if q < 100 else done[-1]))
    print("the last reads to end (list item: the two- and four-slot reads are the last %d or so items of the list; start, duration, histogram phase, us):" % int((d.rlen > 19000).sum()))
    for k in np.argsort(t[:, 4])[-12:]:
        print("    item %6d  start %5.1f  duration %5.1f  histogram %5.1f  end %5.1f" % (items[k], t[k, 0], dur[k], t[k, 1] - t[k, 0], t[k, 4]))
    # when each of the 64 item counters ("heads": list positions h, h + 64, ...) runs dry, and each XCD's group of eight
    n1 = int((d.rlen <= 19000).sum())
    one = items < n1
    head_end = np.array([t[one & (items % 64 == h), 4].max() for h in range(64)])
    print("last read of a head ends at: min %.1f, median %.1f, max %.1f us; by group of eight heads (h %% 8): %s" % (
        head_end.min(), np.median(head_end), head_end.max(), " ".join("%.1f" % head_end[g::8].max() for g in range(8))))
    head_sum = np.array([dur[one & (items % 64 == h)].sum() for h in range(64)])
    print("sum of a head's read times: min %.0f, max %.0f us (/128 wavefronts: %.1f .. %.1f us)" % (head_sum.min(), head_sum.max(), head_sum.min() / 128, head_sum.max() / 128))
    started = np.sort(t[:, 0])
    print("  first start of the last 10 %% of reads: %.1f us; reads started in the first 2 us: %d" % (started[int(len(started) * 0.9)], int((started < 2).sum())))


if __name__ == "__main__":
    main()
