#!/usr/bin/env python
"""Can the statistics sweep of one part (k_cov_stats: HBM-bound, no LDS) run UNDER the mask / annotate kernel of another
(k_mask_annotate_q20: issue-bound, ~30 % of the HBM peak)?  Four resident parts, two HIP streams: stream M runs K1 (+ median) part
after part, stream S runs K2 of part p as soon as its median is there; against the same kernels on one stream.
python tools/probes/overlap_probe.py [--wgs N]   (through gpurun; HINGE_K2_WGS is read when a context is created)"""
import argparse, dataclasses, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--wgs", default="0,1792,1536,1280")
    args = ap.parse_args()
    import torch
    from hinge_amd import capi, synth
    from hinge_amd.config import default_filter_params
    P = default_filter_params()
    dev = torch.device("cuda", 0)
    parts = []
    for k in range(4):
        d = synth.generate(dataclasses.replace(synth.CONFIGS["cfg2_ecoli160"], n_blocks=1, seed=2 + 17 * k))
        pile = synth.to_pileups(d)
        span16, max_pile, in_range = capi.pack_spans(pile.row_ptr, pile.a_span, d.rlen)
        tens = [torch.from_numpy(x).to(dev) for x in (pile.row_ptr, pile.a_span, pile.b_span, pile.b_flag.view(np.int32), span16.view(np.int32))]
        parts.append((d.rlen.copy(), d.n_reads, pile.n_ovl, tens, max_pile, in_range))
    sM, sS = torch.cuda.Stream(), torch.cuda.Stream()
    for wgs in [int(x) for x in args.wgs.split(",")]:
        if wgs:
            os.environ["HINGE_K2_WGS"] = str(wgs)
        else:
            os.environ.pop("HINGE_K2_WGS", None)
        ctxs = []
        for rlen, n, m, tens, max_pile, in_range in parts:
            c = capi.Context(0)
            c.set_stream(sM.cuda_stream)
            c.set_reads(rlen, None)
            c.set_pileups_packed(0, n - 1, tens[0], tens[1], tens[2], tens[3], tens[4], max_pile, in_range, n_ovl=m, on_device=True)
            c.coverage_out(True)
            c.set_min_cov(P.min_cov)
            c.filter_stats_median(P, fetch=True)
            c.filter_mask_annotate(P)
            ctxs.append((c, n))

        def sequential():
            for c, n in ctxs:
                c.set_stream(sM.cuda_stream)
                c.filter_stats(P)
                c.filter_median(P, 0, n - 1, fetch=False)
            for c, n in ctxs:
                c.filter_mask_annotate_async(P)

        def overlapped():
            evs = []
            for c, n in ctxs:
                c.set_stream(sM.cuda_stream)
                c.filter_stats(P)
                c.filter_median(P, 0, n - 1, fetch=False)
                e = torch.cuda.Event()
                e.record(sM)
                evs.append(e)
            for (c, n), e in zip(ctxs, evs):
                sS.wait_event(e)
                c.set_stream(sS.cuda_stream)
                c.filter_mask_annotate_async(P)
            sM.wait_stream(sS)

        for name, fn in (("one stream", sequential), ("two streams", overlapped)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                fn()
            torch.cuda.synchronize()
            print("K2 workgroups %-5s %-12s %.1f us per four parts (K1 + median + K2)" % (wgs or "all", name, 1e6 * (time.perf_counter() - t0) / args.reps), flush=True)
        for c, n in ctxs:
            c.set_stream(sM.cuda_stream)
            c.check()
            c.close()


if __name__ == "__main__":
    main()
