#!/usr/bin/env python
"""Timing of K4 (ProcessAlignment = trim_overlap + AddTypesAsymmetric for EVERY overlap of a part: what `hinge maximal`
classifies) on the bench workload, both kernels: the streaming form (one lane per overlap over an LDS-staged .las; default) and
the eight-lanes-per-overlap form (HINGE_K4_ROWS=1).  The device holds what `hinge maximal` uploads: the SoA columns and the raw
.las image the trace offsets point into.  Roofline as SURVEY.md 8(d) defines it for this kernel: 24 B of record + tlen trace
bytes per classified overlap (+ the 8-byte eff[B] gather).      python tools/k4_bench.py [--genome 4600000] [--reps 5]"""
import argparse
import dataclasses
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=int, default=4_600_000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--json", action="store_true", help="only the form `hinge maximal` runs (+ the column form beside it), one JSON line (bench.py's `maximal_kernel` block)")
    args = ap.parse_args()
    from hinge_amd import capi, synth
    from hinge_amd.config import default_filter_params
    spec = dataclasses.replace(synth.CONFIGS["cfg2_ecoli160"], genome_len=args.genome, n_blocks=1)
    d = synth.generate(spec)
    pile = synth.to_pileups(d)
    assert pile.n_ovl == d.novl, "the bench data set has no self-overlaps"
    tmp = tempfile.mkdtemp(prefix="hinge_k4_")
    path = os.path.join(tmp, "G.las")
    synth.write_las_file(d, path)
    raw = np.fromfile(path, dtype=np.uint8)
    os.remove(path)
    os.rmdir(tmp)
    ts = spec.tspace
    nseg = ((d.ae.astype(np.int64) + ts - 1) // ts - d.ab.astype(np.int64) // ts)
    tlen = (2 * nseg).astype(np.int32)
    rec_start = 12 + np.concatenate([[0], np.cumsum(40 + tlen.astype(np.int64))[:-1]])
    toff = (rec_start + 40).astype(np.int64)
    assert int(rec_start[-1] + 40 + tlen[-1]) == len(raw)
    P = default_filter_params()
    ctx = capi.Context(0)
    ctx.set_reads(d.rlen, None)
    span16, max_pile, in_range = capi.pack_spans(pile.row_ptr, pile.a_span, d.rlen)
    ctx.set_pileups_packed(0, d.n_reads - 1, pile.row_ptr, pile.a_span, pile.b_span, pile.b_flag, span16, max_pile, in_range)
    ctx.set_min_cov(P.min_cov)
    ctx.filter_stats(P)
    ctx.filter_median(P, 0, d.n_reads - 1, fetch=True)
    ctx.filter_mask_annotate(P)
    eff = ctx.get_masks()[0]                       # the masks `hinge filter` writes to .mas
    ctx.set_traces(raw, toff, tlen, 1)
    ctx.set_eff_reads(eff)
    n = pile.n_ovl
    alg = 24.0 * n + float(tlen.sum()) + 8.0 * n
    # the form `hinge maximal` uses since round 5: the .las image + one 32-bit offset per overlap (hinge_set_las_image)
    cimg = capi.Context(0)
    cimg.set_reads(d.rlen, None)
    cimg.set_pileups(0, d.n_reads - 1, pile.row_ptr, pile.a_span, pile.b_span, pile.b_flag)
    from hinge_amd import formats
    win_base, rec_rel = formats.image_windows(rec_start, 40 + tlen.astype(np.int64))
    cimg.set_las_image(raw, win_base, rec_rel, 1)
    cimg.set_eff_reads(eff)
    out = {}
    timing = {}
    variants = (("image", "img"), ("stream", None)) if args.json else (("image", "img"), ("image 12 w/CU", "imgwpc12"), ("image 8 w/CU", "imgwpc8"), ("image cap 8192", "imgcap8192"), ("image cap 12288", "imgcap12288"),
                      ("stream", None), ("stream 16 w/CU", "wpc16"), ("stream 32 w/CU", "wpc32"), ("stream cap 8192", "cap8192"), ("rows", "1"))
    for name, env in variants:
        use = ctx
        if env and env.startswith("img"):
            use, env = cimg, (env[3:] or None)
        os.environ.pop("HINGE_K4_ROWS", None)
        os.environ.pop("HINGE_K4_CAP", None)
        os.environ.pop("HINGE_K4_WAVES_PER_CU", None)
        if env and env.startswith("wpc"):
            os.environ["HINGE_K4_WAVES_PER_CU"] = env[3:]
        elif env == "1":
            os.environ["HINGE_K4_ROWS"] = env
        elif env:
            os.environ["HINGE_K4_CAP"] = env[3:]
        types = use.trim_classify_part(n, 1000, 300, 0)
        use.profile_select(["k_trim_classify"])
        use.profile_enable(2 * args.reps + 4)
        for _ in range(args.reps):
            use.trim_classify_part(n, 1000, 300, 0)
        ms, cnt = use.profile_report()["k_trim_classify"]
        use.profile_enable(0)
        out[name] = types
        t = ms / cnt
        timing[name] = t
        if not args.json:
            print("%-16s %.3f ms per launch, %d overlaps, mean tlen %.1f B, algorithmic %.2f GB -> %.2f TB/s = %.2f of the HBM peak" %
              (name, t, n, float(tlen.mean()), alg / 1e9, alg / (t * 1e-3) / 1e12, alg / (t * 1e-3) / 8e12), flush=True)
    os.environ.pop("HINGE_K4_ROWS", None)
    os.environ.pop("HINGE_K4_CAP", None)
    os.environ.pop("HINGE_K4_WAVES_PER_CU", None)
    assert os.environ.get("K4_NOASSERT") or all(np.array_equal(out["stream"], v) for v in out.values()), "the kernels disagree"   # (K4_NOASSERT: ablation builds)
    if args.json:
        import json
        t = timing["image"]
        print(json.dumps({"kernel": "k_trim_classify_image", "what": "ProcessAlignment (trim_overlap + AddTypesAsymmetric) of EVERY overlap of the bench part: the kernel of `hinge maximal`, "
                          "the .las image resident in HBM (hinge_set_las_image); HIP events around the launches, %d repetitions" % args.reps,
                          "overlaps": int(n), "mean_trace_bytes": float(tlen.mean()), "ms_per_launch": t, "overlaps_per_s": n / (t * 1e-3),
                          "roofline": {"bound": "hbm", "achieved": alg / (t * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": alg / (t * 1e-3) / 8e12,
                                       "algorithmic_bytes_per_launch": alg, "algorithmic_bytes_per_overlap": alg / n,
                                       "note": "SURVEY 8(d): 24 B of record fields + the trace bytes + one 8-byte eff[B] gather per overlap; counter traffic of the same "
                                               "launch: profiles/*_stages_k4_pmc_summary.csv (2 * FETCH_SIZE + WRITE_SIZE = 3.75 GB: 1.14 x)"},
                          "column_form_ms_per_launch": timing["stream"], "types_identical_to_the_column_form": True}))
        return
    print("types identical; histogram:", np.bincount(out["stream"], minlength=14).tolist())


if __name__ == "__main__":
    main()
