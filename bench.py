#!/usr/bin/env python
"""bench.py - overlaps/s through filter + hinge-detect on the E. coli 160x restatement.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2_ecoli160]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the `hinge filter` hot path (coverage statistics -> median / MIN_COV ->
coverage mask + repeat annotation -> hinge calling) over one batch of synthetic pile-ups that is
already resident in HBM when the timed region starts.  With N > 1 every rank owns one DAZZ_DB block
of the same size (weak scaling: N blocks = an N-times larger merged data set) and the path's three
exchange steps run as RCCL all-gathers between the kernels (hinge_amd/dist.py).

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
PATH_BYTES_PER_OVERLAP = 40    # pass 1 = 8 B, pass 2 = 24 B + one 8 B mask gather
# algorithmic bytes of each kernel per overlap it streams (DESIGN.md "Kernels")
KERNEL_BYTES_PER_OVERLAP = {"k_cov_stats": 8, "k_mask_annotate": 8}
# per-read side traffic of each kernel: row_ptr 8 + rlen 4 + outputs
KERNEL_BYTES_PER_READ = {"k_cov_stats": 8 + 4 + 4 + 4, "k_mask_annotate": 8 + 4 + 8 + 8 + 1 + 4 + 4}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2_ecoli160")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-genome", type=int, default=3_000_000, help="genome length of the CPU-baseline sample")
    return ap.parse_args()


def pmc_traffic(kname):
    """HBM bytes per launch of `kname` from the newest committed PMC summary (profiles/*_pmc_summary.csv).

    The counters come from separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of this same
    command (they cannot share a pass with timing); on gfx950 FETCH_SIZE reports half the bytes of a wide
    coalesced read, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  Returns (None, None) without a file.
    """
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_pmc_summary.csv")))
    if not files:
        return None, None
    fetch = write = None
    with open(files[-1], newline="") as f:
        for row in csv.DictReader(f):
            if not row["kernel"].split("::")[-1].split("<")[0].startswith(kname):
                continue   # k_mask_annotate covers k_mask_annotate_q20 and the general k_mask_annotate<RESO>
            v = float(row.get("mean_value_KB") or row.get("mean_value"))
            if row["counter"] == "FETCH_SIZE":
                fetch = (fetch or 0.0) + v
            elif row["counter"] == "WRITE_SIZE":
                write = (write or 0.0) + v
    if fetch is None or write is None:
        return None, None
    return (2.0 * fetch + write) * 1024.0, "profiles/" + os.path.basename(files[-1]) + " (2*FETCH_SIZE + WRITE_SIZE, KB)"



def cpu_baseline(workload: str, sample_genome: int):
    """Time the CPU oracle (single thread, a statement-level port of filter.cpp) on a bounded sample of
    the same workload: same coverage / read-length / repeat model, smaller genome."""
    import dataclasses

    import oracle
    from hinge_amd import synth

    spec = dataclasses.replace(synth.CONFIGS[workload], genome_len=sample_genome, n_repeat_families=1, repeat_copies=(2, 2), n_blocks=1)
    d = synth.generate(spec)
    tmp = tempfile.mkdtemp(prefix="hinge_cpu_")
    try:
        synth.write_dataset(d, tmp, "G", write_bases=False)
        with open(os.path.join(tmp, "nominal.ini"), "w") as f:
            f.write("[filter]\nlength_threshold = 1000;\naln_threshold = 1000;\nmin_cov = 5;\ncut_off = 300;\ntheta = 300;\n")
        lib = oracle.oracle_lib()
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            t0 = time.perf_counter()
            rc = lib.oracle_filter(b"G", b"G.las", 0, b"G", b"nominal.ini", b"")
            dt = time.perf_counter() - t0
        finally:
            os.chdir(cwd)
        assert rc == 0, "oracle_filter rc=%d" % rc
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {
        "value": d.novl / dt,
        "unit": "overlaps/s",
        "cores": 1,
        "kind": "port",
        "sample": "oracle_filter (CPU restatement of filter.cpp incl. .las parse + text output), %d reads / %d overlaps, "
                  "%s at genome %d bp, %.1f s wall" % (d.n_reads, d.novl, workload, sample_genome, dt),
    }


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    from hinge_amd import capi, synth
    from hinge_amd.config import default_filter_params
    from hinge_amd.dist import BlockTable, Exchange, HipBackend, ShardedFilter

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_pg = world > 1 or ("RANK" in os.environ and os.environ.get("HINGE_FORCE_COLLECTIVES", "0") == "1")
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", device_id=dev)

    # ---- workload: every rank generates its own block (seed offset by rank) ---------------------
    import dataclasses
    spec = dataclasses.replace(synth.CONFIGS[args.workload], n_blocks=1)
    spec = dataclasses.replace(spec, seed=spec.seed + 1000 * rank)
    t_gen = time.perf_counter()
    d = synth.generate(spec)
    pile = synth.to_pileups(d)
    t_gen = time.perf_counter() - t_gen

    # block table over ranks (block sizes differ slightly only through the seed)
    sizes = [d.n_reads]
    if use_pg:
        t = torch.tensor([d.n_reads], dtype=torch.int64, device=dev)
        out = torch.empty(world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(out, t)
        sizes = [int(v) for v in out.cpu().tolist()]
    # Every block gets the same number of read ids (the largest block's; the extra ids are reads of length 0 without
    # overlaps): the per-read tables are then exchanged by ONE in-place all-gather.  Only the id space is padded.
    S = max(sizes) + int(os.environ.get("HINGE_BENCH_PAD", "0"))   # HINGE_BENCH_PAD: exercise the padding with one rank
    first = [k * S for k in range(world + 1)]
    blocks = BlockTable(first)
    lo = first[rank]
    hi = lo + d.n_reads          # real reads of this rank: [lo, hi); ids [hi, lo + S) are padding
    n_total = first[-1]

    # global-id views of this rank's block: rlen table of ALL reads, row_ptr with empty rows elsewhere
    rlen_all = np.zeros(n_total, np.int32)
    if use_pg:
        t = torch.zeros(n_total, dtype=torch.int32, device=dev)
        t[lo:hi] = torch.from_numpy(d.rlen).to(dev)
        dist.all_reduce(t)
        rlen_all = t.cpu().numpy()
    else:
        rlen_all[lo:hi] = d.rlen
    row_ptr = np.zeros(n_total + 1, np.int64)
    row_ptr[lo:hi + 1] = pile.row_ptr
    row_ptr[hi + 1:] = pile.row_ptr[-1]
    b_flag = pile.b_flag.copy()
    b_flag = ((b_flag & np.uint32(0x7FFFFFFF)) + np.uint32(lo)) | (b_flag & np.uint32(0x80000000))

    t_row = torch.from_numpy(row_ptr).to(dev)
    t_a = torch.from_numpy(pile.a_span).to(dev)
    t_b = torch.from_numpy(pile.b_span).to(dev)
    t_f = torch.from_numpy(b_flag.view(np.int32)).to(dev)
    n_ovl = pile.n_ovl

    P = default_filter_params()
    ctx = capi.Context(local_rank)
    backend = HipBackend(ctx, P, rlen_all, None, lo, lo + S - 1, t_row, t_a, t_b, t_f)
    xch = Exchange(blocks, dev)
    job = ShardedFilter(backend, xch, mode="merged")

    def sync():
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warmup (also sizes the library's annotation buffers) ------------------------------------
    ctx.filter_stats(P)
    if use_pg:
        xch.all_gather_rows(job.mean_cov)
    ctx.filter_median(P, 0, n_total - 1, fetch=True)
    ctx.filter_mask_annotate(P)          # synchronous variant: regrows the annotation buffer if needed
    if use_pg:
        xch.all_gather_rows(job.mask)
    ctx.filter_hinges(P)                 # synchronous variant: regrows the exact-path buffers if needed
    for _ in range(args.warmup):
        job.step(fetch_hinges=False)
    ctx.check()

    # ---- untimed breakdown pass: events around EVERY kernel (they cost ~60 us of stream time per step, so the
    # timed region below brackets only the kernel it prices) ------------------------------------------------
    n_break = max(3, min(args.steps, 10))
    ctx.profile_select(None)
    ctx.profile_enable(10 * n_break + 16)
    for _ in range(n_break):
        job.step(fetch_hinges=False)
    sync()
    ctx.check()
    breakdown = ctx.profile_report()
    ctx.profile_enable(0)
    dominant = max(((k, v) for k, v in breakdown.items() if v[1] > 0), key=lambda kv: kv[1][0])[0]

    # ---- timed region: K steps, HIP events around the dominant kernel only ---------------------------------
    ctx.profile_select([dominant])
    ctx.profile_enable(2 * args.steps + 16)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        job.step(fetch_hinges=False)
    sync()
    elapsed = time.perf_counter() - t0
    ctx.check()
    prof = ctx.profile_report()
    ctx.profile_enable(0)
    ctx.profile_select(None)
    hinges = job.step(fetch_hinges=True)     # exchange 3, outside the timed region: (read, pos, type) rows
    n_hinges = int(hinges.shape[0])
    counters = ctx.counters()

    if use_pg:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        t = torch.tensor([n_ovl], dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        total_ovl = int(t.item())
    else:
        total_ovl = n_ovl

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = total_ovl * args.steps / elapsed
        # dominant kernel of rank 0 by total HIP-event time
        kname, (kms, kcnt) = max(((k, v) for k, v in prof.items() if v[1] > 0), key=lambda kv: kv[1][0])
        avg_ms = kms / kcnt
        if kname not in KERNEL_BYTES_PER_OVERLAP:
            alg_bytes = None   # sparse kernel: touches only the work-list reads; no per-launch algorithmic figure
            achieved = None
        else:
            alg_bytes = KERNEL_BYTES_PER_OVERLAP[kname] * n_ovl + KERNEL_BYTES_PER_READ[kname] * (hi - lo)
            achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(kname)
        roofline = {
            "bound": "hbm",
            "kernel": kname,
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": (achieved / HBM_PEAK_GBS) if achieved is not None else None,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "avg_launch_ms": avg_ms,
            "algorithmic_bytes_per_launch": alg_bytes,
            "kernels_ms_per_step": {k: v[0] / n_break for k, v in breakdown.items() if v[1] > 0},
            "kernels_ms_note": "untimed breakdown pass with events around every kernel (includes event overhead); avg_launch_ms is from the timed region",
            "path_bytes_per_overlap": PATH_BYTES_PER_OVERLAP,
            "path_achieved_GBs": PATH_BYTES_PER_OVERLAP * n_ovl / (ms_per_step * 1e-3) / 1e9,
        }
        cpu = None
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N = 1 only
            cpu = cpu_baseline(args.workload, args.cpu_sample_genome)
        out = {
            "metric": "overlaps/sec through filter+hinge-detect, E. coli 160x",
            "value": value,
            "unit": "overlaps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {
                "workload": "%s: synthetic restatement of E. coli P6-C4 160x (G=4.6 Mb, lognormal reads mean 8.5 kb, "
                            "7 x 5 kb repeat copies), one such block per GPU" % args.workload,
                "reads_per_gpu": int(hi - lo),
                "overlaps_per_gpu": int(n_ovl),
                "total_overlaps": int(total_ovl),
                "parallelism": "shard-by-block x%d, merged-las semantics; per step one 16 KiB all-reduce (coverage histogram) + one all-gather (masks, 8 B per read)" % world,
                "hinges_found": n_hinges,
                "reads_in_hinge_pass": int(counters[0]),
                "annotations_on_exact_path": int(counters[1]),
                "generate_s": t_gen,
            },
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
