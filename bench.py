#!/usr/bin/env python
"""bench.py - overlaps/s through filter + hinge-detect on the E. coli 160x restatement.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--parts R] [--no-e2e] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one whole pass of the `hinge filter` hot path over one batch = R (default 4) DISTINCT E. coli 160x read sets
(different generator seeds), each ~26 M overlaps, resident in HBM in the layout the ingest produces (int32 columns + the 16|16
span copy + the two pile-up facts), processed back to back.  Every per-part device kernel is inside the step:
coverage statistics -> median / MIN_COV -> coverage mask + repeat annotation (which also stores the `.coverage.txt` bins) ->
hinge calling.  R parts rotate so that nothing a pass reads was left in the 256 MiB Infinity Cache by the previous pass over the
same part (R x ~0.4 GB are streamed between two visits); within one pass the second sweep may hit what the first one brought in,
exactly as it does in production.  With N > 1 every rank owns one DAZZ_DB block of every part (weak scaling) and the path's
exchange steps run as RCCL collectives between the kernels, enqueued asynchronously so that a part's exchanges run under the next
part's kernels (hinge_amd/dist.py, step_pipelined).

At N = 1 rank 0 also reports
  * "e2e": `hinge filter / maximal / layout` (the C++ executables over libhinge_hip, .las ingest and text output included) on the
    part-0 data set written to disk, next to the single-thread CPU oracle on the same files, every output compared byte for byte;
  * "cpu_baseline": the oracle's filter stage of that same run (the full configuration, not a sample).

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import dataclasses
import filecmp
import json
import os
import shutil
import subprocess
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
INI = ("[filter]\nlength_threshold = 1000;\naln_threshold = 1000;\nmin_cov = 5;\ncut_off = 300;\ntheta = 300;\n"
       "[layout]\nhinge_slack = 1000\nmin_connected_component_size = 8\n")
E2E_FILES = [".mas", ".cmas", ".repeat.txt", ".hinges.txt", ".coverage.txt", ".max", ".contained.txt", ".edges.hinges", ".hinge.list", ".deadends.txt"]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2_ecoli160")
    ap.add_argument("--parts", type=int, default=4, help="distinct resident read sets a step passes over (>= 3 keeps the Infinity Cache cold)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    return ap.parse_args()


def pmc_traffic(kname):
    """HBM bytes per launch of `kname` from the newest committed PMC summary (profiles/*_pmc_summary.csv).

    The counters come from separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of this same
    command (they cannot share a pass with timing); on gfx950 FETCH_SIZE reports half the bytes of a wide
    coalesced read, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  Returns (None, None) without a file.
    """
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_rocprofv3_pmc_summary.csv")))
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


def count_pairs(path, skip_last_line=False):
    n = 0
    lines = open(path).read().splitlines()
    for l in (lines[:-1] if skip_last_line else lines):
        n += (len(l.split()) - 1) // 2
    return n


def end_to_end(d, workload):
    """filter / maximal / layout on disk: CPU oracle (1 thread) vs the executables, outputs compared byte for byte.
    Returns (e2e block, cpu_baseline block, hinges in the oracle's .hinges.txt)."""
    import oracle
    from hinge_amd import synth

    tmp = tempfile.mkdtemp(prefix="hinge_e2e_")
    try:
        t0 = time.perf_counter()
        synth.write_dataset(d, tmp, "G", write_bases=False)
        with open(os.path.join(tmp, "nominal.ini"), "w") as f:
            f.write(INI)
        t_write = time.perf_counter() - t0
        las_bytes = os.path.getsize(os.path.join(tmp, "G.las"))
        lib = oracle.oracle_lib()
        cwd = os.getcwd()
        os.chdir(tmp)
        t = {}
        try:   # the oracle writes with prefix O, the executables with prefix H: one copy of the 3 GB .las serves both
            t0 = time.perf_counter(); rc = lib.oracle_filter(b"G", b"G.las", 0, b"O", b"nominal.ini", b""); t["filter"] = time.perf_counter() - t0; assert rc == 0, rc
            t0 = time.perf_counter(); rc = lib.oracle_maximal(b"G", b"G.las", 0, b"O", b"nominal.ini"); t["maximal"] = time.perf_counter() - t0; assert rc == 0, rc
            t0 = time.perf_counter(); rc = lib.oracle_layout(b"G", b"G.las", 0, b"O", b"O", b"nominal.ini"); t["layout"] = time.perf_counter() - t0; assert rc == 0, rc
        finally:
            os.chdir(cwd)
        hinge = os.path.join(ROOT, "hinge_amd", "bin", "hinge")
        # every stage three times in a row, the median quoted: a stage is 0.3-0.8 s of which 0.1-0.3 s is the HIP runtime's
        # start-up in a fresh process, the noisy part (all three samples are reported)
        g, g_runs = {}, {}
        for sub, extra in (("filter", []), ("maximal", []), ("layout", ["-o", "H"])):
            g_runs[sub] = []
            for _ in range(3):
                t0 = time.perf_counter()
                r = subprocess.run([hinge, sub, "--db", "G", "--las", "G.las", "-x", "H", "--config", "nominal.ini"] + extra, cwd=tmp,
                                   stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
                g_runs[sub].append(time.perf_counter() - t0)
                assert r.returncode == 0, r.stderr.decode()[-1000:]
            g[sub] = sorted(g_runs[sub])[1]
        differing = [s for s in E2E_FILES if not filecmp.cmp(os.path.join(tmp, "O" + s), os.path.join(tmp, "H" + s), shallow=False)]
        oracle_hinges = count_pairs(os.path.join(tmp, "O.hinges.txt"))
        e2e = {
            "what": "wall clock of `hinge filter|maximal|layout` (C++ executables over libhinge_hip; .las ingest, H2D, kernels, D2H, text output) "
                    "vs the single-thread CPU oracle on the same files",
            "workload": "%s, %d reads, %d overlap records, .las %.2f GB (page cache warm: just written, %.1f s)" % (workload, d.n_reads, d.novl, las_bytes / 1e9, t_write),
            "cpu_oracle_s": t,
            "gpu_cli_s": g,
            "gpu_cli_s_runs": g_runs,
            "gpu_cli_s_note": "median of three consecutive runs of each stage (every sample listed); the oracle runs once",
            "speedup_filter_layout": (t["filter"] + t["layout"]) / (g["filter"] + g["layout"]),
            "speedup_all_three": sum(t.values()) / sum(g.values()),
            "byte_identical": not differing,
            "files_compared": E2E_FILES,
            "files_differing": differing,
            "host_threads": os.cpu_count(),
        }
        cpu = {
            "value": d.novl / t["filter"],
            "unit": "overlaps/s",
            "cores": 1,
            "kind": "port",
            "sample": "oracle_filter (CPU restatement of filter.cpp incl. .las parse + text output) on the FULL %s data set: %d reads / %d overlaps, %.1f s wall"
                      % (workload, d.n_reads, d.novl, t["filter"]),
        }
        return e2e, cpu, oracle_hinges
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    from hinge_amd import capi, synth
    from hinge_amd.config import default_filter_params
    from hinge_amd.dist import BlockTable, Exchange, HipBackend, ShardedFilter, step_pipelined

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

    P = default_filter_params()
    R = max(1, args.parts)
    base = synth.CONFIGS[args.workload]
    jobs, ctxs, part_ovl, part_reads, part_bins, first_data = [], [], [], [], [], None
    t_gen = time.perf_counter()
    for p in range(R):
        # part p of rank r: its own seed (part 0 of rank 0 is the configuration's data set itself)
        spec = dataclasses.replace(base, n_blocks=1, seed=base.seed + 1000 * rank + 17 * p)
        d = synth.generate(spec)
        pile = synth.to_pileups(d)
        if p == 0:
            first_data = d
        # block table over ranks (block sizes differ slightly through the seed): every block gets the same number of read ids
        # (the largest block's; the extra ids are reads of length 0 without overlaps), so the per-read tables are exchanged by
        # ONE in-place all-gather.  Only the id space is padded.
        sizes = [d.n_reads]
        if use_pg:
            t = torch.tensor([d.n_reads], dtype=torch.int64, device=dev)
            out = torch.empty(world, dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(out, t)
            sizes = [int(v) for v in out.cpu().tolist()]
        S = max(sizes) + int(os.environ.get("HINGE_BENCH_PAD", "0"))   # HINGE_BENCH_PAD: exercise the padding with one rank
        first = [k * S for k in range(world + 1)]
        lo = first[rank]
        hi = lo + d.n_reads          # real reads of this rank: [lo, hi); ids [hi, lo + S) are padding
        n_total = first[-1]
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
        b_flag = ((pile.b_flag & np.uint32(0x7FFFFFFF)) + np.uint32(lo)) | (pile.b_flag & np.uint32(0x80000000))
        # what the ingest hands over besides the columns (hinge_amd/host/host_common.h LasPart::load does the same per record)
        span16, max_pile, in_range = capi.pack_spans(pile.row_ptr, pile.a_span, d.rlen)
        assert span16 is not None
        tens = (torch.from_numpy(row_ptr).to(dev), torch.from_numpy(pile.a_span).to(dev), torch.from_numpy(pile.b_span).to(dev),
                torch.from_numpy(b_flag.view(np.int32)).to(dev), torch.from_numpy(span16.view(np.int32)).to(dev))
        ctx = capi.Context(local_rank)
        last_a = lo + int(d.aread[-1])
        backend = HipBackend(ctx, P, rlen_all, None, lo, lo + S - 1, tens[0], tens[1], tens[2], tens[3], span16=tens[4],
                             facts=(max_pile, in_range), last_a=last_a, coverage_out=True)
        job = ShardedFilter(backend, Exchange(BlockTable(first), dev), mode="merged")
        jobs.append(job); ctxs.append(ctx)
        part_ovl.append(int(pile.n_ovl)); part_reads.append(int(d.n_reads))
        part_bins.append(int(np.sum(d.rlen.astype(np.int64) // P.reso + 2)))
        del pile, d
    t_gen = time.perf_counter() - t_gen
    n_ovl = sum(part_ovl)

    def sync():
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    def one_step():
        # the parts of this rank, software-pipelined: a part's exchanges run under the next part's kernels (one rank without
        # collectives: simply part after part)
        step_pipelined(jobs)

    # ---- warmup (the synchronous variants also size the library's annotation / exact-path buffers) ------------
    for job, ctx in zip(jobs, ctxs):
        ctx.filter_stats(P)
        if use_pg:
            job.x.all_gather_rows(job.mean_cov)
        ctx.filter_median(P, 0, job.x.blocks.n_reads - 1, fetch=True)
        ctx.filter_mask_annotate(P)
        if use_pg:
            job.x.all_gather_rows(job.mask)
        ctx.filter_hinges(P)
    for _ in range(args.warmup):
        one_step()
    for ctx in ctxs:
        ctx.check()

    # ---- untimed breakdown pass: events around EVERY kernel (they cost stream time, so the timed region below brackets
    # only the kernel it prices) --------------------------------------------------------------------------------------
    n_break = max(2, min(args.steps, 5))
    for ctx in ctxs:
        ctx.profile_select(None)
        ctx.profile_enable(12 * n_break + 16)
    for _ in range(n_break):
        one_step()
    sync()
    breakdown = {}
    for ctx in ctxs:
        ctx.check()
        for k, (ms, cnt) in ctx.profile_report().items():
            a = breakdown.setdefault(k, [0.0, 0])
            a[0] += ms; a[1] += cnt
        ctx.profile_enable(0)
    dominant = max(((k, v) for k, v in breakdown.items() if v[1] > 0), key=lambda kv: kv[1][0])[0]

    # ---- timed region: K steps, HIP events around the dominant kernel only ---------------------------------
    for ctx in ctxs:
        ctx.profile_select([dominant])
        ctx.profile_enable(2 * args.steps + 16)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    sync()
    elapsed = time.perf_counter() - t0
    prof_ms, prof_cnt = 0.0, 0
    for ctx in ctxs:
        ctx.check()
        ms, cnt = ctx.profile_report()[dominant]
        prof_ms += ms; prof_cnt += cnt
        ctx.profile_enable(0)
        ctx.profile_select(None)
    # exchange 3, outside the timed region: (read, pos, type) rows of every part
    hinges_per_part, work_reads, exact_annos = [], 0, 0
    for job, ctx in zip(jobs, ctxs):
        rows = job.step(fetch_hinges=True)
        hinges_per_part.append(int(rows.shape[0]))
        c = ctx.counters()
        work_reads += int(c[0]); exact_annos += int(c[1])

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
        kname = dominant
        avg_ms = prof_ms / max(prof_cnt, 1)              # mean launch time over the launches of all R parts
        n_reads = sum(part_reads)
        if kname not in KERNEL_BYTES_PER_OVERLAP:
            alg_bytes = achieved = None                   # sparse kernel: touches only the work-list reads
        else:
            alg_bytes = (KERNEL_BYTES_PER_OVERLAP[kname] * n_ovl + KERNEL_BYTES_PER_READ[kname] * n_reads) / R   # mean per launch
            achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(kname)
        resident = sum(part_ovl) * (8 + 8 + 4 + 4) + sum(part_bins) * 4
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
            "launches_timed": prof_cnt,
            "algorithmic_bytes_per_launch": alg_bytes,
            "coverage_bins_bytes_per_launch": 4 * sum(part_bins) / R if kname == "k_mask_annotate" else None,
            "coverage_bins_note": "k_mask_annotate also stores the cutoff-0 coverage bins (.coverage.txt payload, ~4 B x (rlen/40 + 2) per read); "
                                  "they are NOT counted in `achieved`",
            "kernels_ms_per_step": {k: v[0] / n_break for k, v in breakdown.items() if v[1] > 0},
            "kernels_ms_note": "untimed breakdown pass with events around every kernel (includes event overhead), summed over the %d parts of a step; "
                               "avg_launch_ms is from the timed region" % R,
            "working_set": "%d distinct resident parts, %.2f GB of pile-up columns + span copies + bin output in total; the two streaming kernels of a "
                           "pass read %.0f MB (16|16 span copy) each and K2 writes %.0f MB of bins; %.2f GB pass through between two visits of the same part "
                           "(Infinity Cache: 256 MiB)" % (R, resident / 1e9, 4 * n_ovl / R / 1e6, 4 * sum(part_bins) / R / 1e6, (R - 1) * (2 * 4 * n_ovl / R + 4 * sum(part_bins) / R) / 1e9),
            "path_bytes_per_overlap": PATH_BYTES_PER_OVERLAP,
            "path_achieved_GBs": PATH_BYTES_PER_OVERLAP * n_ovl / (ms_per_step * 1e-3) / 1e9,
        }
        # ---- results are checked, not just printed -----------------------------------------------------------
        expect_path = os.path.join(ROOT, "tests", "golden", "bench_expect.json")
        expect = json.load(open(expect_path)).get(args.workload, {}) if os.path.exists(expect_path) else {}
        checks = {}
        if world == 1:
            want = [expect.get(str(base.seed + 17 * p)) for p in range(R)]
            checks["hinges_expected_per_part"] = want
            checks["hinges_match_committed_oracle_counts"] = all(w is None or w == h for w, h in zip(want, hinges_per_part)) and any(w is not None for w in want)
            assert all(w is None or w == h for w, h in zip(want, hinges_per_part)), "hinge counts %s differ from the oracle's %s" % (hinges_per_part, want)
        e2e = cpu = None
        if world == 1 and not args.no_e2e:
            e2e, cpu, oracle_hinges = end_to_end(first_data, args.workload)
            checks["hinges_part0_vs_live_oracle_run"] = [hinges_per_part[0], oracle_hinges]
            assert hinges_per_part[0] == oracle_hinges, "resident pass found %d hinges on part 0, the oracle's .hinges.txt has %d" % (hinges_per_part[0], oracle_hinges)
            assert e2e["byte_identical"], "executables differ from the oracle: %s" % e2e["files_differing"]
            if args.no_cpu_baseline:
                cpu = None
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
                "workload": "%s: synthetic restatement of E. coli P6-C4 160x (G=4.6 Mb, lognormal reads mean 8.5 kb, 7 x 5 kb repeat copies); "
                            "a step = one whole filter + hinge-detect pass over each of %d distinct such read sets per GPU (seeds %s), all resident in HBM"
                            % (args.workload, R, [base.seed + 17 * p for p in range(R)]),
                "parts_per_step": R,
                "reads_per_gpu": n_reads,
                "overlaps_per_gpu": int(n_ovl),
                "total_overlaps": int(total_ovl),
                "parallelism": "shard-by-block x%d, merged-las semantics; per part one 16 KiB all-reduce (coverage histogram) + one all-gather (masks, 8 B per read)" % world,
                "kernels_in_step": "k_cov_stats, k_median_hist, k_mask_annotate (+ coverage bins), k_hinge_count, k_hinge_call per part; none outside",
                "hinges_found_per_part": hinges_per_part,
                "reads_in_hinge_pass": work_reads,
                "annotations_on_exact_path": exact_annos,
                "generate_s": t_gen,
            },
            "roofline": roofline,
            "checks": checks,
            "e2e": e2e,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
