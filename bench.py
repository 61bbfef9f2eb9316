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
exactly as it does in production.  With N > 1 every rank owns one DAZZ_DB block of every part (weak scaling; teams of two ranks
share a 2-block data set, so half of every pile-up's B reads live on the team mate: hinge_amd/benchsets.py) and the path's
exchange steps run as RCCL collectives between the kernels, batched over the parts: ONE all-reduce (coverage histograms) and
ONE all-gather (masks) per step (hinge_amd/dist.py, PartBatch).  At every N the hinges of every part of every rank - count and
a digest of the (read, position, type) rows - are asserted against the CPU oracle's (tests/golden/bench_expect.json).

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
    ap.add_argument("--gather-groups", type=int, default=1, help="all-gathers of the masks per step (1: all parts at once; 2: the first half's runs under the second half's kernels)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N > 1: weak = every GPU keeps one config-2-sized block per part (teams of two, the default the driver's scaling run "
                         "measures); strong = ONE config-2 data set per part split into N DB blocks (3.3 M overlaps per GPU and part at N = 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    return ap.parse_args()


def pmc_traffic(kname, batched=False):
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
            base = row["kernel"].split("::")[-1].split("<")[0]
            if not base.startswith(kname):
                continue   # k_mask_annotate covers k_mask_annotate_q20 and the general k_mask_annotate<RESO>
            if base.endswith("_batch") != batched:
                continue   # one launch per part (warm-up, HINGE_K2_BATCH=0) and one launch for all parts are different kernels
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
        ref_slice = reference_slice(d, tmp)
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
        # ... and the three stages in ONE process (`hinge pipeline`: one HIP start-up, one ingest, one teardown), prefix P
        pipe_runs = []
        for _ in range(3):
            t0 = time.perf_counter()
            r = subprocess.run([hinge, "pipeline", "--db", "G", "--las", "G.las", "-x", "P", "--config", "nominal.ini", "-o", "P"], cwd=tmp,
                               stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            pipe_runs.append(time.perf_counter() - t0)
            assert r.returncode == 0, r.stderr.decode()[-1000:]
        differing += ["pipeline:" + s for s in E2E_FILES if not filecmp.cmp(os.path.join(tmp, "O" + s), os.path.join(tmp, "P" + s), shallow=False)]
        oracle_hinges = count_pairs(os.path.join(tmp, "O.hinges.txt"))
        e2e = {
            "what": "wall clock of `hinge filter|maximal|layout` (C++ executables over libhinge_hip; .las ingest, H2D, kernels, D2H, text output) "
                    "vs the single-thread CPU oracle on the same files",
            "workload": "%s, %d reads, %d overlap records, .las %.2f GB (page cache warm: just written, %.1f s)" % (workload, d.n_reads, d.novl, las_bytes / 1e9, t_write),
            "cpu_oracle_s": t,
            "gpu_cli_s": g,
            "gpu_cli_s_runs": g_runs,
            "gpu_cli_s_note": "median of three consecutive runs of each stage (every sample listed); the oracle runs once",
            "pipeline_one_process_s": sorted(pipe_runs)[1],
            "pipeline_one_process_s_runs": pipe_runs,
            "pipeline_note": "`hinge pipeline`: the same three stages in one process (files byte-identical, compared above); reported beside the per-stage numbers, not instead",
            "speedup_all_three_one_process": sum(t.values()) / sorted(pipe_runs)[1],
            "speedup_filter_layout": (t["filter"] + t["layout"]) / (g["filter"] + g["layout"]),
            "speedup_all_three": sum(t.values()) / sum(g.values()),
            "speedup_filter_vs_reference_lower_bound": (ref_slice["seconds_full_file"] / g["filter"]) if ref_slice else None,
            "speedup_filter_vs_reference_lower_bound_note": "the reference's OWN getOverlap + pile-up sort + profileCoverage x2 (cpu_baseline.reference_slice: a strict "
                                                            "subset of what its `hinge filter` does) over the same .las on this host, one thread, divided by the GPU executable's "
                                                            "whole `hinge filter` wall clock (ingest, HIP start-up, kernels, text output)",
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
            "reference_slice": ref_slice,
        }
        return e2e, cpu, oracle_hinges
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def reference_slice(d, tmp):
    """The reference's OWN code on this host (oracle/_ref/libhinge_ref.so: LAInterface.cpp, DB.c, align.c compiled unmodified; the
    comparator of filter.cpp through LAInterface.h): `ref_filter_slice` of oracle/ref_shim.cpp - getOverlap over the .las, the pile-up
    index + std::sort(compare_overlap), profileCoverage twice per read - one thread, timed phase by phase.  The whole file by default;
    HINGE_BENCH_REF_SLICE=quarter (or less than 32 GB of free host memory: the reference keeps one heap LOverlap + one trace
    allocation per record, 12.8 GB for this file) runs it on a .las holding the first quarter of the A reads and scales by the
    record count.  None when oracle/_ref was never built."""
    import ctypes
    import numpy as np
    import oracle
    from hinge_amd import synth
    lib = oracle.ref_lib()
    if lib is None or not hasattr(lib, "ref_filter_slice"):
        return None
    mode = os.environ.get("HINGE_BENCH_REF_SLICE", "")
    if not mode:
        try:
            avail = [int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable:")][0] * 1024
        except Exception:
            avail = 0
        mode = "full" if avail >= 32 * 2 ** 30 else "quarter"
    if mode == "none":
        return None
    las = os.path.join(tmp, "G.las")
    n_rec = d.novl
    if mode != "full":
        sel = np.nonzero(d.aread < d.n_reads // 4)[0]
        las = os.path.join(tmp, "Q.las")
        synth.write_las_file(d, las, sel)
        n_rec = int(len(sel))
    lib.ref_filter_slice.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_longlong)]
    lib.ref_filter_slice.restype = ctypes.c_int
    secs = (ctypes.c_double * 3)()
    cnt = (ctypes.c_longlong * 3)()
    rc = lib.ref_filter_slice(os.path.join(tmp, "G").encode(), las.encode(), 40, 300, secs, cnt)
    if rc != 0 or cnt[0] != n_rec:
        return {"error": "ref_filter_slice rc=%d records=%d (expected %d)" % (rc, cnt[0], n_rec)}
    total = sum(secs)
    return {
        "kind": "reference",
        "cores": 1,
        "sample": "%s: %d of %d overlap records (%s)" % (os.path.basename(las), n_rec, d.novl, "the whole file" if mode == "full" else "A reads of the first quarter, scaled by records"),
        "seconds": total,
        "seconds_by_phase": {"openDB + getOverlap": secs[0], "pile-up index + std::sort(compare_overlap)": secs[1], "profileCoverage x2 per read": secs[2]},
        "seconds_full_file": total * d.novl / n_rec,
        "value": n_rec / total,
        "unit": "overlaps/s",
        "covers": "filter.cpp:474-512 (openDB, openAlignmentFile, getOverlap: LAInterface.cpp:1519-1634), :527-548 (pile-up index, self-overlaps dropped), "
                  ":565-567 (std::sort with compare_overlap), :588-598 (profileCoverage with CUT_OFF and with 0: LAInterface.cpp:4298-4320)",
        "omits": "inline code of filter.cpp's main() (needs spdlog: unbuildable here): idx_ab maps + dedup pile-up (:569-583), .coverage.txt text + gradient (:599-610), "
                 "median / MIN_COV (:642-678), coverage + QV masks (:696-789), repeat annotation + merge + gate (:796-865), hinge calling (:867-1068), writers - "
                 "so this is a LOWER bound on the reference's `hinge filter` time on this host",
        "checksum": int(cnt[2]),
    }


def describe_workload(name):
    """What the synthetic read sets of `--workload name` restate (hinge_amd/synth.py CONFIGS): from the spec itself."""
    from hinge_amd import synth
    c = synth.CONFIGS[name]
    what = {"cfg2_ecoli160": "synthetic restatement of E. coli P6-C4 160x", "cfg1_ecoli_demo": "synthetic restatement of the ecoli_demo plumbing case",
            "cfg3_nctc": "synthetic restatement of a repeat-rich NCTC-like bacterial set", "cfg4_yeast": "synthetic restatement of a yeast-like set"}.get(name, "synthetic read set")
    reps = "%d repeat famil%s of %d-%d bp in %d-%d copies" % (c.n_repeat_families, "y" if c.n_repeat_families == 1 else "ies", c.repeat_len[0], c.repeat_len[1],
                                                             c.repeat_copies[0], c.repeat_copies[1])
    return "%s (G=%.1f Mb at %dx, %s reads mean %d bp, %s%s)" % (what, c.genome_len / 1e6, c.coverage, c.len_dist, c.len_mean, reps,
                                                                 ", %.0f %% chimeric reads" % (100 * c.chimera_frac) if getattr(c, "chimera_frac", 0) else "")


def main():
    args = parse_args()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # (the host driver only supports dmabuf IPC: RCCL between processes needs it)
    import torch
    import torch.distributed as dist

    from hinge_amd import benchsets, capi, synth
    from hinge_amd.config import default_filter_params
    from hinge_amd.dist import resident_batch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if not benchsets.supported_world(world, args.scaling):
        raise SystemExit("bench.py: 1 GPU or an even number of GPUs (the ranks work in teams of two, hinge_amd/benchsets.py); --scaling strong takes any N")
    strong = args.scaling == "strong" and world > 1
    # test rig for a 1-GPU box: HINGE_BENCH_ONE_DEVICE=1 puts every rank on device 0 and HINGE_BENCH_BACKEND=gloo moves the
    # collectives through host memory (RCCL refuses two ranks on one device).  Results are asserted as usual; times mean nothing.
    one_device = os.environ.get("HINGE_BENCH_ONE_DEVICE", "0") == "1"
    pg_backend = os.environ.get("HINGE_BENCH_BACKEND", "nccl")
    dev_index = 0 if one_device else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_pg = world > 1 or ("RANK" in os.environ and os.environ.get("HINGE_FORCE_COLLECTIVES", "0") == "1")
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if pg_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(pg_backend)

    def host_all_gather(vals):
        if not use_pg:
            return [list(vals)]
        d_ = dev if pg_backend == "nccl" else torch.device("cpu")
        t = torch.tensor(list(vals), dtype=torch.int64, device=d_)
        out = torch.empty(world * len(vals), dtype=torch.int64, device=d_)
        dist.all_gather_into_tensor(out, t)
        flat = out.cpu().tolist()
        return [flat[k * len(vals):(k + 1) * len(vals)] for k in range(world)]

    P = default_filter_params()
    R = max(1, args.parts)
    base = synth.CONFIGS[args.workload]
    t_gen = time.perf_counter()
    first_data = None
    parts = []
    for p in range(R):
        spec, _ = benchsets.part_spec(base, world, rank, p, args.scaling)
        d = synth.generate(spec)
        if p == 0 and world == 1:
            first_data = d
        parts.append(benchsets.rank_part(base, world, rank, p, data=d, scaling=args.scaling))
        del d
    t_gen = time.perf_counter() - t_gen
    # every block gets the same number of read ids (the largest block's over all ranks and parts; the ids behind a block's
    # reads are reads of length 0 without overlaps), so the mask tables are exchanged by in-place all-gathers
    batch, ctxs = resident_batch(parts, P, dev, gather_groups=args.gather_groups, pad=int(os.environ.get("HINGE_BENCH_PAD", "0")))
    if os.environ.get("HINGE_TEST_CORRUPT_GATHER", "0") == "1":      # tests/test_dist_gpu.py: the assertions below must catch a broken exchange 2
        batch.after_gather = batch.corrupt_foreign_rows
    S = batch.S
    part_ovl = [rp.n_ovl for rp in parts]
    part_reads = [rp.n_reads for rp in parts]
    part_records = [rp.n_records for rp in parts]
    part_bins = [int(np.sum(rp.rlen.astype(np.int64) // P.reso + 2)) for rp in parts]
    del parts
    n_ovl = sum(part_ovl)

    def sync():
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    one_step = batch.step      # all parts of this rank, exchanges batched (hinge_amd/dist.py PartBatch)

    # ---- warmup: the synchronous entry points size the library's annotation / exact-path buffers, then whole steps until no
    # rank reports a full buffer ---------------------------------------------------------------------------------------
    for p, ctx in enumerate(ctxs):
        lo = batch.id_base(p)
        if batch.one_sweep:
            ctx.filter_sweep(P, fetch=True)          # (regrows the annotation buffer and repeats the pass when it overflows)
        else:
            ctx.filter_stats(P)
            ctx.filter_median(P, lo, lo + S - 1, fetch=True)
            ctx.filter_mask_annotate(P)
        ctx.filter_hinges(P)
    batch.settle()
    for _ in range(args.warmup):
        one_step()
    batch.status()

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
    batch.status()
    prof_ms, prof_cnt = 0.0, 0
    for ctx in ctxs:
        ms, cnt = ctx.profile_report()[dominant]
        prof_ms += ms; prof_cnt += cnt
        ctx.profile_enable(0)
        ctx.profile_select(None)

    # ---- results, outside the timed region: exchange 3 (the (read, pos, type) rows of every part from every rank, on every
    # rank) and the per-rank counters --------------------------------------------------------------------------------
    spec_stats = [[int(v) for v in ctx.spec_stats()] for ctx in ctxs]   # one-sweep pass: verifications, mispredictions, guard-band reads
    lists = [t.cpu().numpy() for t in batch.hinge_lists()]
    work_reads = sum(int(ctx.counters()[0]) for ctx in ctxs)
    exact_annos = sum(int(ctx.counters()[1]) for ctx in ctxs)
    table_sums = batch.table_checksums()                    # every rank must hold the same mask tables after exchange 2
    all_sums = host_all_gather(table_sums)
    got = []                                                # [part][rank] -> {"hinges", "digest"}
    for p in range(R):
        rows = lists[p]
        per_rank = []
        for r in range(world):
            lo = batch.id_base(p, r)
            sel = (rows[:, 0] >= lo) & (rows[:, 0] < lo + S)
            loc = rows[sel].astype(np.int64)
            loc[:, 0] -= lo
            per_rank.append({"hinges": int(len(loc)), "digest": benchsets.digest(loc)})
        assert sum(e["hinges"] for e in per_rank) == len(rows)
        got.append(per_rank)
    sizes = host_all_gather(part_reads + part_records)      # [rank] -> reads of every part, then records of every part

    if use_pg:
        d_ = dev if pg_backend == "nccl" else torch.device("cpu")
        t = torch.tensor([elapsed], dtype=torch.float64, device=d_)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        t = torch.tensor([n_ovl], dtype=torch.int64, device=d_)
        dist.all_reduce(t)
        total_ovl = int(t.item())
    else:
        total_ovl = n_ovl

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = total_ovl * args.steps / elapsed
        kname = dominant
        avg_ms = prof_ms / max(prof_cnt, 1)              # mean launch time over the timed region's launches
        # parts per launch: 1 when every part has a launch of its own, R when one launch sweeps all resident parts (the batched
        # k_mask_annotate_q20) - from the launch count itself, so reruns of a part count as launches
        ppl = R * args.steps / max(prof_cnt, 1)
        n_reads = sum(part_reads)
        if kname not in KERNEL_BYTES_PER_OVERLAP:
            alg_bytes = achieved = phys_bytes = None      # sparse kernel: touches only the work-list reads
        else:
            alg_bytes = (KERNEL_BYTES_PER_OVERLAP[kname] * n_ovl + KERNEL_BYTES_PER_READ[kname] * n_reads) / R * ppl   # mean per launch
            achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
            # what the kernel has to move: the 16|16 span copy (4 B per overlap, packed by the ingest), the per-read tables and,
            # for the mask / annotate kernel, the coverage bins it stores
            phys_bytes = (4 * n_ovl + KERNEL_BYTES_PER_READ[kname] * n_reads + (4 * sum(part_bins) if kname == "k_mask_annotate" else 0)) / R * ppl
        one_sweep = bool(getattr(batch, "one_sweep", False)) and "k_cov_stats" not in {k for k, v in breakdown.items() if v[1] > 0}
        traffic, traffic_src = pmc_traffic(kname, batched=ppl > 1.5)
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
            "parts_per_launch": ppl,
            "algorithmic_bytes_per_launch": alg_bytes,
            "physical_bytes_per_launch": phys_bytes,
            "achieved_physical": (phys_bytes / (avg_ms * 1e-3) / 1e9) if phys_bytes else None,
            "frac_physical": (phys_bytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if phys_bytes else None,
            "waste_ratio": (traffic / phys_bytes) if (traffic and phys_bytes) else None,
            "one_sweep_note": ("the dominant kernel is the FIRST AND ONLY sweep of the pass (DESIGN.md 3.6): it does the work SURVEY 8(d) prices as pass 1 (8 B per overlap) "
                               "plus pass 2's span read (8 B per overlap); `frac` keeps crediting it 8 B per overlap - what one sweep has to read - so it is "
                               "comparable with rounds 1-3, where a second kernel (k_cov_stats, 32 us per part) was credited the other 8") if one_sweep else None,
            "frac_if_credited_both_passes": ((alg_bytes + (8 * n_ovl / R + 20 * n_reads / R) * ppl) / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (one_sweep and achieved is not None and kname == "k_mask_annotate") else None,
            "physical_note": "`frac` credits SURVEY 8(d)'s 8 B per overlap (the int32 span pair); the kernel reads the ingest's 16|16 copy (4 B per "
                             "overlap) and, unlike 8(d)'s accounting, writes the coverage bins: physical = 4 B x overlaps + per-read tables + bins; "
                             "waste_ratio = counter traffic / physical",
            "coverage_bins_bytes_per_launch": 4 * sum(part_bins) / R * ppl if kname == "k_mask_annotate" else None,
            "coverage_bins_note": "k_mask_annotate also stores the cutoff-0 coverage bins (.coverage.txt payload, ~4 B x (rlen/40 + 2) per read); "
                                  "they are NOT counted in `achieved`",
            "kernels_ms_per_step": {k: v[0] / n_break for k, v in breakdown.items() if v[1] > 0},
            "kernels_ms_note": "untimed breakdown pass with events around every kernel (includes event overhead), summed over the %d parts of a step; "
                               "avg_launch_ms is from the timed region" % R,
            "working_set": "%d distinct resident parts, %.2f GB of pile-up columns + span copies + bin output in total; a pass's sweep(s) "
                           "read %.0f MB (16|16 span copy) per part and K2 writes %.0f MB of bins; %.2f GB pass through between two visits of the same part "
                           "(Infinity Cache: 256 MiB)" % (R, resident / 1e9, 4 * n_ovl / R / 1e6, 4 * sum(part_bins) / R / 1e6, (R - 1) * (4 * n_ovl / R + 4 * sum(part_bins) / R) / 1e9),
            "path_bytes_per_overlap": PATH_BYTES_PER_OVERLAP,
            "path_credit_GBs": PATH_BYTES_PER_OVERLAP * n_ovl / (ms_per_step * 1e-3) / 1e9,
            "path_credit_note": "SURVEY 8(d)'s 40 B per overlap x overlaps / step time: a figure of merit for the whole path, NOT achieved bandwidth "
                                "(the design reads ~16 B per overlap)",
        }
        # ---- results are checked, not just printed: at EVERY N, every part of every rank against the CPU oracle's hinge count
        # and row digest (tests/golden/bench_expect.json, tools/make_bench_expect.py) ----------------------------------
        expect_path = os.environ.get("HINGE_BENCH_EXPECT") or os.path.join(ROOT, "tests", "golden", "bench_expect.json")
        expect = json.load(open(expect_path)).get(args.workload, {}).get("worlds", {}).get("1" if strong else str(world), {}) if os.path.exists(expect_path) else {}
        if strong:
            # strong scaling: the N blocks are ONE config-2 data set, merged-las semantics - the union of the ranks' rows, in
            # the data set's own read ids, must be the N = 1 result (the committed expectation of world 1)
            merged = []
            for p in range(R):
                first = np.concatenate([[0], np.cumsum([sizes[r][p] for r in range(world)])])
                rows = lists[p].astype(np.int64)
                out_rows = []
                for r in range(world):
                    lo = batch.id_base(p, r)
                    sel = (rows[:, 0] >= lo) & (rows[:, 0] < lo + S)
                    loc = rows[sel].copy()
                    loc[:, 0] += int(first[r]) - lo
                    out_rows.append(loc)
                allr = np.concatenate(out_rows) if out_rows else np.zeros((0, 3), np.int64)
                merged.append([{"hinges": int(len(allr)), "digest": benchsets.digest(allr)}])
            per_rank_got, got = got, merged
            sizes = [[sum(sizes[r][p] for r in range(world)) for p in range(2 * R)]]
        checks = {"hinges_per_part_and_rank": [[e["hinges"] for e in pr] for pr in (per_rank_got if strong else got)], "parts_checked": 0,
                  "mask_tables_identical_on_all_ranks": all(s_ == all_sums[0] for s_ in all_sums)}
        assert checks["mask_tables_identical_on_all_ranks"], "exchange 2 left different mask tables on different ranks: %s" % (all_sums,)
        for p in range(R):
            want = expect.get(str(p))
            if want is None or os.environ.get("HINGE_BENCH_NO_ASSERT") == "1":     # (NO_ASSERT: ablation builds produce garbage on purpose)
                continue
            have_sizes = ([sizes[r][p] for r in range(len(sizes))], [sizes[r][R + p] for r in range(len(sizes))])
            assert (want["reads"], want["records"]) == have_sizes, "part %d: the generated read sets %s are not the ones the expectations were made for %s" % (p, have_sizes, (want["reads"], want["records"]))
            assert want["ranks"] == got[p], "part %d: hinges (count, digest) per rank %s differ from the CPU oracle's %s" % (p, got[p], want["ranks"])
            checks["parts_checked"] += 1
        checks["hinges_and_digests_match_cpu_oracle"] = checks["parts_checked"] > 0
        assert checks["parts_checked"] > 0 or R > 4 or args.workload != "cfg2_ecoli160" or os.environ.get("HINGE_BENCH_NO_ASSERT") == "1", "no committed expectation for N = %d" % world
        e2e = cpu = None
        if world == 1 and not args.no_e2e:
            e2e, cpu, oracle_hinges = end_to_end(first_data, args.workload)
            checks["hinges_part0_vs_live_oracle_run"] = [got[0][0]["hinges"], oracle_hinges]
            assert got[0][0]["hinges"] == oracle_hinges, "resident pass found %d hinges on part 0, the oracle's .hinges.txt has %d" % (got[0][0]["hinges"], oracle_hinges)
            assert e2e["byte_identical"], "executables differ from the oracle: %s" % e2e["files_differing"]
            if args.no_cpu_baseline:
                cpu = None
        consensus = None
        if world == 1 and not args.no_e2e:
            # SURVEY 8(f-4), reported beside the headline metric: `hinge consensus` at E. coli size (tools/cns_bench.py) - the
            # reference's own program (oracle/_ref/consensus, where it was built) against the executable and the kernels; FASTA
            # byte-identical or the run fails
            try:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cns_bench.py")] + (["--no-cpu"] if args.no_cpu_baseline else []),
                                   stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
                assert r.returncode == 0, r.stderr.decode()[-1500:]
                consensus = json.loads(r.stdout.decode().strip().splitlines()[-1])
            except Exception as ex:      # (reported, not fatal: the headline metric is the filter path's)
                consensus = {"error": str(ex)[-800:]}
        draft = None
        if world == 1 and not args.no_e2e:
            # SURVEY 8(f-4), `hinge draft`'s ladder step at E. coli size (tools/draft_bench.py): k_draft_align + k_draft_cns on 5 000
            # ladders of ~25 members against the reference's OWN falcon (oracle/_ref: falcon.c + DW_banded.c unmodified) on a sample
            # of the same ladders, byte-identical or the run fails
            try:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "draft_bench.py")] + (["--no-cpu"] if args.no_cpu_baseline else []),
                                   stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
                assert r.returncode == 0, r.stderr.decode()[-1500:]
                draft = json.loads(r.stdout.decode().strip().splitlines()[-1])
            except Exception as ex:      # (reported, not fatal)
                draft = {"error": str(ex)[-800:]}
        maximal_kernel = None
        if world == 1 and not args.no_e2e:
            # SURVEY 8(a) rows 14-15, reported beside the headline metric: the kernel `hinge maximal` runs over every overlap of the
            # bench part (tools/k4_bench.py --json: k_trim_classify_image, launch time from HIP events, its own roofline)
            try:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "k4_bench.py"), "--json", "--reps", "5"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
                assert r.returncode == 0, r.stderr.decode()[-1500:]
                maximal_kernel = json.loads(r.stdout.decode().strip().splitlines()[-1])
            except Exception as ex:      # (reported, not fatal)
                maximal_kernel = {"error": str(ex)[-800:]}
        collectives = batch.collectives
        out = {
            "metric": "overlaps/sec through filter+hinge-detect, E. coli 160x" if args.workload == "cfg2_ecoli160" else "overlaps/sec through filter+hinge-detect, %s (NOT the headline configuration)" % args.workload,
            "value": value,
            "unit": "overlaps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": args.scaling if world > 1 else "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {
                "workload": ("%s: %s; "
                             "a step = one whole filter + hinge-detect pass over each of %d distinct such read sets per GPU, all resident in HBM"
                             % (args.workload, describe_workload(args.workload), R)) +
                            ("" if world == 1 else "; N > 1: teams of two ranks share a 2-block data set of twice the genome (same reads and overlaps per GPU; "
                                                   "half of every pile-up's B reads are on the team mate: hinge_amd/benchsets.py)"),
                "parts_per_step": R,
                "reads_per_gpu": n_reads,
                "overlaps_per_gpu": int(n_ovl),
                "total_overlaps": int(total_ovl),
                "parallelism": "shard-by-block x%d, merged-las semantics; per STEP %s" % (
                    world, ("one all-reduce (the %d coverage histograms, %d KiB) + %d all-gather(s) (masks, 8 B per read)"
                            % (R, R * 16, len(batch.groups))) if collectives else "no collective (one rank)"),
                "collectives_per_step": (1 + len(batch.groups)) if collectives else 0,
                "process_group": (pg_backend + (" (test rig: all ranks on one device)" if one_device else "")) if use_pg else None,
                "kernels_in_step": "one-sweep pass: k_spec_predict (MIN_COV from a sample of each part), k_mask_annotate (k_mask_annotate_q20_batch: + coverage bins, + the per-read coverage sums), "
                                   "k_median_hist (the exact median: verification), k_mask_annotate_final (the guard-band reads; every read of a part whose prediction missed the band), "
                                   "k_hinge_count, k_hinge_call - every one of them ONE launch for all parts; none outside.  HINGE_K2_BATCH=0: a k_mask_annotate launch per part; HINGE_ONE_SWEEP=0: k_cov_stats first (rounds 1-3)",
                "one_sweep": {"per_part": [{"passes_verified": s_[0], "exact_differs_from_prediction": s_[1], "prediction_outside_band_whole_part_redone": s_[2],
                                            "guard_band_reads_last_pass": s_[3], "predicted_min_cov": s_[4], "exact_min_cov": s_[5]} for s_ in spec_stats],
                              "note": "cumulative over warm-up, breakdown and timed steps; redone parts and guard-band launches are inside the timed region"},
                "reads_in_hinge_pass": work_reads,
                "annotations_on_exact_path": exact_annos,
                "generate_s": t_gen,
            },
            "roofline": roofline,
            "checks": checks,
            "e2e": e2e,
            "draft": draft,
            "consensus": consensus,
            "maximal_kernel": maximal_kernel,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
