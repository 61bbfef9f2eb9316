#!/usr/bin/env python
"""`hinge filter --mlas`, `hinge maximal --mlas` and `hinge layout --mlas` with one rank per visible GPU, twice: the ranks' exchanges
over RCCL (the default where every rank has its own device: the mask rows of filter - hinge_comm_exchange_mask_rows -, the
containment candidates of maximal and the classified matches of layout - hinge_comm_allgather_rows) and through the host
(HINGE_HOST_EXCHANGE=1); every stage's output files must be byte-identical, and the first run's logs must say which exchange went
over RCCL.  Part of tools/scale_smoke.sh (first contact with a multi-GPU node).  On a 1-GPU box the three stages are the
sequential loop; with --one-rank (HINGE_COMM_ONE_RANK=1) maximal and layout still send their rows through a one-rank communicator,
which is what this check then compares with the host path.     python tools/mlas_rccl_check.py [--genome 3000000] [--one-rank]"""
import argparse
import dataclasses
import filecmp
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=int, default=3_000_000)
    ap.add_argument("--one-rank", action="store_true", help="HINGE_COMM_ONE_RANK=1: one-rank communicators where a stage has one rank")
    args = ap.parse_args()
    import conftest
    from hinge_amd import synth
    spec = dataclasses.replace(synth.CONFIGS["cfg4_yeast"], genome_len=args.genome)
    d = synth.generate(spec)
    tmp = tempfile.mkdtemp(prefix="hinge_mlas_")
    outs = {}
    bin_ = os.path.join(ROOT, "hinge_amd", "bin")
    stages = (("Reads_filter", ["--mlas"], "mask rows"), ("get_maximal_reads", ["--mlas"], "containment candidates"), ("hinging", ["--mlas", "-o", "G"], "classified matches"))
    for mode, env in (("rccl", {"HINGE_COMM_ONE_RANK": "1"} if args.one_rank else {}), ("host", {"HINGE_HOST_EXCHANGE": "1"})):
        wd = os.path.join(tmp, mode)
        os.makedirs(wd)
        synth.write_dataset(d, wd, "G", write_bases=False)
        conftest.write_ini(os.path.join(wd, "nominal.ini"))
        outs[mode] = {}
        for exe, extra, what in stages:
            r = subprocess.run([os.path.join(bin_, exe), "--db", "G", "--las", "G", "-x", "G", "--config", "nominal.ini"] + extra, cwd=wd,
                               env=dict(os.environ, HINGE_DEBUG_PATHS="1", **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            outs[mode][exe] = r.stdout.decode(errors="replace")
            print("[%s] %s rc=%d  %s" % (mode, exe, r.returncode, " | ".join(l for l in outs[mode][exe].splitlines() if "ranks" in l or "RCCL" in l)))
            if r.returncode != 0:
                print(outs[mode][exe][-2000:])
                return 2
    files = sorted(f for f in os.listdir(os.path.join(tmp, "rccl")) if f.startswith("G.") and not f.endswith((".las", ".db")))
    diff = [f for f in files if not filecmp.cmp(os.path.join(tmp, "rccl", f), os.path.join(tmp, "host", f), shallow=False)]
    over = {exe: "over RCCL" in outs["rccl"][exe] for exe, _, _ in stages}
    host = {exe: "over RCCL" in outs["host"][exe] for exe, _, _ in stages}
    print("files compared: %d, differing: %s; exchanged over RCCL in the first run: %s; in the HINGE_HOST_EXCHANGE=1 run: %s" % (len(files), diff, over, host))
    shutil.rmtree(tmp, ignore_errors=True)
    if diff or any(host.values()):
        return 1
    want = ("get_maximal_reads", "hinging") if args.one_rank else tuple(over)
    return 0 if all(over[e] for e in want) else 3      # 3: identical, but RCCL was not used (one visible GPU and no --one-rank)


if __name__ == "__main__":
    sys.exit(main())
