#!/usr/bin/env python
"""`hinge filter --mlas` with one rank per visible GPU, twice: mask rows over RCCL (the default where every rank has its own
device) and through the host (HINGE_HOST_EXCHANGE=1); the stage's output files must be byte-identical, and the first run's log
must say that the rows went over RCCL.  Part of tools/scale_smoke.sh (first contact with a multi-GPU node); on a 1-GPU box both
runs are the sequential loop and the check only says so.     python tools/mlas_rccl_check.py [--genome 3000000]"""
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
    args = ap.parse_args()
    import conftest
    from hinge_amd import synth
    spec = dataclasses.replace(synth.CONFIGS["cfg4_yeast"], genome_len=args.genome)
    d = synth.generate(spec)
    tmp = tempfile.mkdtemp(prefix="hinge_mlas_")
    outs = {}
    exe = os.path.join(ROOT, "hinge_amd", "bin", "Reads_filter")
    for mode, env in (("rccl", {}), ("host", {"HINGE_HOST_EXCHANGE": "1"})):
        wd = os.path.join(tmp, mode)
        os.makedirs(wd)
        synth.write_dataset(d, wd, "G", write_bases=False)
        conftest.write_ini(os.path.join(wd, "nominal.ini"))
        r = subprocess.run([exe, "--db", "G", "--las", "G", "--mlas", "-x", "G", "--config", "nominal.ini"], cwd=wd,
                           env=dict(os.environ, HINGE_DEBUG_PATHS="1", **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        outs[mode] = r.stdout.decode(errors="replace")
        print("[%s] rc=%d  %s" % (mode, r.returncode, " | ".join(l for l in outs[mode].splitlines() if "ranks" in l or "RCCL" in l)))
        if r.returncode != 0:
            print(outs[mode][-2000:])
            return 2
    files = sorted(f for f in os.listdir(os.path.join(tmp, "rccl")) if f.startswith("G.") and not f.endswith((".las", ".db")))
    diff = [f for f in files if not filecmp.cmp(os.path.join(tmp, "rccl", f), os.path.join(tmp, "host", f), shallow=False)]
    over_rccl = "over RCCL" in outs["rccl"]
    print("files compared: %d, differing: %s; first run exchanged over RCCL: %s" % (len(files), diff, over_rccl))
    shutil.rmtree(tmp, ignore_errors=True)
    return 1 if diff else (0 if over_rccl else 3)      # 3: identical, but RCCL was not used (one visible GPU)


if __name__ == "__main__":
    sys.exit(main())
