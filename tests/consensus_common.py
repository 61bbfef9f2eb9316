"""Shared by the `hinge consensus` tests: data sets, the reference's own program (oracle/_ref/consensus) where it exists."""
import ctypes
import hashlib
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "consensus")
EXE = os.path.join(ROOT, "hinge_amd", "bin", "consensus")
GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "consensus_golden.json")))


def sha(b):
    return hashlib.sha256(b).hexdigest()


def make(name, wd, **over):
    import dataclasses
    from hinge_amd import synth_consensus as sc
    spec = sc.CONFIGS[name]
    if over:
        spec = dataclasses.replace(spec, **over)
    d = sc.generate(spec)
    sc.write_dataset(d, wd)
    return d


def run_reference(wd, out="ref.fasta"):
    """(fasta bytes, stdout bytes) of the reference's own program; None when it was never built."""
    if not os.path.exists(REF_BIN):
        return None
    r = subprocess.run([REF_BIN, "draft", "reads", "draft.reads.las", out, "nominal.ini"], cwd=wd, stdout=subprocess.PIPE, check=True)
    return open(os.path.join(wd, out), "rb").read(), r.stdout


def run_oracle(lib, wd, out="ora.fasta", dump=None):
    lib.oracle_consensus.argtypes = [ctypes.c_char_p] * 7
    old = os.getcwd()
    os.chdir(wd)
    try:
        rc = lib.oracle_consensus(b"draft", b"reads", b"draft.reads.las", out.encode(), b"nominal.ini", b"ora.log", dump.encode() if dump else None)
    finally:
        os.chdir(old)
    assert rc == 0, rc
    return open(os.path.join(wd, out), "rb").read(), open(os.path.join(wd, "ora.log"), "rb").read()


def run_product(wd, out="hip.fasta", env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([EXE, "draft", "reads", "draft.reads.las", out, "nominal.ini"], cwd=wd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return open(os.path.join(wd, out), "rb").read(), r.stdout
