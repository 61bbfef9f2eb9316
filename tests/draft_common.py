"""Shared by the `hinge draft` tests: the synthetic chain data set -> filter -> maximal -> layout -> clip -> draft-path, and the
oracle's / the product's `hinge draft` on what comes out."""
import ctypes
import os
import subprocess

import numpy as np

from conftest import NOMINAL_INI

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "hinge_amd", "bin", "draft_assembly")
DRAFT_INI = NOMINAL_INI + "\n[draft]\nmin_cov = 10;\ntrim = 200;\nedge_safe = 100;\ntspace = 900;\nstep = 50;\n"


def bind(lib):
    c = ctypes
    lib.oracle_draft.argtypes = [c.c_char_p, c.c_char_p, c.c_int] + [c.c_char_p] * 4
    lib.oracle_falcon_ladder.restype = c.c_long
    lib.oracle_falcon_ladder.argtypes = [c.c_int, c.POINTER(c.c_char_p), c.c_int, c.c_char_p, c.c_long]
    lib.oracle_falcon_align.restype = c.c_long
    lib.oracle_falcon_align.argtypes = [c.c_char_p, c.c_char_p, c.c_int, c.c_char_p, c.c_char_p, c.c_long]
    return lib


def bind_ref(ref):
    c = ctypes
    ref.ref_falcon_ladder.restype = c.c_long
    ref.ref_falcon_ladder.argtypes = [c.c_int, c.POINTER(c.c_char_p), c.c_int, c.c_char_p, c.c_long]
    ref.ref_falcon_align.restype = c.c_long
    ref.ref_falcon_align.argtypes = [c.c_char_p, c.c_char_p, c.c_int, c.c_char_p, c.c_char_p, c.c_long]
    ref.ref_get_coverage.restype = None
    ref.ref_get_coverage.argtypes = [c.c_int, c.POINTER(c.c_int), c.POINTER(c.c_int), c.c_int, c.POINTER(c.c_int)]
    return ref


def ladder_call(fn, members, mx):
    arr = (ctypes.c_char_p * len(members))(*[m.encode() for m in members])
    cap = 4 * max(len(m) for m in members) + 100
    out = ctypes.create_string_buffer(cap)
    n = fn(len(members), arr, mx, out, cap)
    return n, out.value.decode()


def noisy(rng, t, err):
    """A copy of string t with substitutions / insertions / deletions at a total rate of err."""
    ps, pi, pd = err * 0.3, err * 0.45, err * 0.25
    out = []
    for ch in t:
        r = rng.random()
        if rng.random() < pi:
            out.append("acgt"[rng.integers(4)])
        if r < pd:
            continue
        out.append("acgt"[rng.integers(4)] if r < pd + ps else ch)
    return "".join(out)


def random_ladder(rng, case):
    L = int(rng.choice([5, 30, 200, 900, 1200]))
    truth = "".join("acgt"[i] for i in rng.integers(0, 4, L))
    n = int(rng.integers(2, 7))
    err = float(rng.choice([0, 0.02, 0.1, 0.2]))
    mem = [noisy(rng, truth, err) for _ in range(n)]
    if case % 7 == 0:
        mem[1] = mem[1][:len(mem[1]) // 2]                      # a member that ends early
    if case % 11 == 0:
        mem[0] = "".join("acgt"[i] for i in rng.integers(0, 4, L))   # an unrelated member
    mem = [m if m else "a" for m in mem]
    return mem, int(rng.integers(0, n))


def prepare(lib, name, wd, stages="oracle"):
    """Data set `name` of hinge_amd.synth_draft in wd, run through filter / maximal / layout (the oracle's or the executables'),
    `hinge clip` and `hinge draft-path`.  Returns the DraftData."""
    from hinge_amd import clip, draft_path, synth_draft as sd
    d = sd.generate(sd.CONFIGS[name])
    sd.write_dataset(d, wd, "G")
    with open(os.path.join(wd, "nominal.ini"), "w") as f:
        f.write(DRAFT_INI)
    old = os.getcwd()
    os.chdir(wd)
    try:
        if stages == "oracle":
            assert lib.oracle_filter(b"G", b"G.las", 0, b"G", b"nominal.ini", b"") == 0
            assert lib.oracle_maximal(b"G", b"G.las", 0, b"G", b"nominal.ini") == 0
            assert lib.oracle_layout(b"G", b"G.las", 0, b"G", b"G", b"nominal.ini") == 0
        else:
            bin_ = os.path.join(ROOT, "hinge_amd", "bin")
            for prog, extra in (("Reads_filter", []), ("get_maximal_reads", []), ("hinging", ["-o", "G"])):
                r = subprocess.run([os.path.join(bin_, prog), "--db", "G", "--las", "G.las", "-x", "G", "--config", "nominal.ini"] + extra,
                                   stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                assert r.returncode == 0, (prog, r.stderr.decode()[-1500:])
        assert clip.main(["G.edges.hinges", "G.hinge.list", ".clip", "nominal.ini"]) == 0
        assert draft_path.main([".", "G", "G.clip.G2.graphml"]) == 0
    finally:
        os.chdir(old)
    return d


def run_oracle(lib, wd, out="G.ora"):
    old = os.getcwd()
    os.chdir(wd)
    try:
        rc = lib.oracle_draft(b"G", b"G.las", 0, b"G", out.encode(), b"nominal.ini", b"ora.log")
    finally:
        os.chdir(old)
    assert rc == 0, rc
    return open(os.path.join(wd, out + ".fasta"), "rb").read(), open(os.path.join(wd, "ora.log"), "rb").read()


def run_product(wd, out="G.hip", env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([EXE, "--db", "G", "--las", "G.las", "-x", "G", "-o", out, "--config", "nominal.ini"], cwd=wd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    text = b"".join(l for l in r.stdout.splitlines(keepends=True) if not l.startswith(b"[log]"))
    return open(os.path.join(wd, out + ".fasta"), "rb").read(), text


def contigs_of(fasta: bytes):
    out, name = [], None
    for l in fasta.decode().splitlines():
        if l.startswith(">"):
            name = l
        else:
            out.append((name, l))
    return out


def inner_mismatches(d, seq: str, margin: int = 3000):
    """For a draft of NOISE-FREE reads: positions where the draft (lower-cased, the first / last `margin` bases left out - the
    reference cuts prefix and suffix from the wrong strand for strand-1 ends) differs from the genome it is anchored on by a
    60-mer from its middle.  (positions, strand) or None when the anchor is not found."""
    gs = "".join("acgt"[x] for x in d.genome)
    if d.spec.circular:
        gs = gs + gs + gs
    comp = {"a": "t", "c": "g", "g": "c", "t": "a"}
    grc = "".join(comp[c] for c in reversed(gs))
    s = seq.lower()
    mid = len(s) // 2
    for strand, g in (("+", gs), ("-", grc)):
        at = g.find(s[mid:mid + 60], len(gs) // 3 - len(s) if d.spec.circular and False else 0)
        if at < 0:
            continue
        off = at - mid
        a = np.frombuffer(s.encode(), np.uint8)
        lo, hi = max(margin, -off), min(len(s) - margin, len(g) - off)
        b = np.frombuffer(g.encode(), np.uint8)[off + lo:off + hi]
        return np.nonzero(a[lo:hi] != b)[0] + lo, strand
    return None
