"""`hinge draft` on the GPU (hinge_amd/bin/draft_assembly over hinge_draft_*).
 * the ladder consensus (k_draft_align + k_draft_cns) against the REFERENCE's own falcon code: the golden vectors it made
   (tests/golden/draft_falcon_golden.json) and, where oracle/_ref/libhinge_ref.so exists (it ships to the GPU box), live on
   fresh random ladders; against the oracle's restatement too;
 * the A-to-B maps of hinge_draft_mappings against get_mapping of the oracle's recoverAlignment + getAlignmentTags (pinned
   through `hinge consensus`);
 * the executable against the oracle's restatement of draft.cpp on synthetic chains (linear, circular, a spanned repeat,
   two-byte traces; noise-free and noisy reads): FASTA and stdout byte for byte - parity unpinned for the program itself
   (draft.cpp needs spdlog + Boost), like the three graph stages."""
import json
import os

import numpy as np
import pytest

import draft_common as dc

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "draft_falcon_golden.json")))


def _db_of_members(tmp, ladders):
    """A read DB whose reads are the ladders' members (every member its own read, some stored reverse-complemented so that the
    strand-1 path of the kernels is used): returns (db path, ladders as (read, strand, start, end) lists)."""
    from hinge_amd import formats
    code = {"a": 0, "c": 1, "g": 2, "t": 3}
    reads, out = [], []
    k = 0
    for mem in ladders:
        rungs = []
        for m in mem:
            fwd = np.array([code[c] for c in m], np.uint8)
            pad_l, pad_r = k % 3, (k // 3) % 4                  # the member sits inside a longer read
            body = np.concatenate([np.full(pad_l, 1, np.uint8), fwd, np.full(pad_r, 2, np.uint8)])
            strand = k % 2
            stored = (3 - body[::-1]).astype(np.uint8) if strand else body
            reads.append(stored)
            rungs.append((k, strand, pad_l, pad_l + len(fwd)))
            k += 1
        out.append(rungs)
    db = os.path.join(tmp, "L")
    formats.write_db(db, np.asarray([len(r) for r in reads], np.int32), bases=reads)
    return db, out


def test_ladder_consensus_matches_the_reference_golden(tmp_path):
    from hinge_amd import capi
    ladders = [c["members"] for c in GOLD]
    db, rungs = _db_of_members(str(tmp_path), ladders)
    ctx = capi.Context(0)
    dr = capi.Draft(ctx, db)
    got = dr.ladders(rungs, [c["mx"] for c in GOLD])
    for k, (g, c) in enumerate(zip(got, GOLD)):
        assert g == c["cns"], (k, len(g), len(c["cns"]))
    ctx.close()


def test_ladder_consensus_matches_the_reference_live(oracle_lib, ref_lib, tmp_path):
    from hinge_amd import capi
    lib, ref = dc.bind(oracle_lib), dc.bind_ref(ref_lib)
    rng = np.random.default_rng(5)
    cases = [dc.random_ladder(rng, case) for case in range(1500)]
    # deep ladders and long members too (64 members is the kernel's limit)
    for n, L in ((40, 300), (64, 120), (3, 4000)):
        truth = "".join("acgt"[i] for i in rng.integers(0, 4, L))
        cases.append(([dc.noisy(rng, truth, 0.12) for _ in range(n)], int(rng.integers(0, n))))
    # runs of 32+ inserted bases in front of one template position (round 6: their column scores live outside LDS in k_draft_cns):
    # most members carry the same 40- / 100- / 200-base insertion the template lacks, noise on top
    for ins_len in (33, 40, 100, 200):
        truth = "".join("acgt"[i] for i in rng.integers(0, 4, 400))
        ins = "".join("acgt"[i] for i in rng.integers(0, 4, ins_len))
        with_ins = truth[:200] + ins + truth[200:]
        mem = [truth] + [dc.noisy(rng, with_ins, e) for e in (0.0, 0.0, 0.02, 0.05, 0.05)] + [dc.noisy(rng, truth, 0.02)]
        cases.append((mem, 0))
    db, rungs = _db_of_members(str(tmp_path), [m for m, _ in cases])
    ctx = capi.Context(0)
    dr = capi.Draft(ctx, db)
    got = dr.ladders(rungs, [mx for _, mx in cases])
    for k, (mem, mx) in enumerate(cases):
        want = dc.ladder_call(ref.ref_falcon_ladder, mem, mx)[1]
        assert got[k] == want, (k, len(mem), mx)
        assert dc.ladder_call(lib.oracle_falcon_ladder, mem, mx)[1] == want
    ctx.close()


def test_mappings_match_get_mapping_of_the_oracle(oracle_lib, tmp_path):
    """hinge_draft_mappings == get_mapping (draft.cpp:70-87) of the gapped rows the oracle's recoverAlignment + getAlignmentTags
    make: read through the oracle's draft log?  No - straight from the rows: the consensus oracle dumps every used alignment's indel
    list, the rows follow from it."""
    from hinge_amd import capi, formats
    from hinge_amd import synth_draft as sd
    wd = str(tmp_path)
    d = sd.generate(sd.CONFIGS["draft_noisy"])
    sd.write_dataset(d, wd, "G")
    las = formats.read_las(os.path.join(wd, "G.las"))
    rng = np.random.default_rng(3)
    picks = np.sort(rng.choice(len(las.rec), size=300, replace=False))
    ctx = capi.Context(0)
    dr = capi.Draft(ctx, os.path.join(wd, "G"))
    maps = dr.mappings(las, picks)
    # the same alignments through `hinge consensus`' test hook (indel lists, pinned against the reference program's behaviour)
    dr.cns.run(las, picks)
    for k, r in enumerate(picks):
        rec = las.rec[r]
        ind = dr.cns.indels(k)
        L = int(rec["aepos"] - rec["abpos"])
        want = np.zeros(L, np.int64)
        gap = np.zeros(L, bool)
        i, j = int(rec["abpos"]) + 1, int(rec["bbpos"]) + 1          # 1-based, as getAlignmentTags walks them
        for p in ind:
            if p < 0:
                run = -p - i
                want[i - 1 - rec["abpos"]:i - 1 - rec["abpos"] + run] = np.arange(j - 1 - rec["bbpos"], j - 1 - rec["bbpos"] + run)
                i += run; j += run
                j += 1                                                # a B base against a gap in A
            else:
                run = p - j
                want[i - 1 - rec["abpos"]:i - 1 - rec["abpos"] + run] = np.arange(j - 1 - rec["bbpos"], j - 1 - rec["bbpos"] + run)
                i += run; j += run
                want[i - 1 - rec["abpos"]] = j - 1 - rec["bbpos"]; gap[i - 1 - rec["abpos"]] = True
                i += 1
        run = int(rec["aepos"]) + 1 - i
        want[i - 1 - rec["abpos"]:] = np.arange(j - 1 - rec["bbpos"], j - 1 - rec["bbpos"] + run)
        got = maps[k]
        assert np.array_equal(got & 0x7FFFFFFF, want), (k, r)
        assert np.array_equal((got >> 31).astype(bool), gap), (k, r)
    ctx.close()


@pytest.mark.parametrize("name", ["draft_clean", "draft_clean_circular", "draft_noisy", "draft_noisy_circular", "draft_repeat", "draft_twobyte"])
def test_draft_executable_matches_the_oracle(oracle_lib, tmp_path, name):
    lib = dc.bind(oracle_lib)
    wd = str(tmp_path)
    dc.prepare(lib, name, wd, stages="executables")
    want_fa, want_log = dc.run_oracle(lib, wd)
    got_fa, got_log = dc.run_product(wd)
    assert got_fa == want_fa, "FASTA differs from the oracle's"
    assert got_log == want_log, "stdout differs from the oracle's"
    ctgs = dc.contigs_of(got_fa)
    assert len(ctgs) >= 2 and max(len(s) for _, s in ctgs) > 10000
    for side in (".garbage.txt", ".contained.txt"):
        assert os.path.getsize(os.path.join(wd, "G" + side)) == 0          # truncated, as the reference's unused streams do
    if name == "draft_clean":
        # and batching: a scratch budget of 1 GiB -> the same text
        assert dc.run_product(wd, out="G.hip2", env={"HINGE_DRAFT_SCRATCH_GB": "1"})[0] == want_fa


def test_dispatcher_runs_draft_path_and_draft(oracle_lib, tmp_path):
    """`hinge draft-path` + `hinge draft` through the dispatcher script, as demo/ecoli_demo/run.sh:29-33 calls them."""
    import subprocess
    lib = dc.bind(oracle_lib)
    wd = str(tmp_path)
    dc.prepare(lib, "draft_clean", wd, stages="executables")
    os.remove(os.path.join(wd, "G.edges.list"))
    hinge = os.path.join(dc.ROOT, "hinge_amd", "bin", "hinge")
    r = subprocess.run([hinge, "draft-path", wd, "G", os.path.join(wd, "G.clip.G2.graphml")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    r = subprocess.run([hinge, "draft", "--db", "G", "--las", "G.las", "--prefix", "G", "--config", "nominal.ini", "--out", "G.draft"], cwd=wd,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    assert open(os.path.join(wd, "G.draft.fasta"), "rb").read() == dc.run_oracle(lib, wd)[0]


def test_chain_layout_to_consensus_recovers_the_genome(oracle_lib, tmp_path):
    """layout -> clip(G2) -> draft-path -> draft -> consensus END TO END through the executables on noisy reads (10 % errors) off a
    planted genome: `hinge filter | maximal | layout` (GPU), `hinge clip`, `hinge draft-path`, `hinge draft` (GPU), then `hinge
    consensus` (GPU) over contig-vs-read alignments composed from the generator's edit scripts (tests/chain_common.py: the
    reference's pipeline runs DALIGNER there).  The consensus FASTA is the CPU oracle's byte for byte, and every contig equals the
    planted genome in >= 99.9 % of the positions of its interior (6 kb left out at either end; the draft: ~97-98 %) - physics, not a pin."""
    import ctypes
    import subprocess
    import chain_common as cc
    lib = dc.bind(oracle_lib)
    wd = str(tmp_path)
    d = dc.prepare(lib, "draft_noisy", wd, stages="executables")
    draft_fa, _ = dc.run_product(wd)
    exe = os.path.join(dc.ROOT, "hinge_amd", "bin", "consensus")

    def run(wd):
        r = subprocess.run([exe, "draft", "G", "draft.G.las", "cns.fasta", "nominal.ini"], cwd=wd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        return open(os.path.join(wd, "cns.fasta"), "rb").read()
    res = cc.polish(d, wd, draft_fa, run)
    assert len(res) >= 2
    for nm, before, after, seq in res:
        (ident, span), (ident0, _) = after.inner_identity(6000), before.inner_identity(6000)
        assert span > 0.5 * d.spec.genome_len, (nm, span)
        assert ident >= 0.999 and ident > ident0 and after.identity > before.identity, (nm, ident0, ident, before.identity, after.identity)
    lib.oracle_consensus.argtypes = [ctypes.c_char_p] * 7
    old = os.getcwd()
    os.chdir(wd)
    try:
        assert lib.oracle_consensus(b"draft", b"G", b"draft.G.las", b"ora.fasta", b"nominal.ini", b"ora.log", None) == 0
    finally:
        os.chdir(old)
    assert open(os.path.join(wd, "ora.fasta"), "rb").read() == open(os.path.join(wd, "cns.fasta"), "rb").read()
