"""GPU parity of the installed command-line tools (hinge filter / maximal / layout = Reads_filter,
get_maximal_reads, hinging over libhinge_hip) against the CPU oracle: every output file, byte for byte."""
import filecmp
import os
import subprocess

import pytest

from conftest import ROOT, clone_dataset, run_in, write_ini

pytestmark = pytest.mark.gpu
HINGE = os.path.join(ROOT, "hinge_amd", "bin", "hinge")

FILES = ["G.mas", "G.cmas", "G.repeat.txt", "G.hinges.txt", "G.coverage.txt", "G.cov.flag", "G.self.flag", "G.homologous.txt",
         "G.filtered.fasta", "G.max", "G.contained.txt", "G.garbage.txt", "G.killed.hinges", "G.edges.hinges", "G.edges.hinges2",
         "G.hinge.list", "G.deadends.txt", "G.hgraph", "G.debug", "G.edges.greedy", "G.edges.1", "G.edges.2", "G.edges.skipped",
         "edges.g_out.txt", "edges.fwd.backup.txt", "edges.bkw.backup.txt"]


def _oracle(lib, wd, mlas, ini):
    las = b"G" if mlas else b"G.las"
    return [run_in(wd, lib.oracle_filter, b"G", las, int(mlas), b"G", ini.encode(), b""),
            run_in(wd, lib.oracle_maximal, b"G", las, int(mlas), b"G", ini.encode()),
            run_in(wd, lib.oracle_layout, b"G", las, int(mlas), b"G", b"G", ini.encode())]


def _cli(wd, mlas, ini):
    las = ["--las", "G", "--mlas"] if mlas else ["--las", "G.las"]
    rcs = []
    for sub, extra in (("filter", []), ("maximal", []), ("layout", ["-o", "G"])):
        r = subprocess.run([HINGE, sub, "--db", "G"] + las + ["-x", "G", "--config", ini] + extra, cwd=wd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        rcs.append(r.returncode)
        assert r.returncode == 0, r.stdout.decode()[-2000:]
    return rcs


CASES = [("tiny", False, "", ""), ("tiny_qv", False, "", ""), ("tiny_mlas", True, "", ""), ("tiny_mlas", False, "", ""), ("ties", False, "", ""),
         ("chimera", False, "", ""), ("long_repeat", False, "", ""), ("tspace200", False, "", ""), ("edges", False, "", ""),
         ("edges", False, "length_threshold = 400\n", "del_telomere = 1\ndel_telomeres = 1\n"),
         ("tiny", False, "", "min_connected_component_size = 2\n"),
         ("tiny_qv", False, "ec = 60\nhinge_min_support = 3\nhinge_unbridged = 2\nhinge_min_pileup = 3\n", "del_telomere = 1\ndel_telomeres = 1\nuse_two_matches = 0\n"),
         ("long_repeat", False, "theta2 = 100\naln_threshold = 2500\n", "hinge_tolerance = 400\nmatching_hinge_slack = 500\nmin_connected_component_size = 1\nhinge_slack = 10\n")]


@pytest.mark.parametrize("name,mlas,extra_filter,extra_layout", CASES)
def test_cli_pipeline_matches_oracle(datasets, oracle_lib, tmp_path, name, mlas, extra_filter, extra_layout):
    src, _ = datasets(name)
    wd_o = clone_dataset(src, str(tmp_path / "oracle"))
    wd_h = clone_dataset(src, str(tmp_path / "hip"))
    for wd in (wd_o, wd_h):
        write_ini(os.path.join(wd, "v.ini"), extra_filter=extra_filter, extra_layout=extra_layout)
    assert _oracle(oracle_lib, wd_o, mlas, "v.ini") == [0, 0, 0]
    assert _cli(wd_h, mlas, "v.ini") == [0, 0, 0]
    bad = [f for f in FILES if not filecmp.cmp(os.path.join(wd_o, f), os.path.join(wd_h, f), shallow=False)]
    assert not bad, "differs from the oracle: %s" % bad
    assert os.path.getsize(os.path.join(wd_h, "G.edges.hinges")) > 0 and os.path.getsize(os.path.join(wd_h, "G.max")) > 0
    if not mlas and not extra_filter:
        # ... and `hinge clip` on top of the pipeline's files (hinge_amd/clip.py, parity unpinned: here only that the dispatcher runs
        # it where the reference runs its script, and that identical layout files give identical graphs)
        for wd in (wd_o, wd_h):
            r = subprocess.run([HINGE, "clip", "G.edges.hinges", "G.hinge.list", ".c", "v.ini"], cwd=wd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            assert r.returncode == 0, r.stdout.decode()[-2000:]
        for f in ("G.c.G0.graphml", "G.c.G1.graphml"):
            assert os.path.getsize(os.path.join(wd_h, f)) > 0 and filecmp.cmp(os.path.join(wd_o, f), os.path.join(wd_h, f), shallow=False)


def test_cli_reads_outside_the_overlap_id_range(datasets, oracle_lib, tmp_path):
    """Reads before the first / after the last A id get no .mas line (filter.cpp:515-517,696); filter still
    has defined output, maximal and layout would read uninitialised masks in the reference: refused here."""
    src, d = datasets("orphan_ends")
    assert d.aread.min() == 3 and d.aread.max() == d.n_reads - 4
    wd_o = clone_dataset(src, str(tmp_path / "oracle"))
    wd_h = clone_dataset(src, str(tmp_path / "hip"))
    assert run_in(wd_o, oracle_lib.oracle_filter, b"G", b"G.las", 0, b"G", b"nominal.ini", b"") == 0
    run = lambda *a: subprocess.run([HINGE] + list(a), cwd=wd_h, stdout=subprocess.PIPE, stderr=subprocess.STDOUT).returncode  # noqa: E731
    assert run("filter", "--db", "G", "--las", "G.las", "-x", "G", "--config", "nominal.ini") == 0
    filt = ["G.mas", "G.cmas", "G.repeat.txt", "G.hinges.txt", "G.coverage.txt", "G.cov.flag", "G.self.flag", "G.homologous.txt", "G.filtered.fasta"]
    bad = [f for f in filt if not filecmp.cmp(os.path.join(wd_o, f), os.path.join(wd_h, f), shallow=False)]
    assert not bad, "differs from the oracle: %s" % bad
    assert sum(1 for _ in open(os.path.join(wd_h, "G.mas"))) == d.n_reads - 6
    # strict mode: both sides refuse to guess what the reference's uninitialised fields held
    os.environ["HINGE_STRICT_MAS"] = "1"
    try:
        assert _oracle(oracle_lib, wd_o, False, "nominal.ini")[1:] == [-3, -3]          # (maximal truncates .coverage.txt first)
        assert run("maximal", "--db", "G", "--las", "G.las", "-x", "G", "--config", "nominal.ini") == 2
        assert run("layout", "--db", "G", "--las", "G.las", "-x", "G", "-o", "G", "--config", "nominal.ini") == 2
    finally:
        del os.environ["HINGE_STRICT_MAS"]
    # default: such reads are inactive with mask (0, 0) - what a fresh heap gives the reference - and all stages run
    assert run_in(wd_o, oracle_lib.oracle_filter, b"G", b"G.las", 0, b"G", b"nominal.ini", b"") == 0
    assert _oracle(oracle_lib, wd_o, False, "nominal.ini")[1:] == [0, 0]
    assert run("filter", "--db", "G", "--las", "G.las", "-x", "G", "--config", "nominal.ini") == 0
    assert run("maximal", "--db", "G", "--las", "G.las", "-x", "G", "--config", "nominal.ini") == 0
    assert run("layout", "--db", "G", "--las", "G.las", "-x", "G", "-o", "G", "--config", "nominal.ini") == 0
    bad = [f for f in FILES if not filecmp.cmp(os.path.join(wd_o, f), os.path.join(wd_h, f), shallow=False)]
    assert not bad, "differs from the oracle: %s" % bad


def test_cli_inputs_the_reference_cannot_process(oracle_lib, tmp_path):
    """An empty .las ("No alignments!", exit 1 as filter.cpp:511-514) and a data set without a read of 5000 bp (the
    reference indexes an empty vector, filter.cpp:660-666: the oracle says -3, the executable refuses with an error)."""
    import numpy as np
    from hinge_amd import formats, synth
    d = synth.generate(synth.SynthSpec(genome_len=30_000, coverage=25, len_min=1500, len_max=4000, seed=3))
    wd = str(tmp_path / "short")
    synth.write_dataset(d, wd, "G")
    write_ini(os.path.join(wd, "nominal.ini"))
    assert int(np.max(d.rlen)) < 5000
    assert run_in(wd, oracle_lib.oracle_filter, b"G", b"G.las", 0, b"G", b"nominal.ini", b"") == -3
    r = subprocess.run([HINGE, "filter", "--db", "G", "--las", "G.las", "-x", "G", "--config", "nominal.ini"], cwd=wd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode != 0 and b"the reference is undefined here" in r.stdout, r.stdout.decode()[-1000:]
    # the same DB with an empty .las
    recs = formats.read_las(os.path.join(wd, "G.las"))
    empty = formats.LasRecords(tspace=recs.tspace, rec=recs.rec[:0], trace=recs.trace[:0], trace_off=recs.trace_off[:1])
    formats.write_las(os.path.join(wd, "E.las"), empty)
    for sub, extra in (("filter", []), ("maximal", []), ("layout", ["-o", "G"])):
        r = subprocess.run([HINGE, sub, "--db", "G", "--las", "E.las", "-x", "G", "--config", "nominal.ini"] + extra, cwd=wd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert r.returncode == 1 or (sub != "filter" and r.returncode == 2), (sub, r.returncode, r.stdout.decode()[-500:])


def test_cli_error_behaviour(datasets, tmp_path):
    """Exit codes of the reference's argument / config error paths (filter.cpp:218-226,372-375, hinging.cpp required flags)."""
    src, _ = datasets("tiny")
    wd = clone_dataset(src, str(tmp_path / "w"))
    run = lambda *a: subprocess.run([HINGE] + list(a), cwd=wd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT).returncode  # noqa: E731
    assert run("filter", "--db", "G", "--las", "G.las", "--paf", "x.paf", "-x", "G", "--config", "nominal.ini") == 1
    assert run("filter", "-x", "G", "--config", "nominal.ini") == 1
    assert run("filter", "--db", "G", "--las", "G.las", "-x", "G", "--config", "missing.ini") == 1
    assert run("filter", "--db", "nope", "--las", "G.las", "-x", "G", "--config", "nominal.ini") == 1
    assert run("layout", "--db", "G", "--las", "G.las", "--config", "nominal.ini") == 1          # -x / -o are required
    assert run("filter", "--db", "G", "--las", "G.las", "--bogus", "-x", "G", "--config", "nominal.ini") == 1
    assert run("nosuchcommand") == 1


# ---- FASTA + PAF input (the reference's second input mode, filter.cpp:289-291,499-503) --------------------------------
def _write_paf_inputs(d, wd, shuffle, gz):
    import numpy as np
    from hinge_amd import formats
    os.makedirs(wd, exist_ok=True)
    ext = ".gz" if gz else ""
    formats.write_fasta(os.path.join(wd, "G.fasta" + ext), d.rlen, seed=3, gz=gz)
    order = np.arange(d.novl)
    if shuffle:   # PAF lines need not be grouped by query: the reference files each line under its A read in file order
        order = np.random.default_rng(5).permutation(d.novl)
        # the reference takes r_begin / r_end from the first / last LINE (filter.cpp:515-516): keep the whole id range defined
        first = int(np.nonzero(d.aread[order] == d.aread.min())[0][0])
        order[[0, first]] = order[[first, 0]]
        last = int(np.nonzero(d.aread[order] == d.aread.max())[0][-1])
        order[[-1, last]] = order[[last, -1]]
    formats.write_paf(os.path.join(wd, "G.paf" + ext), d.rlen, d.aread[order], d.bread[order], d.comp[order], d.ab[order], d.ae[order],
                      d.bb[order], d.be[order], gz=gz)
    return "G.fasta" + ext, "G.paf" + ext


@pytest.mark.parametrize("name,shuffle,gz", [("tiny", False, False), ("long_repeat", False, True), ("chimera", True, False), ("long_repeat", True, False)])
def test_cli_paf_pipeline_matches_oracle(datasets, oracle_lib, tmp_path, name, shuffle, gz):
    _, d = datasets(name)
    wd_o, wd_h = str(tmp_path / "oracle"), str(tmp_path / "hip")
    for wd in (wd_o, wd_h):
        fa, paf = _write_paf_inputs(d, wd, shuffle, gz)
        write_ini(os.path.join(wd, "v.ini"))
    rcs = [run_in(wd_o, oracle_lib.oracle_filter_paf, fa.encode(), paf.encode(), b"G", b"v.ini"),
           run_in(wd_o, oracle_lib.oracle_maximal_paf, fa.encode(), paf.encode(), b"G", b"v.ini"),
           run_in(wd_o, oracle_lib.oracle_layout_paf, fa.encode(), paf.encode(), b"G", b"G", b"v.ini")]
    assert rcs == [0, 0, 0]
    for sub, extra in (("filter", []), ("maximal", []), ("layout", ["-o", "G"])):
        r = subprocess.run([HINGE, sub, "--fasta", fa, "--paf", paf, "-x", "G", "--config", "v.ini"] + extra, cwd=wd_h, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT)
        assert r.returncode == 0, r.stdout.decode()[-2000:]
    bad = [f for f in FILES if not filecmp.cmp(os.path.join(wd_o, f), os.path.join(wd_h, f), shallow=False)]
    assert not bad, "differs from the oracle: %s" % bad
    assert os.path.getsize(os.path.join(wd_h, "G.edges.hinges")) > 0 and os.path.getsize(os.path.join(wd_h, "G.max")) > 0
    if name == "long_repeat" and not shuffle:
        assert sum((len(l.split()) - 1) // 2 for l in open(os.path.join(wd_h, "G.hinges.txt"))) > 0


def test_cli_paf_bad_read_names(datasets, tmp_path):
    _, d = datasets("tiny")
    wd = str(tmp_path / "w")
    fa, paf = _write_paf_inputs(d, wd, False, False)
    write_ini(os.path.join(wd, "v.ini"))
    txt = open(os.path.join(wd, paf)).read().replace("synth/1/", "synth_1_", 1)   # a name without "/id/": the reference dereferences NULL
    open(os.path.join(wd, paf), "w").write(txt)
    r = subprocess.run([HINGE, "filter", "--fasta", fa, "--paf", paf, "-x", "G", "--config", "v.ini"], cwd=wd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 1


@pytest.mark.parametrize("name,mlas", [("long_repeat", False), ("tiny_mlas", True)])
def test_cli_filter_restrictreads(datasets, oracle_lib, tmp_path, name, mlas):
    """--restrictreads FILE (filter.cpp:300-316,680-694,767-773): the listed reads and their neighbours keep their masks."""
    src, d = datasets(name)
    wd_o = clone_dataset(src, str(tmp_path / "oracle"))
    wd_h = clone_dataset(src, str(tmp_path / "hip"))
    keep = [5, 17, 40, d.n_reads - 3]
    for wd in (wd_o, wd_h):
        write_ini(os.path.join(wd, "v.ini"))
        open(os.path.join(wd, "keep.txt"), "w").write("".join("%d\n" % k for k in keep))
    las = b"G" if mlas else b"G.las"
    assert run_in(wd_o, oracle_lib.oracle_filter, b"G", las, int(mlas), b"G", b"v.ini", b"keep.txt") == 0
    args = ["--las", "G", "--mlas"] if mlas else ["--las", "G.las"]
    r = subprocess.run([HINGE, "filter", "--db", "G"] + args + ["-x", "G", "--config", "v.ini", "--restrictreads", "keep.txt"], cwd=wd_h,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    files = ["G.mas", "G.cmas", "G.repeat.txt", "G.hinges.txt", "G.coverage.txt", "G.cov.flag", "G.self.flag"]
    bad = [f for f in files if not filecmp.cmp(os.path.join(wd_o, f), os.path.join(wd_h, f), shallow=False)]
    assert not bad, "differs from the oracle: %s" % bad
    emptied = sum(1 for l in open(os.path.join(wd_h, "G.mas")) if l.split()[1] == l.split()[2])
    assert 0 < emptied < d.n_reads


@pytest.mark.parametrize("ranks", [2, 3, 8])
def test_cli_mlas_on_several_ranks(oracle_lib, tmp_path, ranks):
    """`hinge filter --mlas` with the parts spread over HINGE_RANKS ranks (one host thread + one context each; on a node with
    several GPUs one per device, here all on the one GPU): waves of `ranks` parts run concurrently, the running MIN_COV and the
    mask table are exchanged between them, and every output file equals the oracle's sequential --mlas loop byte for byte.
    8 blocks (BASELINE config 4's shape, scaled down), so 2 and 3 ranks need several waves and 3 leaves a ragged last wave."""
    import dataclasses
    from hinge_amd import synth
    d = synth.generate(dataclasses.replace(synth.CONFIGS["cfg4_yeast"], genome_len=500_000))
    assert d.spec.n_blocks == 8
    src = str(tmp_path / "src")
    synth.write_dataset(d, src, "G", write_bases=False)
    write_ini(os.path.join(src, "nominal.ini"))
    wd_o = clone_dataset(src, str(tmp_path / "oracle"))
    wd_h = clone_dataset(src, str(tmp_path / "hip"))
    assert run_in(wd_o, oracle_lib.oracle_filter, b"G", b"G", 1, b"G", b"nominal.ini", b"") == 0
    assert run_in(wd_o, oracle_lib.oracle_maximal, b"G", b"G", 1, b"G", b"nominal.ini") == 0
    assert run_in(wd_o, oracle_lib.oracle_layout, b"G", b"G", 1, b"G", b"G", b"nominal.ini") == 0
    # get_maximal_reads: the parts' classification side by side, containment in part order; hinging: the parts' front halves
    # (grouping, packing, classification) side by side, matches consumed in part order, selection in one K5 launch
    for sub, extra in (("filter", []), ("maximal", []), ("layout", ["-o", "G"])):
        r = subprocess.run([HINGE, sub, "--db", "G", "--las", "G", "--mlas", "-x", "G", "--config", "nominal.ini"] + extra, cwd=wd_h,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=dict(os.environ, HINGE_RANKS=str(ranks)))
        assert r.returncode == 0, r.stdout.decode()[-2000:]
    bad = [f for f in FILES if not filecmp.cmp(os.path.join(wd_o, f), os.path.join(wd_h, f), shallow=False)]
    assert not bad, "differs from the oracle: %s" % bad
    assert 0 < sum(1 for _ in open(os.path.join(wd_o, "G.max"))) < d.n_reads
    assert os.path.getsize(os.path.join(wd_o, "G.edges.hinges")) > 0
    assert sum((len(l.split()) - 1) // 2 for l in open(os.path.join(wd_o, "G.hinges.txt"))) > 0


@pytest.mark.parametrize("name,mlas", [("tiny", False), ("tiny_qv", False), ("long_repeat", False), ("chimera", False), ("tiny_mlas", True)])
def test_hinge_pipeline_in_one_process(datasets, oracle_lib, tmp_path, name, mlas):
    """`hinge pipeline` = filter, maximal and layout in ONE process (one HIP start-up, one ingest of a single .las, the part handed
    from stage to stage): every file byte-identical to the oracle's - i.e. to the three separate runs, which the case list above
    holds to the same files.  A failing first stage ends the run with that stage's exit code."""
    src, _ = datasets(name)
    wd_o = clone_dataset(src, str(tmp_path / "oracle"))
    wd_h = clone_dataset(src, str(tmp_path / "hip"))
    for wd in (wd_o, wd_h):
        write_ini(os.path.join(wd, "v.ini"))
    assert _oracle(oracle_lib, wd_o, mlas, "v.ini") == [0, 0, 0]
    las = ["--las", "G", "--mlas"] if mlas else ["--las", "G.las"]
    r = subprocess.run([HINGE, "pipeline", "--db", "G"] + las + ["-x", "G", "--config", "v.ini", "-o", "G"], cwd=wd_h, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    bad = [f for f in FILES if not filecmp.cmp(os.path.join(wd_o, f), os.path.join(wd_h, f), shallow=False)]
    assert not bad, "differs from the oracle: %s" % bad
    # no config file: `hinge filter` returns 1 ("Can't load"), and so does the pipeline, before any later stage runs
    r = subprocess.run([HINGE, "pipeline", "--db", "G"] + las + ["-x", "Q", "--config", "missing.ini", "-o", "Q"], cwd=wd_h, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 1 and not os.path.exists(os.path.join(wd_h, "Q.max"))


@pytest.mark.parametrize("name,mlas", [("tiny", False), ("tiny_mlas", True)])
def test_cli_rows_through_a_one_rank_communicator(datasets, oracle_lib, tmp_path, name, mlas):
    """Round 6: `hinge maximal` / `hinge layout` hand their ranks' rows (containment candidates, classified matches) to the sequential
    pass through hinge_comm_allgather_rows.  A 1-GPU box has one rank: HINGE_COMM_ONE_RANK=1 sends the rows through a one-rank
    communicator (both grouped ncclAllGathers run); the logs must say so and every file must still equal the oracle's."""
    src, _ = datasets(name)
    wd_o = clone_dataset(src, str(tmp_path / "oracle"))
    wd_h = clone_dataset(src, str(tmp_path / "hip"))
    assert _oracle(oracle_lib, wd_o, mlas, "nominal.ini") == [0, 0, 0]
    las = ["--las", "G", "--mlas"] if mlas else ["--las", "G.las"]
    env = dict(os.environ, HINGE_COMM_ONE_RANK="1", HINGE_RANKS="1")
    for sub, extra, what in (("filter", [], None), ("maximal", [], "containment candidates"), ("layout", ["-o", "G"], "classified matches")):
        r = subprocess.run([HINGE, sub, "--db", "G"] + las + ["-x", "G", "--config", "nominal.ini"] + extra, cwd=wd_h, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        text = r.stdout.decode()
        assert r.returncode == 0, text[-2000:]
        if what:
            assert "%s over RCCL" % what in text and "exchanged over RCCL" in text, text[-1500:]
    bad = [f for f in FILES if not filecmp.cmp(os.path.join(wd_o, f), os.path.join(wd_h, f), shallow=False)]
    assert not bad, "differs from the oracle: %s" % bad
