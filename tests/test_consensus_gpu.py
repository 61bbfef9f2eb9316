"""`hinge consensus` on the GPU (hinge_amd/bin/consensus over hinge_consensus_*) against the reference's OWN program
(oracle/_ref/consensus, which travels to the GPU box) and its golden digests: FASTA and stdout byte for byte; the recovered
indel lists against the oracle's, alignment by alignment."""
import ctypes
import os
import struct

import numpy as np
import pytest

import consensus_common as cc

pytestmark = pytest.mark.gpu
NAMES = sorted(cc.GOLDEN)


@pytest.mark.parametrize("name", NAMES)
def test_consensus_matches_the_reference_program(tmp_path, name):
    wd = str(tmp_path)
    cc.make(name, wd)
    fasta, out = cc.run_product(wd)
    g = cc.GOLDEN[name]
    ref = cc.run_reference(wd)
    if ref is not None:
        assert ref[0] == fasta, "FASTA differs from the reference program's"
        assert ref[1] == out, "stdout differs from the reference program's"
    assert cc.sha(fasta) == g["fasta_sha256"]
    assert cc.sha(out) == g["stdout_sha256"]


def test_indel_lists_match_the_oracle(oracle_lib, tmp_path):
    """recoverAlignment's result per alignment (LAInterface.cpp:4125-4244): the same entries in the same order."""
    from hinge_amd import capi, formats
    wd = str(tmp_path)
    d = cc.make("cns_noisy", wd)
    cc.run_oracle(oracle_lib, wd, dump="ora.dump")
    raw = open(os.path.join(wd, "ora.dump"), "rb").read()
    want = {}
    p = 0
    while p < len(raw):
        contig, pos, off, n = struct.unpack_from("<4i", raw, p)
        p += 16
        want[pos] = (off, np.frombuffer(raw, dtype=np.int32, count=n, offset=p).copy())
        p += 4 * n
    ctx = capi.Context(0)
    cns = capi.Consensus(ctx, os.path.join(wd, "draft"), os.path.join(wd, "reads"))
    las = formats.read_las(os.path.join(wd, "draft.reads.las"))
    picks = sorted(want)
    cns.run(las, picks)
    offs = cns.offsets()
    for k, pos in enumerate(picks):
        got = cns.indels(k)
        assert np.array_equal(got, want[pos][1]), "alignment %d: indel list differs" % pos
        assert offs[k] == want[pos][0]


def test_inconsistent_diffs_are_refused(tmp_path):
    """A trace whose recorded diffs are too small for its bases: the reference overruns the arrays it sized from them
    (LAInterface.cpp:3444-3466); the library returns HINGE_E_RANGE."""
    from hinge_amd import capi, formats
    wd = str(tmp_path)
    cc.make("cns_noisy", wd)
    ctx = capi.Context(0)
    cns = capi.Consensus(ctx, os.path.join(wd, "draft"), os.path.join(wd, "reads"))
    las = formats.read_las(os.path.join(wd, "draft.reads.las"))
    las.trace[0::2] = 0          # every segment claims to be error-free
    with pytest.raises(capi.HingeError) as e:
        cns.run(las, list(range(len(las.rec))))
    assert e.value.code == capi.HINGE_E_RANGE


def test_inconsistent_trace_advances_are_refused_by_the_device_table(tmp_path):
    """The segment table is made on the device (k_cns_segments, round 5): B advances that run past bepos / the read's end -
    trace points inconsistent with the alignment's coordinates - are refused with HINGE_E_RANGE as the host loop refused them, the
    context stays usable (the same alignments with their real trace then run and give the stored indel lists), and an alignment
    without any trace pair is ONE segment (computeTracePTS's last add, LAInterface.cpp:3470-3500)."""
    from hinge_amd import capi, formats
    wd = str(tmp_path)
    cc.make("cns_tiny", wd)
    ctx = capi.Context(0)
    cns = capi.Consensus(ctx, os.path.join(wd, "draft"), os.path.join(wd, "reads"))
    las = formats.read_las(os.path.join(wd, "draft.reads.las"))
    picks = list(range(len(las.rec)))
    good = las.trace.copy()
    las.trace[1::2] = 250                                  # every segment advances B by 250: far past bepos
    with pytest.raises(capi.HingeError) as e:
        cns.run(las, picks)
    assert e.value.code == capi.HINGE_E_RANGE and "inconsistent" in str(e.value)
    las.trace[:] = good
    cns.run(las, picks)
    first = [cns.indels(k).copy() for k in range(min(len(picks), 20))]
    cns.run(las, picks)                                    # a second call on the same context: same lists (buffers are reused)
    for k, w in enumerate(first):
        assert np.array_equal(cns.indels(k), w)
    ctx.close()


def test_consensus_cli_error_paths(tmp_path):
    """An unreadable config: "Can't load <name>" on stdout and exit code 1 as consensus.cpp:88-93 (the output file exists by then:
    the ofstream is the program's first statement); too few arguments: usage, exit 1 (the reference reads argv[5] unchecked)."""
    import subprocess
    wd = str(tmp_path)
    cc.make("cns_tiny", wd)
    r = subprocess.run([cc.EXE, "draft", "reads", "draft.reads.las", "o.fasta", "missing.ini"], cwd=wd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and r.stdout == b"Can't load missing.ini\n" and os.path.getsize(os.path.join(wd, "o.fasta")) == 0
    if os.path.exists(cc.REF_BIN):
        q = subprocess.run([cc.REF_BIN, "draft", "reads", "draft.reads.las", "p.fasta", "missing.ini"], cwd=wd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert (q.returncode, q.stdout) == (r.returncode, r.stdout)
    assert subprocess.run([cc.EXE, "draft", "reads"], cwd=wd, stdout=subprocess.PIPE, stderr=subprocess.PIPE).returncode == 1


def test_consensus_through_the_dispatcher_and_min_length(oracle_lib, tmp_path):
    """`hinge consensus ...` (src/hinge:33-35) and another [consensus] min_length: fewer alignments vote, the files still equal the
    reference program's."""
    import subprocess
    wd = str(tmp_path)
    cc.make("cns_small", wd)
    with open(os.path.join(wd, "nominal.ini"), "w") as f:
        f.write("[consensus]\nmin_length = 3000;\n")
    hinge = os.path.join(cc.ROOT, "hinge_amd", "bin", "hinge")
    r = subprocess.run([hinge, "consensus", "draft", "reads", "draft.reads.las", "hip.fasta", "nominal.ini"], cwd=wd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-1000:]
    ref = cc.run_reference(wd) or cc.run_oracle(oracle_lib, wd)      # (the reference's own program where it was built, else the restatement pinned to it)
    assert open(os.path.join(wd, "hip.fasta"), "rb").read() == ref[0] and r.stdout == ref[1]


def test_consensus_at_bench_size(oracle_lib, tmp_path):
    """An E. coli-sized draft (4 contigs of ~1.1 Mb at 30x: 18 820 alignments, 1.33 M trace-point segments, 131 M aligned bases)
    through the executable against the reference's own program on the same files: 4.5 MB of FASTA and the whole stdout text
    (18 820 chop offsets among it), byte for byte."""
    wd = str(tmp_path)
    d = cc.make("cns_bench", wd)
    assert d.n_alignments > 15_000
    fasta, out = cc.run_product(wd)
    ref = cc.run_reference(wd) or cc.run_oracle(oracle_lib, wd)
    assert fasta == ref[0] and out == ref[1]
    assert len(fasta) > 4_000_000


def _sharded_worker(rank, world, port, wd, ret):
    import torch.distributed as dist
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hinge_amd import consensus
        rc, text = consensus.run_consensus(os.path.join(wd, "draft"), os.path.join(wd, "reads"), os.path.join(wd, "draft.reads.las"),
                                           os.path.join(wd, "sharded.fasta"), os.path.join(wd, "nominal.ini"), device=0)
        dist.barrier()
        ret.put((rank, rc, text))
    finally:
        dist.destroy_process_group()


def test_python_driver_and_contig_sharding_on_the_gpu(oracle_lib, tmp_path):
    """hinge_amd/consensus.py over the HIP backend: one rank, and two ranks that share the contigs (both on this box's one GPU, gloo
    for the final gather): the reference program's FASTA and its whole stdout text."""
    import torch.multiprocessing as mp
    from hinge_amd import consensus
    wd = str(tmp_path)
    cc.make("cns_midsize", wd)
    ref = cc.run_reference(wd) or cc.run_oracle(oracle_lib, wd)
    rc, text = consensus.run_consensus(os.path.join(wd, "draft"), os.path.join(wd, "reads"), os.path.join(wd, "draft.reads.las"), os.path.join(wd, "py.fasta"),
                                       os.path.join(wd, "nominal.ini"), device=0)
    assert rc == 0 and open(os.path.join(wd, "py.fasta"), "rb").read() == ref[0] and text == ref[1]
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    mp.spawn(_sharded_worker, args=(2, 37950 + os.getpid() % 40, wd, ret), nprocs=2, join=True)
    got = sorted(ret.get() for _ in range(2))
    assert [g[1] for g in got] == [0, 0] and got[0][2] == ref[1] and got[1][2] == ref[1]
    assert open(os.path.join(wd, "sharded.fasta"), "rb").read() == ref[0]


def test_contigs_in_several_batches(oracle_lib, tmp_path):
    """HINGE_CNS_SLOT_BUDGET small enough for one contig per hinge_consensus_run (what a draft of hundreds of Mb triggers by itself):
    the files are those of the single run."""
    wd = str(tmp_path)
    cc.make("cns_small", wd)
    ref = cc.run_reference(wd) or cc.run_oracle(oracle_lib, wd)
    import subprocess
    r = subprocess.run([cc.EXE, "draft", "reads", "draft.reads.las", "b.fasta", "nominal.ini"], cwd=wd, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, HINGE_CNS_SLOT_BUDGET="1", HINGE_HOST_TIMING="1"))
    assert r.returncode == 0, r.stderr.decode()[-1000:]
    assert b"4 contig batch(es)" in r.stderr
    assert open(os.path.join(wd, "b.fasta"), "rb").read() == ref[0] and r.stdout == ref[1]
