"""The multi-process product path on a real GPU: dist.ShardedFilter / dist.PartBatch over dist.HipBackend (HIP kernels through
the C ABI) with REAL exchanges between ranks, compared with the CPU oracle's files.

A 1-GPU box cannot form an RCCL communicator of several ranks (one device per rank), so
  * the multi-rank cases run one process per rank on the SAME device with the gloo transport (dist.py stages the device tensors
    through host memory for it): every kernel, every table, every id mapping and every exchange's DATA FLOW is the product's;
  * the RCCL calls themselves (in-place all_gather_into_tensor, all_reduce, asynchronous handles, stream-side waits) run with
    one rank and HINGE_FORCE_COLLECTIVES=1.
B reads cross blocks in all of them (DBsplit data sets), so hinge calling only gets the oracle's answers if exchange 2 delivered
the other ranks' masks and exchange 1 the global median.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

import conftest

from conftest import run_in, write_ini

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pairs(path):
    rows = []
    for line in open(path):
        tok = line.split()
        for j in range(1, len(tok) - 1, 2):
            rows.append((int(tok[0]), int(tok[j]), int(tok[j + 1])))
    return np.array(rows, np.int32).reshape(-1, 3)


def _table(path):
    return np.loadtxt(path, dtype=np.int64).reshape(-1, 3)


def _sharded_filter_worker(rank, world, port, backend, wd, first, mode, median, force, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if force:
        os.environ["HINGE_FORCE_COLLECTIVES"] = "1"
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hinge_amd import capi, formats
        from hinge_amd.config import IniFile, filter_params
        from hinge_amd.dist import BlockTable, Exchange, HipBackend, ShardedFilter
        rlen = formats.read_db_index(os.path.join(wd, "G"))["rlen"]
        lo, hi = first[rank], first[rank + 1]
        name = os.path.join(wd, "G.%d.las" % (rank + 1)) if world > 1 else os.path.join(wd, "G.las")
        recs = formats.read_las(name)                                   # this rank's block only
        pile = formats.pileups_from_las(recs, rlen)
        assert int(recs.rec["aread"][0]) >= lo and int(recs.rec["aread"][-1]) < hi
        b = pile.b_flag & np.uint32(0x7FFFFFFF)
        if world > 1:
            assert ((b < lo) | (b >= hi)).mean() > 0.3, "the data set has no B reads outside the rank's block"
        P = filter_params(IniFile(os.path.join(wd, "nominal.ini")), False)
        ctx = capi.Context(0)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        be = HipBackend(ctx, P, rlen, None, lo, hi - 1, t(pile.row_ptr), t(pile.a_span), t(pile.b_span), t(pile.b_flag.view(np.int32)),
                        last_a=int(recs.rec["aread"][-1]))
        job = ShardedFilter(be, Exchange(BlockTable(first), dev), mode=mode, median=median)
        assert job.x.force == bool(force)
        rows = job.step(fetch_hinges=True).cpu().numpy()

        # ---- against the oracle's files ----
        want_mask = _table(os.path.join(wd, "G.mas"))[:, 1:].astype(np.int32)
        got_mask = job.mask.cpu().numpy()
        visible = want_mask.copy()
        if mode == "mlas":
            visible[hi:] = 0                                            # later parts are not masked yet when part p runs
        assert np.array_equal(got_mask, visible), "mask table after exchange 2"
        mask, cmask, flags = ctx.get_masks()
        assert np.array_equal(mask, want_mask[lo:hi])
        assert np.array_equal(cmask, _table(os.path.join(wd, "G.cmas"))[lo:hi, 1:])
        off, pos, typ, ish = ctx.get_annotations()
        if mode == "merged" or rank == 0:                               # --mlas closes .repeat.txt after the first part (filter.cpp:1086)
            lines = open(os.path.join(wd, "G.repeat.txt")).read().splitlines()
            for k, i in enumerate(range(lo, hi)):
                tok = [int(v) for v in lines[i].split()]
                assert tok[0] == i
                s, e = off[k], off[k + 1]
                assert tok[1:] == [v for q in range(s, e) for v in (int(pos[q]), int(typ[q]))], "annotations of read %d" % i
        want_rows = _pairs(os.path.join(wd, "G.hinges.txt"))
        assert np.array_equal(rows, want_rows), "the global hinge list (exchange 3): %d rows, the oracle has %d" % (len(rows), len(want_rows))
        ret[rank] = int(len(rows))
    finally:
        dist.destroy_process_group()


def _spawn(fn, world, args):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=fn, args=(r, world) + args + (ret,)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    for p in procs:
        if p.is_alive():
            p.kill()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert len(ret) == world
    return dict(ret)


def _dataset(tmp_path, oracle_lib, mlas, n_blocks=3):
    import dataclasses
    from hinge_amd import synth
    d = synth.generate(dataclasses.replace(synth.CONFIGS["tiny_mlas"], n_blocks=n_blocks))
    wd = str(tmp_path / "data")
    synth.write_dataset(d, wd, "G")
    write_ini(os.path.join(wd, "nominal.ini"))
    assert run_in(wd, oracle_lib.oracle_filter, b"G", b"G" if mlas else b"G.las", int(mlas), b"G", b"nominal.ini", b"") == 0
    return wd, d


@pytest.mark.parametrize("mode,median", [("merged", "hist"), ("merged", "gather"), ("mlas", "gather")])
def test_sharded_filter_hip_three_ranks(oracle_lib, tmp_path, mode, median):
    """Three ranks (processes), one block each, B reads everywhere: masks, annotations and the gathered hinge list equal the
    oracle's .mas / .cmas / .repeat.txt / .hinges.txt of ONE run over all blocks (merged .las, or the --mlas loop)."""
    wd, d = _dataset(tmp_path, oracle_lib, mode == "mlas")
    port = conftest.free_port()
    ret = _spawn(_sharded_filter_worker, 3, (port, "gloo", wd, list(d.block_first), mode, median, False))
    assert len(set(ret.values())) == 1 and ret[0] > 0


@pytest.mark.parametrize("median", ["hist", "gather"])
def test_sharded_filter_hip_rccl_forced(oracle_lib, tmp_path, median):
    """One rank, the collectives forced: the same checks with every exchange going through RCCL."""
    wd, d = _dataset(tmp_path, oracle_lib, False, n_blocks=1)
    port = conftest.free_port()
    ret = _spawn(_sharded_filter_worker, 1, (port, "nccl", wd, [0, d.n_reads], "merged", median, True))
    assert ret[0] > 0


# ---- sharded `hinge layout` -------------------------------------------------------------------------------------------------
LAYOUT_FILES = [".garbage.txt", ".killed.hinges", ".hgraph", ".hinge.list", ".edges.hinges", ".edges.hinges2", ".edges.skipped", ".deadends.txt"]


def _layout_worker(rank, world, port, wd, first, mlas, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hinge_amd import capi, formats
        from hinge_amd import layout as L
        from hinge_amd.config import IniFile
        from hinge_amd.dist import BlockTable, Exchange, HipLayoutBackend, ShardedLayout
        rlen = formats.read_db_index(os.path.join(wd, "G"))["rlen"]
        n = len(rlen)
        eff = np.zeros((n, 2), np.int64)
        for line in open(os.path.join(wd, "G.mas")):
            i, s, e = (int(t) for t in line.split())
            eff[i] = (s, e)
        maximal = np.zeros(n, bool)
        for line in open(os.path.join(wd, "G.max")):
            maximal[int(line)] = True
        P = L.LayoutParams.from_ini(IniFile(os.path.join(wd, "nominal.ini")))
        repeats = L.read_pairs_file(os.path.join(wd, "G.repeat.txt"), n)
        hinges = L.read_pairs_file(os.path.join(wd, "G.hinges.txt"), n)
        recs = formats.read_las(os.path.join(wd, "G.%d.las" % (rank + 1)) if mlas else os.path.join(wd, "G.las"))
        pile = formats.pileups_from_las(recs, rlen)
        tb = 1 if recs.tspace <= formats.TRACE_XOVR else 2
        be = HipLayoutBackend(capi.Context(0), rlen, eff, pile, recs.trace, recs.trace_off[:-1][pile.las_index], recs.rec["tlen"][pile.las_index], tb)
        job = ShardedLayout(be, Exchange(BlockTable(first), dev), P, eff, maximal, repeats, hinges)
        files = job.step()
        bad = [f for f in LAYOUT_FILES if files[f] != open(os.path.join(wd, "G" + f)).read().split("\n")[:-1]]
        assert not bad, "rank %d: %s differ from the oracle's" % (rank, bad)
        ret[rank] = len(files[".edges.hinges"])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("tiny_mlas", 3), ("tiny_mlas", 1), ("ties", 1), ("chimera", 1), ("tspace200", 1), ("edges", 1)])
def test_sharded_layout_hip(oracle_lib, tmp_path, name, world):
    """`hinge layout` as dist.ShardedLayout over HipLayoutBackend (k_trim_classify, k_matching_position, k_select_edges): three
    ranks with one block each (exchanges 5-7 for real), and single-block runs on data sets with heavy ties, chimeric reads,
    two-byte traces and A == B records; eight output files equal the oracle's."""
    import dataclasses
    from hinge_amd import synth
    d = synth.generate(dataclasses.replace(synth.CONFIGS[name], n_blocks=world if world > 1 else 1))
    wd = str(tmp_path / "data")
    synth.write_dataset(d, wd, "G")
    write_ini(os.path.join(wd, "nominal.ini"))
    mlas = world > 1
    las = b"G" if mlas else b"G.las"
    assert run_in(wd, oracle_lib.oracle_filter, b"G", las, int(mlas), b"G", b"nominal.ini", b"") == 0
    assert run_in(wd, oracle_lib.oracle_maximal, b"G", las, int(mlas), b"G", b"nominal.ini") == 0
    assert run_in(wd, oracle_lib.oracle_layout, b"G", las, int(mlas), b"G", b"G", b"nominal.ini") == 0
    n_edges = len(open(os.path.join(wd, "G.edges.hinges")).read().split("\n")) - 1
    assert n_edges > 20
    first = list(d.block_first) if mlas else [0, d.n_reads]
    ret = _spawn(_layout_worker, world, (35300 + (os.getpid() % 1500) + 7 * world + len(name), wd, first, mlas))
    assert all(v == n_edges for v in ret.values())


# ---- PartBatch: several parts per rank, exchanges batched (what bench.py runs) ---------------------------------------------
BATCH_INI = ("[filter]\nlength_threshold = 1000;\naln_threshold = 1000;\nmin_cov = 5;\ncut_off = 300;\ntheta = 300;\n"
             "[layout]\nhinge_slack = 1000\nmin_connected_component_size = 8\n")


def _batch_worker(rank, world, port, backend, workload, R, G, force, want, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if force:
        os.environ["HINGE_FORCE_COLLECTIVES"] = "1"
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hinge_amd import benchsets, synth
        from hinge_amd.config import default_filter_params
        from hinge_amd.dist import resident_batch
        base = synth.CONFIGS[workload]
        parts = [benchsets.rank_part(base, world, rank, p) for p in range(R)]
        batch, ctxs = resident_batch(parts, default_filter_params(), dev, gather_groups=G, pad=3)
        assert batch.collectives
        batch.settle()
        batch.step()                                                    # a second pass over warm buffers gives the same answer
        batch.status()
        lists = [t.cpu().numpy() for t in batch.hinge_lists()]
        got = []
        for p in range(R):
            per_rank = []
            for r in range(world):
                lo = batch.id_base(p, r)
                loc = lists[p][(lists[p][:, 0] >= lo) & (lists[p][:, 0] < lo + batch.S)].astype(np.int64)
                loc[:, 0] -= lo
                per_rank.append({"hinges": int(len(loc)), "digest": benchsets.digest(loc)})
            got.append(per_rank)
        assert got == want, "hinges per part and rank:\n%s\nthe CPU oracle's:\n%s" % (got, want)
        ret[rank] = batch.table_checksums()
    finally:
        dist.destroy_process_group()


def _oracle_expectation(oracle_lib, workload, world, R):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_bench_expect as mbe
    from hinge_amd import benchsets, synth
    base = synth.CONFIGS[workload]
    quiet = lambda *a: None
    want = []
    for p in range(R):
        if world == 1:
            d = synth.generate(benchsets.part_spec(base, 1, 0, p)[0])
            rows, _, _ = mbe.run_oracle(oracle_lib, d, BATCH_INI)
            want.append([mbe.entry(rows, 0, d.n_reads)])
        else:
            team = mbe.team_runs(oracle_lib, base, world // 2, p, BATCH_INI, quiet)
            want.append(mbe.world_entries(oracle_lib, team, world, BATCH_INI, 5, quiet)["ranks"])
    assert sum(e["hinges"] for pr in want for e in pr) > 100
    return want


@pytest.mark.parametrize("world,G", [(2, 1), (2, 2), (4, 1)])
def test_part_batch_hip_ranks_share_one_gpu(oracle_lib, world, G):
    """bench.py's N > 1 path in small: teams of two ranks, 3 parts per rank, one all-reduce + G all-gathers per step; every
    part of every rank gives the oracle's hinges (count + row digest), and all ranks end with the same mask tables."""
    R = 3
    want = _oracle_expectation(oracle_lib, "chimera", world, R)
    port = conftest.free_port()
    ret = _spawn(_batch_worker, world, (port, "gloo", "chimera", R, G, False, want))
    assert all(v == ret[0] for v in ret.values()), "mask tables differ between ranks"


def test_part_batch_hip_rccl_forced(oracle_lib):
    R = 3
    want = _oracle_expectation(oracle_lib, "chimera", 1, R)
    for G in (1, 2):
        _spawn(_batch_worker, 1, (34900 + (os.getpid() % 1500) + G, "nccl", "chimera", R, G, True, want))


def test_part_batch_mixed_parts_share_launches(oracle_lib):
    """One rank, no collectives: four very different resident parts (a 450x data set whose pile-ups need the large instance of
    k_hinge_call, heavy ties, chimeric reads, a small one) go through ONE median launch and ONE launch per hinge kernel; every
    part gives its own oracle answer, pass after pass."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_bench_expect as mbe
    from hinge_amd import benchsets, synth
    from hinge_amd.config import default_filter_params
    from hinge_amd.dist import resident_batch
    names = ["deep", "ties", "chimera", "tiny"]
    parts, want = [], []
    for name in names:
        d = synth.generate(synth.CONFIGS[name])
        parts.append(benchsets.rank_part(synth.CONFIGS[name], 1, 0, 0, data=d))
        rows, _, _ = mbe.run_oracle(oracle_lib, d, BATCH_INI)
        want.append(mbe.entry(rows, 0, d.n_reads))
    assert all(w["hinges"] > 0 for w in want)
    dev = torch.device("cuda", 0)
    batch, ctxs = resident_batch(parts, default_filter_params(), dev, pad=5)
    assert not batch.collectives
    assert ctxs[0].pileup_facts()[0] > 2048, "the deep part should need the large k_hinge_call instance"
    for c in ctxs:                                   # the synchronous entry points size every part's buffers once
        c.filter_stats_median(default_filter_params(), fetch=True)
        c.filter_mask_annotate(default_filter_params())
        c.filter_hinges(default_filter_params())
    for rep in range(3):
        batch.step()
        batch.status()
        lists = [t.cpu().numpy() for t in batch.hinge_lists()]
        for p in range(len(names)):
            loc = lists[p].astype(np.int64)
            loc[:, 0] -= batch.id_base(p)
            got = {"hinges": int(len(loc)), "digest": benchsets.digest(loc)}
            assert got == want[p], (rep, names[p], got, want[p])
        used = [c.heavy_items() for c in ctxs]
    assert used[0][1] > 0, "no annotation of the deep part reached the large instance: %s" % (used,)


# ---- bench.py itself with two ranks, and with a deliberately broken exchange 2 ---------------------------------------------
def _run_bench(extra_env, workload="chimera", nproc=2):
    env = dict(os.environ)
    env.update({"HINGE_BENCH_BACKEND": "gloo", "HINGE_BENCH_ONE_DEVICE": "1", "HINGE_BENCH_EXPECT": extra_env.pop("expect")})
    env.update(extra_env)
    port = conftest.free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "3", "--warmup", "1", "--workload", workload, "--parts", "2"]
    return subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)


def test_bench_two_ranks_asserts_results(oracle_lib, tmp_path):
    """bench.py --gpus 2 (test rig: both ranks on device 0, gloo) on a small workload: passes against the oracle's expectations
    and FAILS when exchange 2 is corrupted (other ranks' mask rows zeroed after the all-gather)."""
    import json
    want = _oracle_expectation(oracle_lib, "chimera", 2, 2)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from hinge_amd import benchsets, synth
    exp = {"chimera": {"worlds": {"2": {}}}}
    for p in range(2):
        d = synth.generate(benchsets.part_spec(synth.CONFIGS["chimera"], 2, 0, p)[0])
        reads = [d.block_first[k + 1] - d.block_first[k] for k in range(2)]
        records = [int(np.sum((d.aread >= d.block_first[k]) & (d.aread < d.block_first[k + 1]))) for k in range(2)]
        exp["chimera"]["worlds"]["2"][str(p)] = {"ranks": want[p], "reads": reads, "records": records}
    path = str(tmp_path / "expect.json")
    json.dump(exp, open(path, "w"))
    r = _run_bench({"expect": path})
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["checks"]["parts_checked"] == 2 and line["checks"]["hinges_and_digests_match_cpu_oracle"]
    assert line["config"]["collectives_per_step"] == 2
    bad = _run_bench({"expect": path, "HINGE_TEST_CORRUPT_GATHER": "1"})
    assert bad.returncode != 0 and (b"differ from the CPU oracle" in bad.stderr or b"different mask tables on different ranks" in bad.stderr), bad.stderr.decode()[-2000:]
