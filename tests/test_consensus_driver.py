"""hinge_amd/consensus.py - the Python twin of the `consensus` executable and its contig-sharded form - on CPU: the host logic
(the per-contig std::sort order through hinge_sort_order_desc, remove_multialign's count, the deal of contigs to ranks, the
gather, the text) runs for real; the per-contig compute is replaced by a table taken from the reference program's own output
(or the oracle's) for the same files.  World 1 and world 2 (gloo)."""
import os
import re

import pytest

import conftest
import torch.distributed as dist
import torch.multiprocessing as mp

import consensus_common as cc


def _parse_reference(fasta: bytes, log: bytes):
    """contig -> ContigResult from the reference program's files."""
    from hinge_amd.consensus import ContigResult
    seqs = {}
    lines = fasta.decode().split("\n")
    for k in range(0, len(lines) - 1, 2):
        seqs[int(lines[k][len(">Consensus"):])] = lines[k + 1].encode()
    out, cur, offs = {}, None, []
    text = log.decode().split("\n")
    clen = {}
    for l in text:
        m = re.match(r"^(\d+)\t(\d+)$", l)
        if m:
            clen[int(m.group(1))] = int(m.group(2))
    i = text.index("Building consensus sequences...") + 1
    while i < len(text):
        m = re.match(r"^Contig (\d+): (\d+) reads$", text[i])
        if not m:
            i += 1
            continue
        c, n = int(m.group(1)), int(m.group(2))
        i += 1
        offs = [int(text[i + k]) for k in range(n)]
        i += n
        if n == 0:
            out[c] = ContigResult(seqs[c], (0, clen[c], 0, 0, 0, clen[c], 0), [])
            continue
        avg = float(text[i].split()[-1])
        g, ins, dl, low = [int(text[i + k].split(":")[1].split("/")[0]) for k in (1, 2, 3, 4)]
        cl = int(text[i + 5].split(":")[1])
        out[c] = ContigResult(seqs[c], (int(round(avg * clen[c])), clen[c], g, ins, dl, low, cl), offs)
        i += 6
    return out


class TableBackend:
    def __init__(self, table):
        self.table, self.asked = table, None

    def run(self, las, sel, mine):
        self.asked = list(mine)
        return {c: self.table[c] for c in mine}


def _expected(oracle_lib, wd):
    return cc.run_reference(wd) or cc.run_oracle(oracle_lib, wd)


@pytest.mark.parametrize("name", ["cns_small", "cns_midsize", "cns_twobyte"])
def test_driver_text_and_selection_match_the_reference(oracle_lib, tmp_path, name):
    from hinge_amd import consensus
    wd = str(tmp_path)
    cc.make(name, wd)
    fasta, log = _expected(oracle_lib, wd)
    be = TableBackend(_parse_reference(fasta, log))
    rc, text = consensus.run_consensus(os.path.join(wd, "draft"), os.path.join(wd, "reads"), os.path.join(wd, "draft.reads.las"), os.path.join(wd, "py.fasta"),
                                       os.path.join(wd, "nominal.ini"), backend=be)
    assert rc == 0
    assert open(os.path.join(wd, "py.fasta"), "rb").read() == fasta
    # everything of the stdout text but the float line is host logic: contig sizes, the listed counts, `Contig i: n reads` (the sort
    # order + remove_multialign's count), the order of the chop offsets
    strip = lambda t: b"\n".join(l for l in t.split(b"\n") if not l.startswith(b"Average coverage"))     # noqa: E731
    assert strip(text) == strip(log)
    rc, text = consensus.run_consensus(os.path.join(wd, "draft"), os.path.join(wd, "reads"), os.path.join(wd, "draft.reads.las"), os.path.join(wd, "q.fasta"),
                                       os.path.join(wd, "missing.ini"), backend=be)
    assert rc == 1 and text == ("Can't load %s\n" % os.path.join(wd, "missing.ini")).encode()


def _worker(rank, world, port, wd, table, fasta, ret):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hinge_amd import consensus
        be = TableBackend(table)
        rc, text = consensus.run_consensus(os.path.join(wd, "draft"), os.path.join(wd, "reads"), os.path.join(wd, "draft.reads.las"),
                                           os.path.join(wd, "sharded.fasta"), os.path.join(wd, "nominal.ini"), backend=be)
        dist.barrier()
        ok = rc == 0 and open(os.path.join(wd, "sharded.fasta"), "rb").read() == fasta
        ret.put((rank, ok, be.asked, text))
    finally:
        dist.destroy_process_group()


def test_contigs_sharded_over_two_ranks(oracle_lib, tmp_path):
    """World 2 (gloo): each rank computes only the contigs dealt to it, rank 0 writes the reference's FASTA, both return its text."""
    wd = str(tmp_path)
    cc.make("cns_small", wd)
    fasta, log = _expected(oracle_lib, wd)
    table = _parse_reference(fasta, log)
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = conftest.free_port()
    mp.spawn(_worker, args=(2, port, wd, table, fasta, ret), nprocs=2, join=True)
    got = sorted(ret.get() for _ in range(2))
    assert all(g[1] for g in got)
    asked = [set(g[2]) for g in got]
    assert asked[0] | asked[1] == set(table) and not (asked[0] & asked[1]) and asked[0] and asked[1]
    assert got[0][3] == got[1][3]
