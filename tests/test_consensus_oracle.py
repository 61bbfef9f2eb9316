"""`hinge consensus`, CPU side: oracle/consensus_oracle.cpp (the restatement) against the reference's OWN program -
oracle/_ref/consensus, built unmodified from src/consensus/consensus.cpp by oracle/Makefile - live where it exists, and against
the golden digests it produced (tests/golden/consensus_golden.json, make_consensus_golden.py) everywhere.  FASTA and stdout,
byte for byte.  This stage is PINNED (the three graph stages are not: they need spdlog / Boost.Graph)."""
import os

import pytest

import consensus_common as cc

NAMES = sorted(cc.GOLDEN)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_the_reference_program(oracle_lib, tmp_path, name):
    wd = str(tmp_path)
    cc.make(name, wd)
    fasta, log = cc.run_oracle(oracle_lib, wd)
    g = cc.GOLDEN[name]
    assert cc.sha(fasta) == g["fasta_sha256"], "oracle FASTA differs from the reference program's golden output"
    assert cc.sha(log) == g["stdout_sha256"], "oracle stdout differs from the reference program's golden output"
    ref = cc.run_reference(wd)
    if ref is not None:       # live: the reference binary itself, same files
        assert ref[0] == fasta and ref[1] == log
    if name == "cns_tiny":
        assert fasta == open(os.path.join(cc.ROOT, "tests", "golden", "consensus_cns_tiny.fasta"), "rb").read()


def test_reference_quirks_are_exercised(oracle_lib, tmp_path):
    """The data sets must reach the branches the restatement has to get right: lower-case (coverage < 3) stretches, inserted and
    deleted bases, a contig without alignments, remove_multialign's count dropping the sorted list's tail."""
    wd = str(tmp_path)
    d = cc.make("cns_small", wd)
    fasta, log = cc.run_oracle(oracle_lib, wd)
    text = log.decode()
    lines = fasta.decode().split("\n")
    assert any(c.islower() for c in lines[1]) and any(c.isupper() for c in lines[1])
    assert "Contig 3: 0 reads" in text and lines[7].islower()          # the empty contig is printed as it is
    ins = [int(l.split()[1].split("/")[0]) for l in text.split("\n") if l.startswith("Insertions:")]
    dels = [int(l.split()[1].split("/")[0]) for l in text.split("\n") if l.startswith("Deletions:")]
    assert min(ins) > 0 and min(dels) > 0
    # fewer reads used than alignments listed (a duplicate B read and the short alignments below min_length)
    listed = [int(l.split()[1]) for l in text.split("\n")[7:11]]
    used = [int(l.split()[2]) for l in text.split("\n") if l.startswith("Contig ")]
    assert any(u < n for u, n in zip(used, listed))
