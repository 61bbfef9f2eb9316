"""The bench workload through the three executables against the oracle, every output file compared byte for byte
(tools/e2e_bench.py) at the FULL size of BASELINE config 2 (the E. coli 160x restatement: 86 588 reads, 26.2 M overlap
records, a 3.5 GB .las; about two minutes, most of it the single-thread oracle).  HINGE_SMALL=1 shrinks the genomes
(G = 1 Mb: 5.4 M overlaps) for a quick run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_executables_byte_identical_on_the_bench_workload():
    genome = 1_000_000 if os.environ.get("HINGE_SMALL") == "1" else 4_600_000
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "e2e_bench.py"), "--genome", str(genome), "--exact-config"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=1200)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert out["byte_identical"] is True
    assert out["overlaps"] > (20_000_000 if genome > 4_000_000 else 4_000_000)


@pytest.mark.parametrize("name,genome,full", [("cfg1_ecoli_demo", 1_000_000, 4_600_000), ("cfg4_yeast", 1_200_000, 3_000_000), ("cfg3_nctc", 600_000, 2_000_000)])
def test_other_baseline_configs_through_the_executables(oracle_lib, name, genome, full):
    """BASELINE.json's other configurations through all three executables, 20 output files against the oracle: config 1
    (ecoli_demo plumbing: 4.6 Mb at 30x, FULL size by default - 20 k reads, 1 M overlaps), the yeast-like one (8 DB blocks,
    --mlas) and the repeat-rich NCTC-like one (chimeric reads) at 3 Mb / 2 Mb genomes (6 M / 14 M overlaps).
    HINGE_SMALL=1: the smaller genomes."""
    import dataclasses
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_pipeline
    from hinge_amd import synth
    spec = dataclasses.replace(synth.CONFIGS[name], genome_len=genome if os.environ.get("HINGE_SMALL") == "1" else full)
    res = fuzz_pipeline.run_case(0, spec, "", "", oracle_lib, "")
    assert res.startswith("ok"), res


@pytest.mark.parametrize("name", ["cfg3_nctc", "cfg4_yeast", "cfg2_ecoli160"])
def test_configs_at_their_full_size_against_committed_digests(name, tmp_path):
    """BASELINE configs 3 (NCTC-like: 5 Mb at 100x, 40 repeat families, chimeric reads: 62 500 reads, 18.7 M overlaps, 14 657
    hinges), 4 (yeast-like: 12 Mb at 80x, 8 DB blocks, --mlas over one rank per visible GPU: 120 000 reads, 18.8 M overlaps) and
    2 (E. coli 160x: 86 588 reads, 26.2 M overlaps) at their OWN size through all three executables; the 20 output files against
    the sha256 digests of the CPU oracle's files, made once in the build container (tests/golden/make_full_size_digests.py ->
    tests/golden/full_size_digests.json: 96-190 s of single-thread oracle per configuration that the GPU box does not spend).
    The generated input is digested too, so a drifting generator is reported as such."""
    import hashlib
    import conftest
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import fuzz_pipeline
    import make_full_size_digests as mk
    from hinge_amd import synth
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "full_size_digests.json")))[name]
    spec = synth.CONFIGS[name]
    d = synth.generate(spec)
    assert (int(d.n_reads), int(d.novl)) == (want["reads"], want["records"])
    assert mk.input_digest(d) == want["input_sha256"], "the generator no longer produces the data set the digests were made on"
    wd = str(tmp_path)
    synth.write_dataset(d, wd, "G", write_bases=False)
    del d
    conftest.write_ini(os.path.join(wd, "v.ini"))
    mlas = spec.n_blocks > 1
    hinge = os.path.join(ROOT, "hinge_amd", "bin", "hinge")
    for sub, extra in (("filter", []), ("maximal", []), ("layout", ["-o", "G"])):
        argv = [hinge, sub, "--db", "G", "--las", "G" if mlas else "G.las"] + (["--mlas"] if mlas else []) + ["-x", "G", "--config", "v.ini"] + extra
        r = subprocess.run(argv, cwd=wd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        assert r.returncode == 0, (sub, r.stderr.decode()[-1500:])
    bad = [f for f in fuzz_pipeline.FILES if mk.sha(os.path.join(wd, f)) != want["sha256"][f]]
    assert not bad, "differs from the oracle's digests in %s" % bad
    hinges = sum((len(l.split()) - 1) // 2 for l in open(os.path.join(wd, "G.hinges.txt")))
    assert hinges == want["hinges"]
