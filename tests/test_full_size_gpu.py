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


@pytest.mark.skipif(os.environ.get("HINGE_FULL_SIZE") != "1", reason="several minutes of single-thread oracle: HINGE_FULL_SIZE=1 (run once per round, log under profiles/)")
@pytest.mark.parametrize("name", ["cfg3_nctc", "cfg4_yeast"])
def test_configs_3_and_4_at_their_full_size(oracle_lib, name):
    """BASELINE configs 3 (NCTC-like: 5 Mb at 100x, 40 repeat families, chimeric reads) and 4 (yeast-like: 12 Mb at 80x, 8 DB
    blocks, --mlas over one rank per visible GPU) at their OWN size through all three executables, 20 output files against the
    oracle byte for byte."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_pipeline
    from hinge_amd import synth
    res = fuzz_pipeline.run_case(0, synth.CONFIGS[name], "", "", oracle_lib, "")
    assert res.startswith("ok"), res
    print(name, res)
