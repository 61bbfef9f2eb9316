"""The bench workload through the three executables against the oracle, every output file compared byte for byte
(tools/e2e_bench.py).  Default: the cfg2 generator at G = 1 Mb (5.4 M overlaps, a 0.7 GB .las); HINGE_FULL_SIZE=1 runs
the full E. coli-sized restatement (24.7 M overlaps, 3.35 GB .las, about a minute)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_executables_byte_identical_on_the_bench_workload():
    genome = 4_600_000 if os.environ.get("HINGE_FULL_SIZE") == "1" else 1_000_000
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "e2e_bench.py"), "--genome", str(genome)], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=1200)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert out["byte_identical"] is True
    assert out["overlaps"] > (20_000_000 if genome > 4_000_000 else 4_000_000)


@pytest.mark.parametrize("name,genome,full", [("cfg4_yeast", 1_200_000, 3_000_000), ("cfg3_nctc", 600_000, 2_000_000)])
def test_other_baseline_configs_through_the_executables(oracle_lib, name, genome, full):
    """BASELINE.json's yeast-like configuration (8 DB blocks, --mlas) and the repeat-rich NCTC-like one (chimeric reads)
    through all three executables, 20 output files against the oracle.  HINGE_FULL_SIZE=1: 3 Mb / 2 Mb genomes
    (6 M / 14 M overlaps; clean when run for round 1)."""
    import dataclasses
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_pipeline
    from hinge_amd import synth
    spec = dataclasses.replace(synth.CONFIGS[name], genome_len=full if os.environ.get("HINGE_FULL_SIZE") == "1" else genome)
    res = fuzz_pipeline.run_case(0, spec, "", "", oracle_lib, "")
    assert res.startswith("ok"), res
