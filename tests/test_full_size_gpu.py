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
