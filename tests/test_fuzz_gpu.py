"""Randomised differential test (tools/fuzz_pipeline.py): random generator settings and nominal.ini values, the three
executables against the oracle, every output file byte for byte.  Nine cases here; `python tools/fuzz_pipeline.py
--cases 100 --seed N [--paths]` for more (1820 cases were run for round 1: no difference)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_cases_match_the_oracle(oracle_lib):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_pipeline
    rng = np.random.default_rng(9)   # (a mix of PAF input, the general K2 kernel, the exact hinge paths, several reads per wavefront: ~10 s)
    results = []
    for k in range(9):
        spec, filt, lay = fuzz_pipeline.random_case(rng)
        env, paf = fuzz_pipeline.random_paths(rng, spec)     # alternative kernel paths, thread counts, FASTA + PAF input
        results.append(fuzz_pipeline.run_case(k, spec, filt, lay, oracle_lib, "", env, paf))
    assert not [r for r in results if r.startswith("FAIL")], results
    assert sum(r.startswith("ok") for r in results) >= 5, results
