"""How long does the LDS replay of std::sort (block_std_sort_desc) take for one list?  One workgroup per call through
hinge_debug_pileup_order; run under `rocprofv3 --kernel-trace` and read the dispatches' durations (tools/probes/sort_probe.sh)."""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from hinge_amd import capi
ctx = capi.Context(0)
rng = np.random.default_rng(1)
cases = []
for n in (64, 256, 768, 1700):
    cases.append(("distinct", n, rng.permutation(n * 4)[:n]))
    cases.append(("ties/8", n, rng.integers(0, max(2, n // 8), size=n)))
    cases.append(("ties/64", n, rng.integers(0, max(2, n // 64), size=n)))
    cases.append(("3 values", n, rng.integers(0, 3, size=n)))
    cases.append(("sorted desc", n, np.sort(rng.integers(0, n, size=n))[::-1]))
for name, n, key in cases:
    for _ in range(5):
        ctx.debug_pileup_order(np.ascontiguousarray(key, np.int32))
print("\n".join("%s %d" % (name, n) for name, n, _ in cases))
