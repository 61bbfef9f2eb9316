"""hinge_amd/csrc/stdsort_emul.h (the host/device replay of libstdc++ std::sort the HIP kernels use)
compiled for the host and compared with the real std::sort (through the oracle library)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ip = ctypes.POINTER(ctypes.c_int)
P = lambda a: a.ctypes.data_as(ip)  # noqa: E731

HARNESS = r'''
#include <algorithm>
#include <vector>
#include "stdsort_emul.h"
extern "C" void emul_sort_perm(int n, const int* key, int desc, int* perm) {
    for (int i = 0; i < n; i++) perm[i] = i;
    hinge_sort::std_sort(perm, n, key, desc);
}
// McIlroy's "antiquicksort" adversary run against std::sort: produces keys that drive the
// median-of-3 introsort into its depth limit (the heapsort fallback).
static std::vector<int> val; static int nsolid, candidate, gas;
static bool adv_cmp(int x, int y) {
    if (val[x] == gas && val[y] == gas) { if (x == candidate) val[x] = nsolid++; else val[y] = nsolid++; }
    if (val[x] == gas) candidate = x; else if (val[y] == gas) candidate = y;
    return val[x] < val[y];
}
extern "C" void killer_keys(int n, int* out) {
    val.assign(n, 0); gas = n - 1; nsolid = 0; candidate = 0;
    std::vector<int> idx(n);
    for (int i = 0; i < n; i++) { idx[i] = i; val[i] = gas; }
    std::sort(idx.begin(), idx.end(), adv_cmp);
    for (int i = 0; i < n; i++) out[i] = val[i];
}
'''


@pytest.fixture(scope="module")
def emul(tmp_path_factory):
    d = tmp_path_factory.mktemp("emul")
    src = d / "h.cpp"
    src.write_text(HARNESS)
    so = d / "h.so"
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-I", os.path.join(ROOT, "hinge_amd", "csrc"), "-o", str(so), str(src)])
    lib = ctypes.CDLL(str(so))
    lib.emul_sort_perm.argtypes = [ctypes.c_int, ip, ctypes.c_int, ip]
    lib.killer_keys.argtypes = [ctypes.c_int, ip]
    return lib


def _check(emul, oracle_lib, key):
    key = np.ascontiguousarray(key, np.int32)
    n = len(key)
    for mode_o, desc in ((0, 1), (1, 0)):
        a = np.zeros(max(n, 1), np.int32)
        b = np.zeros(max(n, 1), np.int32)
        oracle_lib.oracle_sort_perm(n, P(key), mode_o, P(a))
        emul.emul_sort_perm(n, P(key), desc, P(b))
        assert np.array_equal(a, b), (n, desc)


def test_replay_matches_std_sort(emul, oracle_lib):
    rng = np.random.default_rng(0)
    for trial in range(1500):
        n = int(rng.choice([0, 1, 2, 3, 15, 16, 17, 18, 31, 33, 64, 100, 257, 1000, 5000, int(rng.integers(1, 3000))]))
        key = [rng.integers(0, 4, size=n), rng.integers(0, max(1, n // 8) + 1, size=n), np.sort(rng.integers(0, 50, size=n)),
               np.sort(rng.integers(0, 50, size=n))[::-1], rng.integers(0, 1 << 20, size=n)][trial % 5]
        _check(emul, oracle_lib, key)


def test_replay_matches_std_sort_in_heapsort_fallback(emul, oracle_lib):
    """Adversarial keys push introsort past its depth limit: the heap part must match too."""
    for n in (200, 1000, 4096, 20000):
        k = np.zeros(n, np.int32)
        emul.killer_keys(n, P(k))
        _check(emul, oracle_lib, k)            # ascending comparator = the adversary's own
        _check(emul, oracle_lib, -k)
