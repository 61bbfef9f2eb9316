"""The C-ABI shared library: loads here (no GPU), exports every symbol include/hinge_hip.h declares,
and refuses to work without a device instead of falling back to anything."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "hinge_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(hinge_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from hinge_amd import capi
    lib = capi.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), "libhinge_hip.so lacks %s" % name
    bound = {s[0] for s in capi.SYMBOLS + capi.EXTRA_SYMBOLS}
    assert set(declared) <= bound, "capi.py does not bind: %s" % sorted(set(declared) - bound)
    # and the other way round: nothing is exported that the header does not declare
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], stdout=subprocess.PIPE, check=True).stdout.decode()
    exported = {l.split()[-1] for l in out.splitlines() if l.split() and l.split()[-1].startswith("hinge_") and " T " in l}
    assert exported <= set(declared), "exported but not declared in include/hinge_hip.h: %s" % sorted(exported - set(declared))


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from hinge_amd import capi
    with pytest.raises(capi.HingeError) as e:
        capi.Context(0)
    assert e.value.code == -2     # HINGE_E_DEVICE


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under hinge_amd/ or include/ may reference it."""
    bad = []
    for base in ("hinge_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".inc", ".hpp")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"^\s*(import|from)\s+oracle\b|libhinge_oracle|oracle/|#include\s+\"oracle", txt, flags=re.M):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_pile_bins_restates_profile_coverage_bin_count():
    """capi.pile_bins (what an ingest hands to hinge_set_pile_bins): K of profileCoverage per read (LAInterface.cpp:4298-4320: the
    largest event / reso + 2; 0 for an empty pile-up), -1 where the fast kernel must not take the read."""
    import numpy as np
    from hinge_amd import capi
    rlen = np.array([5000, 4000, 100, 7000, 3000], np.int32)
    row_ptr = np.array([0, 2, 2, 3, 5, 6], np.int64)
    a_span = np.array([[0, 4100], [300, 1999], [10, 90], [0, 7000], [6999, 7001], [-1, 50]], np.int32)
    got = capi.pile_bins(row_ptr, a_span, rlen)
    assert got.tolist() == [4100 // 40 + 2, 0, 90 // 40 + 2, -1, -1]          # read 3: 7001 > rlen; read 4: a negative coordinate
    # a slice of a larger table (absolute offsets), as hinge_amd/stages.py passes it
    assert capi.pile_bins(row_ptr[2:5], a_span, rlen[2:4]).tolist() == [90 // 40 + 2, -1]
    assert capi.pile_bins(np.array([0, 0], np.int64), np.zeros((0, 2), np.int32), rlen[:1]).tolist() == [0]


@pytest.mark.gpu
def test_rccl_between_contexts_one_rank():
    """hinge_comm_create / hinge_comm_exchange_mask_rows (the executables' RCCL exchange of mask rows) as far as a 1-GPU
    box can run them: librccl is dlopen'ed, ncclCommInitAll makes a one-rank communicator, the grouped ncclAllGather runs on the
    context's stream, the own rows survive; two contexts on ONE device are refused (RCCL takes one rank per device), which is
    where the executables fall back to host exchanges."""
    import ctypes as C
    import numpy as np
    from hinge_amd import capi
    lib = capi.load_library()
    n = 1000
    rlen = np.full(n, 6000, np.int32)
    row_ptr = np.arange(n + 1, dtype=np.int64)
    a_span = np.tile(np.array([[0, 5000]], np.int32), (n, 1))
    b_span = a_span.copy()
    b_flag = ((np.arange(n) + 1) % n).astype(np.uint32)
    ctx = capi.Context(0)
    ctx.set_reads(rlen, None)
    ctx.set_pileups(0, n - 1, row_ptr, a_span, b_span, b_flag)
    rows = np.arange(2 * n, dtype=np.int32).reshape(n, 2)
    ctx._ck(lib.hinge_set_mask_rows(ctx.h, 0, n - 1, rows.ctypes.data_as(C.c_void_p)))
    arr = (C.c_void_p * 1)(ctx.h)
    rc = lib.hinge_comm_create(arr, 1)
    assert rc == 0, lib.hinge_last_error(ctx.h)
    lo, hi = np.array([100], np.int32), np.array([899], np.int32)
    for phase in (0, 1):
        rc = lib.hinge_comm_exchange_mask_rows(arr, 1, lo.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p), phase)
        assert rc == 0, lib.hinge_last_error(ctx.h)
    mask, _, _ = ctx.get_masks()
    assert np.array_equal(mask, rows)
    # phase 1 places rows from the stage of the phase-0 call of the same wave: without one, or over other rows, it is refused
    assert lib.hinge_comm_exchange_mask_rows(arr, 1, lo.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p), 1) == capi.HINGE_E_ARG
    assert lib.hinge_comm_exchange_mask_rows(arr, 1, lo.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p), 0) == 0
    lo2 = np.array([50], np.int32)
    assert lib.hinge_comm_exchange_mask_rows(arr, 1, lo2.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p), 1) == capi.HINGE_E_ARG
    assert b"phase 1 without" in lib.hinge_last_error(ctx.h)
    other = capi.Context(0)
    two = (C.c_void_p * 2)(ctx.h, other.h)
    assert lib.hinge_comm_create(two, 2) == capi.HINGE_E_DEVICE
    assert b"share a device" in lib.hinge_last_error(ctx.h)
    other.close()
    ctx.close()


def _allgather_rows(lib, ctxs, rows):
    """hinge_comm_allgather_rows over the given contexts: rows[k] = rank k's (n_k, w) int32 array.  Returns the gathered array."""
    import ctypes as C
    import numpy as np
    n = len(ctxs)
    arr = (C.c_void_p * n)(*[c.h for c in ctxs])
    w = rows[0].shape[1]
    ptrs = (C.c_void_p * n)(*[r.ctypes.data_as(C.c_void_p).value if len(r) else None for r in rows])
    counts = np.array([len(r) for r in rows], np.int64)
    out = np.full((max(int(counts.sum()), 1), w), -1, np.int32)
    got = np.zeros(n, np.int64)
    rc = lib.hinge_comm_allgather_rows(arr, n, ptrs, counts.ctypes.data_as(C.c_void_p), 4 * w, out.ctypes.data_as(C.c_void_p), int(counts.sum()),
                                       got.ctypes.data_as(C.c_void_p))
    assert rc == 0, lib.hinge_last_error(ctxs[0].h)
    assert np.array_equal(got, counts)
    return out[:int(counts.sum())]


@pytest.mark.gpu
def test_rccl_allgather_rows_one_rank():
    """hinge_comm_allgather_rows (round 6: containment candidates / classified matches of the executables' ranks over RCCL) as far
    as a 1-GPU box can run it: a one-rank communicator, both grouped ncclAllGathers, the rows back unchanged; empty and one-row
    inputs; without a communicator it is refused."""
    import ctypes as C
    import numpy as np
    from hinge_amd import capi
    lib = capi.load_library()
    ctx = capi.Context(0)
    arr = (C.c_void_p * 1)(ctx.h)
    rng = np.random.default_rng(1)
    rows = rng.integers(-2 ** 31, 2 ** 31 - 1, size=(12345, 10), dtype=np.int64).astype(np.int32)
    ptrs = (C.c_void_p * 1)(rows.ctypes.data_as(C.c_void_p).value)
    cnt = np.array([len(rows)], np.int64)
    out = np.zeros_like(rows)
    assert lib.hinge_comm_allgather_rows(arr, 1, ptrs, cnt.ctypes.data_as(C.c_void_p), 40, out.ctypes.data_as(C.c_void_p), len(rows), None) == capi.HINGE_E_ARG
    assert b"hinge_comm_create" in lib.hinge_last_error(ctx.h)
    assert lib.hinge_comm_create(arr, 1) == 0, lib.hinge_last_error(ctx.h)
    for r in (rows, rows[:1], rows[:0], rows[:777, :2].copy()):
        assert np.array_equal(_allgather_rows(lib, [ctx], [r]), r)
    assert lib.hinge_comm_allgather_rows(arr, 1, ptrs, cnt.ctypes.data_as(C.c_void_p), 40, out.ctypes.data_as(C.c_void_p), len(rows) - 1, None) == capi.HINGE_E_CAPACITY
    ctx.close()
