"""The C++ hosts' RCCL exchange (hinge_comm_create / hinge_comm_exchange_mask_rows, hinge_amd/csrc/comm_capi.inc) between DISTINCT
devices - what `hinge filter --mlas` runs with one rank per visible GPU (filter_main.cpp:103,294; the reference's counterpart is the
sequential loop over the parts, filter.cpp:474,534,778-787).  Needs two GPUs: on the single-GPU boxes of this build every test
here is skipped with that reason (the one-rank form runs in tests/test_capi_library.py); tools/scale_smoke.sh runs this file first
on the first multi-GPU node it meets.  (Named to sort LAST: under `pytest -x` a first-contact failure between devices must not hide
the rest of the suite.)"""
import ctypes as C
import itertools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _n_devices():
    from hinge_amd import capi
    return capi.load_library().hinge_device_count()


def _contexts(devs, n_reads):
    from hinge_amd import capi
    lib = capi.load_library()
    rlen = np.full(n_reads, 6000, np.int32)
    row_ptr = np.arange(n_reads + 1, dtype=np.int64)
    a_span = np.tile(np.array([[0, 5000]], np.int32), (n_reads, 1))
    b_flag = ((np.arange(n_reads) + 1) % n_reads).astype(np.uint32)
    ctxs = []
    for d in devs:
        c = capi.Context(d)
        c.set_reads(rlen, None)
        c.set_pileups(0, n_reads - 1, row_ptr, a_span, a_span.copy(), b_flag)
        ctxs.append(c)
    return lib, ctxs


def _exchange(lib, ctxs, lo, hi, rows_of_rank, n_reads):
    """Every rank's table starts with its own rows only; phase 0 + phase 1 must leave every table with every rank's rows."""
    arr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
    for k, c in enumerate(ctxs):
        table = np.full((n_reads, 2), -7, np.int32)
        table[lo[k]:hi[k] + 1] = rows_of_rank[k]
        c._ck(lib.hinge_set_mask_rows(c.h, 0, n_reads - 1, table.ctypes.data_as(C.c_void_p)))
    rc = lib.hinge_comm_create(arr, len(ctxs))
    assert rc == 0, lib.hinge_last_error(ctxs[0].h)
    lo_a, hi_a = np.array(lo, np.int32), np.array(hi, np.int32)
    after0 = None
    for phase in (0, 1):
        rc = lib.hinge_comm_exchange_mask_rows(arr, len(ctxs), lo_a.ctypes.data_as(C.c_void_p), hi_a.ctypes.data_as(C.c_void_p), phase)
        assert rc == 0, lib.hinge_last_error(ctxs[0].h)
        if phase == 0:
            after0 = [c.get_masks()[0].copy() for c in ctxs]
    return after0, [c.get_masks()[0] for c in ctxs]


@pytest.mark.parametrize("ragged", [False, True])
def test_mask_rows_between_devices_every_order(ragged):
    nd = _n_devices()
    if nd < 2:
        pytest.skip("one visible GPU: RCCL takes one rank per device (the one-rank form is in tests/test_capi_library.py)")
    n_reads = 4001
    use = min(nd, 4)
    for devs in itertools.islice(itertools.permutations(range(nd), use), 6):     # rank k on device devs[k]: every order matters (stream / device binding)
        lib, ctxs = _contexts(devs, n_reads)
        cuts = np.linspace(0, n_reads, use + 1).astype(int)
        if ragged:
            cuts[1] = cuts[0] + 3                                                # one rank with 3 rows, one empty block below
        lo = [int(cuts[k]) for k in range(use)]
        hi = [int(cuts[k + 1]) - 1 for k in range(use)]
        if ragged and use > 2:
            hi[2] = lo[2] - 1                                                    # rank 2 contributes nothing
        rows = [np.stack([1000 * k + np.arange(max(hi[k] - lo[k] + 1, 0)), -np.arange(max(hi[k] - lo[k] + 1, 0))], 1).astype(np.int32) for k in range(use)]
        after0, after1 = _exchange(lib, ctxs, lo, hi, rows, n_reads)
        want = np.full((n_reads, 2), -7, np.int32)
        for k in range(use):
            want[lo[k]:hi[k] + 1] = rows[k]
        for k in range(use):
            # phase 0: rank k sees the ranks before it (what part k of the sequential --mlas loop sees while its hinges are called)
            w0 = np.full((n_reads, 2), -7, np.int32)
            for q in range(k + 1):
                w0[lo[q]:hi[q] + 1] = rows[q]
            assert np.array_equal(after0[k], w0), (devs, k, "phase 0")
            assert np.array_equal(after1[k], want), (devs, k, "phase 1")
        for c in ctxs:
            c.close()


def test_two_contexts_on_one_device_are_refused():
    """(runs on every box) RCCL takes one rank per device: the executables then exchange through the host."""
    lib, ctxs = _contexts([0, 0], 100)
    arr = (C.c_void_p * 2)(*[c.h for c in ctxs])
    assert lib.hinge_comm_create(arr, 2) != 0
    assert b"share a device" in lib.hinge_last_error(ctxs[0].h)
    for c in ctxs:
        c.close()


def test_allgather_rows_between_devices_every_order():
    """hinge_comm_allgather_rows (the containment candidates of `hinge maximal --mlas`, the classified matches of `hinge layout --mlas`,
    maximal.cpp:805-857 / hinging.cpp:917-936) between distinct devices: ragged and empty blocks, every rank order."""
    from test_capi_library import _allgather_rows
    nd = _n_devices()
    if nd < 2:
        pytest.skip("one visible GPU: RCCL takes one rank per device (the one-rank form is in tests/test_capi_library.py)")
    use = min(nd, 4)
    rng = np.random.default_rng(3)
    for devs in itertools.islice(itertools.permutations(range(nd), use), 6):
        lib, ctxs = _contexts(devs, 100)
        arr = (C.c_void_p * use)(*[c.h for c in ctxs])
        assert lib.hinge_comm_create(arr, use) == 0, lib.hinge_last_error(ctxs[0].h)
        for sizes in ([5000, 1, 0, 70000][:use], [0] * use, [3] * use, [1, 200000, 2, 3][:use]):
            rows = [rng.integers(-1000, 1000, size=(s, 10)).astype(np.int32) + 100000 * k for k, s in enumerate(sizes)]
            want = np.concatenate(rows) if sum(sizes) else np.zeros((0, 10), np.int32)
            assert np.array_equal(_allgather_rows(lib, ctxs, rows), want), (devs, sizes)
        for c in ctxs:
            c.close()
