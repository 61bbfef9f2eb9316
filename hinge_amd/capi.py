"""ctypes binding of libhinge_hip.so (include/hinge_hip.h).

There is no fallback: if the HIP library is missing, or no GPU is visible when a context is
created, this raises.  numpy arrays are passed as plain pointers; torch tensors via data_ptr().
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HINGE_LIB") or os.path.join(_HERE, "lib", "libhinge_hip.so")   # HINGE_LIB: another build of the library (tools/ablate_k2.sh)

HINGE_OK = 0
HINGE_E_ARG, HINGE_E_DEVICE, HINGE_E_CAPACITY, HINGE_E_UNDEFINED, HINGE_E_RANGE = -1, -2, -3, -4, -5
ERR_NAMES = {-1: "HINGE_E_ARG", -2: "HINGE_E_DEVICE", -3: "HINGE_E_CAPACITY", -4: "HINGE_E_UNDEFINED", -5: "HINGE_E_RANGE"}


class HingeError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("%s (%d): %s" % (ERR_NAMES.get(code, "HINGE_E_?"), code, msg))
        self.code = code


class FilterParams(C.Structure):
    """hinge_filter_params: the [filter] keys as filter.cpp:377-406 reads them."""

    _fields_ = [(n, C.c_int32) for n in (
        "reso", "cut_off", "min_cov", "est_cov", "theta", "coverage_fraction", "min_repeat_annotation",
        "max_repeat_annotation", "repeat_annotation_gap", "no_hinge_region", "hinge_min_support", "hinge_bin_pileup",
        "hinge_unbridged", "hinge_tolerance", "use_qv_mask", "use_coverage_mask", "delete_telomere")]


class CovEstimate(C.Structure):
    _fields_ = [("cov_est", C.c_int32), ("n_long", C.c_int32), ("total_cov", C.c_int64), ("num_slot", C.c_int64)]


# every symbol include/hinge_hip.h declares: (name, restype, argtypes)
_VP = C.c_void_p
SYMBOLS = [
    ("hinge_device_count", C.c_int, []),
    ("hinge_ctx_device_memory", C.c_int, [_VP, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("hinge_ctx_create", C.c_int, [C.c_int, C.POINTER(_VP)]),
    ("hinge_ctx_destroy", None, [_VP]),
    ("hinge_last_error", C.c_char_p, [_VP]),
    ("hinge_set_stream", C.c_int, [_VP, _VP]),
    ("hinge_synchronize", C.c_int, [_VP]),
    ("hinge_set_reads", C.c_int, [_VP, C.c_int32, _VP, _VP]),
    ("hinge_set_pileups", C.c_int, [_VP, C.c_int32, C.c_int32, C.c_int64, _VP, _VP, _VP, _VP, C.c_int]),
    ("hinge_set_pileups_packed", C.c_int, [_VP, C.c_int32, C.c_int32, C.c_int64, _VP, _VP, _VP, _VP, _VP, C.c_uint32, C.c_int, C.c_int]),
    ("hinge_span16_pad", C.c_int, []),
    ("hinge_get_pileup_facts", C.c_int, [_VP, C.POINTER(C.c_uint32), C.POINTER(C.c_int)]),
    ("hinge_set_pile_bins", C.c_int, [_VP, C.c_int32, _VP, C.c_int]),
    ("hinge_attach_mask_table", C.c_int, [_VP, _VP]),
    ("hinge_attach_mean_cov", C.c_int, [_VP, _VP]),
    ("hinge_set_mask_rows", C.c_int, [_VP, C.c_int32, C.c_int32, _VP]),
    ("hinge_clear_masks", C.c_int, [_VP]),
    ("hinge_filter_stats", C.c_int, [_VP, C.POINTER(FilterParams)]),
    ("hinge_filter_median", C.c_int, [_VP, C.POINTER(FilterParams), C.c_int32, C.c_int32, C.POINTER(CovEstimate)]),
    ("hinge_filter_stats_median", C.c_int, [_VP, C.POINTER(FilterParams), _VP, C.POINTER(CovEstimate)]),
    ("hinge_filter_median_hist", C.c_int, [_VP, C.POINTER(FilterParams), C.c_int32, C.c_int32, _VP]),
    ("hinge_filter_median_from_hist", C.c_int, [_VP, C.POINTER(FilterParams), _VP]),
    ("hinge_filter_median_from_hist_batch", C.c_int, [C.POINTER(_VP), C.c_int32, C.POINTER(FilterParams), _VP, C.c_int64]),
    ("hinge_filter_median_batch", C.c_int, [C.POINTER(_VP), C.c_int32, C.POINTER(FilterParams), _VP, C.c_int64]),
    ("hinge_filter_hinges_batch_async", C.c_int, [C.POINTER(_VP), C.c_int32, C.POINTER(FilterParams)]),
    ("hinge_filter_sweep_batch_async", C.c_int, [C.POINTER(_VP), C.c_int32, C.POINTER(FilterParams), _VP, C.c_int64]),
    ("hinge_filter_finish_batch_async", C.c_int, [C.POINTER(_VP), C.c_int32, C.POINTER(FilterParams)]),
    ("hinge_filter_sweep", C.c_int, [_VP, C.POINTER(FilterParams), C.POINTER(CovEstimate)]),
    ("hinge_filter_spec_stats", C.c_int, [_VP, _VP]),
    ("hinge_debug_spec", C.c_int, [_VP, C.c_int, C.c_int, C.c_int]),
    ("hinge_set_read_restriction", C.c_int, [_VP, _VP]),
    ("hinge_filter_set_min_cov", C.c_int, [_VP, C.c_int32]),
    ("hinge_filter_get_min_cov", C.c_int, [_VP, C.POINTER(C.c_int32)]),
    ("hinge_filter_mask_annotate", C.c_int, [_VP, C.POINTER(FilterParams)]),
    ("hinge_filter_hinges", C.c_int, [_VP, C.POINTER(FilterParams)]),
    ("hinge_filter_run", C.c_int, [_VP, C.POINTER(FilterParams)]),
    ("hinge_filter_get_masks", C.c_int, [_VP, _VP, _VP, _VP]),
    ("hinge_filter_get_annotations", C.c_int, [_VP, _VP, _VP, _VP, _VP]),
    ("hinge_filter_coverage_bins", C.c_int, [_VP, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _VP, _VP, C.c_int64]),
    ("hinge_filter_coverage_out", C.c_int, [_VP, C.c_int]),
    ("hinge_filter_get_coverage", C.c_int, [_VP, _VP, _VP, _VP, C.c_int64]),
    ("hinge_filter_counters", C.c_int, [_VP, _VP]),
    ("hinge_set_traces", C.c_int, [_VP, _VP, C.c_int64, _VP, _VP, C.c_int, C.c_int]),
    ("hinge_set_las_image", C.c_int, [_VP, _VP, C.c_int64, _VP, _VP, C.c_int, C.c_int]),
    ("hinge_set_eff_reads", C.c_int, [_VP, _VP]),
    ("hinge_set_trim", C.c_int, [_VP, C.c_int]),
    ("hinge_trim_classify", C.c_int, [_VP, C.c_int64, _VP, _VP, C.c_int32, C.c_int32, C.c_int32, _VP]),
    ("hinge_trim_classify_types", C.c_int, [_VP, C.c_int64, _VP, _VP, C.c_int32, C.c_int32, C.c_int32, _VP]),
    ("hinge_trim_classify_part", C.c_int, [_VP, C.c_int32, C.c_int32, C.c_int32, _VP]),
    ("hinge_trim_classify_part_full", C.c_int, [_VP, C.c_int32, C.c_int32, C.c_int32, _VP]),
    ("hinge_matching_position", C.c_int, [_VP, C.c_int64, _VP, _VP, _VP]),
    ("hinge_select_edges", C.c_int, [_VP, C.c_int32, _VP, C.c_int64, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int32, C.c_int32, _VP, _VP, _VP]),
    ("hinge_filter_mask_annotate_async", C.c_int, [_VP, C.POINTER(FilterParams)]),
    ("hinge_filter_hinges_async", C.c_int, [_VP, C.POINTER(FilterParams)]),
    ("hinge_filter_check", C.c_int, [_VP]),
    ("hinge_profile_enable", C.c_int, [_VP, C.c_int]),
    ("hinge_profile_select", C.c_int, [_VP, C.c_uint32]),
    ("hinge_profile_kernels", C.c_int, []),
    ("hinge_profile_kernel_name", C.c_char_p, [C.c_int]),
    ("hinge_resolve_containment", C.c_int, [C.c_int32, _VP, C.c_int64, _VP, _VP]),
    ("hinge_sort_order_desc", C.c_int, [C.c_int32, _VP, C.c_int32, _VP]),
    ("hinge_pick_pairs", C.c_int64, [C.c_int32, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _VP, _VP, C.c_int64]),
    ("hinge_comm_create", C.c_int, [C.POINTER(_VP), C.c_int32]),
    ("hinge_comm_exchange_mask_rows", C.c_int, [C.POINTER(_VP), C.c_int32, _VP, _VP, C.c_int32]),
    ("hinge_comm_allgather_rows", C.c_int, [C.POINTER(_VP), C.c_int32, C.POINTER(_VP), _VP, C.c_int32, _VP, C.c_int64, _VP]),
    ("hinge_consensus_set_db", C.c_int, [_VP, C.c_int32, C.c_int32, _VP, _VP, _VP, C.c_int64]),
    ("hinge_consensus_run", C.c_int, [_VP, C.c_int64, _VP, _VP, C.c_int64, C.c_int32]),
    ("hinge_consensus_get_contig", C.c_int, [_VP, C.c_int32, _VP, C.c_int64, C.POINTER(C.c_int64), _VP]),
    ("hinge_consensus_get_offsets", C.c_int, [_VP, _VP]),
    ("hinge_consensus_get_indels", C.c_int, [_VP, C.c_int64, _VP, C.c_int64, C.POINTER(C.c_int64)]),
    ("hinge_draft_mappings", C.c_int, [_VP, C.c_int64, _VP, _VP, C.c_int64, C.c_int32, _VP, _VP]),
    ("hinge_draft_ladders", C.c_int, [_VP, C.c_int64, _VP, _VP, _VP, C.c_int32, _VP, _VP, _VP]),
    ("hinge_profile_report", C.c_int, [_VP, _VP, _VP]),
    ("hinge_timer_start", C.c_int, [_VP]),
    ("hinge_timer_stop_ms", C.c_int, [_VP, C.POINTER(C.c_float)]),
]

_lib = None


def _share_hip_runtime_with_torch() -> None:
    """A PyTorch-ROCm wheel bundles its own libamdhip64 / libhsa-runtime64.  Two HIP runtimes in one process do not
    share the GPU: whichever comes second reports "No HIP GPUs are available".  If torch is installed but not imported
    yet, load ITS runtime first (same SONAME), so that libhinge_hip.so binds to it and a later `import torch` finds the
    GPU whatever the import order.  HINGE_SYSTEM_HIP=1 keeps the ROCm installation's runtime instead."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("HINGE_SYSTEM_HIP") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def load_library() -> C.CDLL:
    """Load libhinge_hip.so and bind every declared symbol (fails loudly if one is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libhinge_hip.so not built (%s): run `make` / __graft_entry__.build()" % LIB_PATH)
    _share_hip_runtime_with_torch()
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS + EXTRA_SYMBOLS:
        f = getattr(lib, name)
        f.restype = res
        f.argtypes = args
    _lib = lib
    return lib


# test hooks (declared at the end of include/hinge_hip.h)
EXTRA_SYMBOLS = [
    ("hinge_debug_force_exact", C.c_int, [_VP, C.c_int]),
    ("hinge_debug_force_general_mask", C.c_int, [_VP, C.c_int]),
    ("hinge_debug_fallback_reads", C.c_int, [_VP, _VP]),
    ("hinge_debug_heavy_items", C.c_int, [_VP, _VP]),
    ("hinge_debug_pileup_order", C.c_int, [_VP, C.c_int32, _VP, _VP]),
]


def _ptr(a) -> Optional[int]:
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return a.ctypes.data
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    return int(a)


class Context:
    """One hinge_ctx (one GPU)."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = _VP()
        rc = self.lib.hinge_ctx_create(device, C.byref(h))
        if rc != HINGE_OK:
            raise HingeError(rc, "hinge_ctx_create(device=%d) failed: no usable HIP device" % device)
        self.h = h
        self._keep = []
        self.r_begin = 0
        self.r_end = -1
        self.n_reads = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.hinge_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc: int):
        if rc != HINGE_OK:
            raise HingeError(rc, self.lib.hinge_last_error(self.h).decode())

    def set_stream(self, stream_ptr: int):
        self._ck(self.lib.hinge_set_stream(self.h, _VP(stream_ptr)))

    def synchronize(self):
        self._ck(self.lib.hinge_synchronize(self.h))

    def set_reads(self, rlen: np.ndarray, qv_mask: Optional[np.ndarray] = None):
        rlen = np.ascontiguousarray(rlen, dtype=np.int32)
        q = None if qv_mask is None else np.ascontiguousarray(qv_mask, dtype=np.int32)
        self.n_reads = int(rlen.shape[0])
        self._ck(self.lib.hinge_set_reads(self.h, self.n_reads, _ptr(rlen), _ptr(q)))

    def set_pileups(self, r_begin: int, r_end: int, row_ptr, a_span, b_span, b_flag, n_ovl: Optional[int] = None, on_device: bool = False):
        if not on_device:
            row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
            a_span = np.ascontiguousarray(a_span, dtype=np.int32)
            b_span = np.ascontiguousarray(b_span, dtype=np.int32)
            b_flag = np.ascontiguousarray(b_flag, dtype=np.uint32)
            n_ovl = int(b_flag.shape[0])
        self._keep = [row_ptr, a_span, b_span, b_flag]
        self.r_begin, self.r_end = int(r_begin), int(r_end)
        self._ck(self.lib.hinge_set_pileups(self.h, r_begin, r_end, int(n_ovl), _ptr(row_ptr), _ptr(a_span), _ptr(b_span), _ptr(b_flag),
                                            1 if on_device else 0))

    def set_pileups_packed(self, r_begin: int, r_end: int, row_ptr, a_span, b_span, b_flag, span16, max_pile: int, spans_in_range: bool,
                           n_ovl: Optional[int] = None, on_device: bool = False):
        """set_pileups with the ingest's facts: span16 (None, or abpos | aepos << 16 with span16_pad() spare elements behind it),
        the largest pile-up and whether every span lies inside its read (see pack_spans)."""
        if not on_device:
            row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
            a_span = np.ascontiguousarray(a_span, dtype=np.int32)
            b_span = np.ascontiguousarray(b_span, dtype=np.int32)
            b_flag = np.ascontiguousarray(b_flag, dtype=np.uint32)
            span16 = None if span16 is None else np.ascontiguousarray(span16, dtype=np.uint32)
            n_ovl = int(b_flag.shape[0])
        self._keep = [row_ptr, a_span, b_span, b_flag, span16]
        self.r_begin, self.r_end = int(r_begin), int(r_end)
        self._ck(self.lib.hinge_set_pileups_packed(self.h, r_begin, r_end, int(n_ovl), _ptr(row_ptr), _ptr(a_span), _ptr(b_span), _ptr(b_flag),
                                                   _ptr(span16), int(max_pile), 1 if spans_in_range else 0, 1 if on_device else 0))

    def set_pile_bins(self, nbins, reso: int = 40, on_device: bool = False):
        """The per-read bin counts of the pile-ups just set (pile_bins()): the one-sweep pass then launches no k_cov_stats at all."""
        self._keep.append(nbins)
        self._ck(self.lib.hinge_set_pile_bins(self.h, int(reso), _ptr(nbins), 1 if on_device else 0))

    def pileup_facts(self):
        """(largest pile-up, every span inside its read) of the current part."""
        mp, ok = C.c_uint32(), C.c_int()
        self._ck(self.lib.hinge_get_pileup_facts(self.h, C.byref(mp), C.byref(ok)))
        return int(mp.value), bool(ok.value)

    def coverage_out(self, on: bool):
        """K2 also stores the cutoff-0 coverage bins (the .coverage.txt payload); read them with get_coverage()."""
        self._ck(self.lib.hinge_filter_coverage_out(self.h, 1 if on else 0))

    def get_coverage(self):
        """(nbins[n], cov) with the bins of consecutive reads packed back to back (the layout coverage_bins() returns)."""
        n = self.r_end - self.r_begin + 1
        off = np.zeros(n + 1, np.int64)
        self._ck(self.lib.hinge_filter_get_coverage(self.h, _ptr(off), None, None, 0))
        nb = np.zeros(n, np.int32)
        raw = np.zeros(max(int(off[-1]), 1), np.int32)
        self._ck(self.lib.hinge_filter_get_coverage(self.h, _ptr(off), _ptr(nb), _ptr(raw), int(off[-1])))
        if n == 0 or int(nb.sum()) == 0:
            return nb, np.zeros(0, np.int32)
        idx = np.repeat(off[:-1] - np.concatenate([[0], np.cumsum(nb.astype(np.int64))[:-1]]), nb.astype(np.int64)) + np.arange(int(nb.sum()), dtype=np.int64)
        return nb, raw[idx]

    def attach_mask_table(self, dev_ptr):
        self._ck(self.lib.hinge_attach_mask_table(self.h, _VP(_ptr(dev_ptr)) if dev_ptr is not None else None))

    def attach_mean_cov(self, dev_ptr):
        self._ck(self.lib.hinge_attach_mean_cov(self.h, _VP(_ptr(dev_ptr)) if dev_ptr is not None else None))

    def clear_masks(self):
        self._ck(self.lib.hinge_clear_masks(self.h))

    def force_exact(self, mode):
        """0 normal, 1 everything through k_hinge_exact, 2 always use the in-kernel exact pile-up order."""
        self._ck(self.lib.hinge_debug_force_exact(self.h, int(mode)))

    def force_general_mask(self, on):
        """Run the general mask/annotate kernel even where the 20-bp fast kernel applies (tests)."""
        self._ck(self.lib.hinge_debug_force_general_mask(self.h, int(on)))

    def fallback_reads(self) -> int:
        """Reads the last mask/annotate pass handed from the 20-bp fast kernel back to the general one."""
        out = np.zeros(1, np.int64)
        self._ck(self.lib.hinge_debug_fallback_reads(self.h, _ptr(out)))
        return int(out[0])

    def heavy_items(self):
        """(half-size, full-size): undecided annotations of the last hinge pass by the k_hinge_call instance that took them."""
        out = np.zeros(2, np.int64)
        self._ck(self.lib.hinge_debug_heavy_items(self.h, _ptr(out)))
        return int(out[0]), int(out[1])

    def debug_pileup_order(self, keys: np.ndarray) -> np.ndarray:
        keys = np.ascontiguousarray(keys, dtype=np.int32)
        pos = np.zeros(max(len(keys), 1), np.int32)
        self._ck(self.lib.hinge_debug_pileup_order(self.h, len(keys), _ptr(keys), _ptr(pos)))
        return pos[:len(keys)]

    # ---- filter -------------------------------------------------------------------------------
    def filter_stats(self, p: FilterParams):
        self._ck(self.lib.hinge_filter_stats(self.h, C.byref(p)))

    def filter_median(self, p: FilterParams, lo: int, hi: int, fetch: bool = True) -> Optional[CovEstimate]:
        est = CovEstimate()
        self._ck(self.lib.hinge_filter_median(self.h, C.byref(p), lo, hi, C.byref(est) if fetch else None))
        return est if fetch else None

    def filter_stats_median(self, p: FilterParams, hist_dev=None, fetch: bool = False) -> Optional[CovEstimate]:
        """K1 + the median of the part's own reads in one launch.  hist_dev (device int32[4096 + 2]): sharded form, the part's
        histogram goes there instead (all-reduce it, then filter_median_from_hist)."""
        est = CovEstimate()
        self._ck(self.lib.hinge_filter_stats_median(self.h, C.byref(p), _VP(_ptr(hist_dev)) if hist_dev is not None else None,
                                                    C.byref(est) if (fetch and hist_dev is None) else None))
        return est if (fetch and hist_dev is None) else None

    def filter_median_hist(self, p: FilterParams, lo: int, hi: int, hist_dev):
        """Local histogram of the mean coverages of reads lo..hi into hist_dev (device int32[4096 + 2])."""
        self._ck(self.lib.hinge_filter_median_hist(self.h, C.byref(p), lo, hi, _VP(_ptr(hist_dev))))

    def filter_median_from_hist(self, p: FilterParams, hist_dev):
        self._ck(self.lib.hinge_filter_median_from_hist(self.h, C.byref(p), _VP(_ptr(hist_dev))))

    def set_min_cov(self, v: int):
        self._ck(self.lib.hinge_filter_set_min_cov(self.h, int(v)))

    def get_min_cov(self) -> int:
        v = C.c_int32()
        self._ck(self.lib.hinge_filter_get_min_cov(self.h, C.byref(v)))
        return int(v.value)

    def filter_mask_annotate(self, p: FilterParams):
        self._ck(self.lib.hinge_filter_mask_annotate(self.h, C.byref(p)))

    def filter_hinges(self, p: FilterParams):
        self._ck(self.lib.hinge_filter_hinges(self.h, C.byref(p)))

    def filter_sweep(self, p: FilterParams, fetch: bool = True) -> Optional[CovEstimate]:
        """The one-sweep pass of one part, synchronously (statistics, median, masks, annotations)."""
        est = CovEstimate()
        self._ck(self.lib.hinge_filter_sweep(self.h, C.byref(p), C.byref(est) if fetch else None))
        return est if fetch else None

    def spec_stats(self):
        """(passes verified, exact != predicted, outside the band, guard-band reads of the last pass, predicted, exact MIN_COV)"""
        out = (C.c_int64 * 6)()
        self._ck(self.lib.hinge_filter_spec_stats(self.h, out))
        return tuple(int(v) for v in out)

    def debug_spec(self, band: int = -1, sample: int = 0, bias: int = 0):
        self._ck(self.lib.hinge_debug_spec(self.h, int(band), int(sample), int(bias)))

    def filter_run(self, p: FilterParams):
        self._ck(self.lib.hinge_filter_run(self.h, C.byref(p)))

    def get_masks(self):
        n = self.r_end - self.r_begin + 1
        mask = np.zeros((n, 2), np.int32)
        cmask = np.zeros((n, 2), np.int32)
        flags = np.zeros(n, np.uint8)
        self._ck(self.lib.hinge_filter_get_masks(self.h, _ptr(mask), _ptr(cmask), _ptr(flags)))
        return mask, cmask, flags

    def get_annotations(self):
        n = self.r_end - self.r_begin + 1
        off = np.zeros(n + 1, np.int64)
        self._ck(self.lib.hinge_filter_get_annotations(self.h, _ptr(off), None, None, None))
        tot = int(off[-1])
        pos = np.zeros(max(tot, 1), np.int32)
        typ = np.zeros(max(tot, 1), np.int32)
        ish = np.zeros(max(tot, 1), np.uint8)
        self._ck(self.lib.hinge_filter_get_annotations(self.h, _ptr(off), _ptr(pos), _ptr(typ), _ptr(ish)))
        return off, pos[:tot], typ[:tot], ish[:tot]

    def coverage_bins(self, r0: int, r1: int, reso: int, cutoff: int):
        n = r1 - r0 + 1
        nb = np.zeros(n, np.int32)
        self._ck(self.lib.hinge_filter_coverage_bins(self.h, r0, r1, reso, cutoff, _ptr(nb), None, 0))
        tot = int(nb.astype(np.int64).sum())
        cov = np.zeros(max(tot, 1), np.int32)
        self._ck(self.lib.hinge_filter_coverage_bins(self.h, r0, r1, reso, cutoff, _ptr(nb), _ptr(cov), tot))
        return nb, cov[:tot]

    def counters(self):
        out = np.zeros(4, np.int64)
        self._ck(self.lib.hinge_filter_counters(self.h, _ptr(out)))
        return out

    # ---- maximal / layout ---------------------------------------------------------------------
    def set_traces(self, trace: np.ndarray, trace_off: np.ndarray, tlen: np.ndarray, tbytes: int = 1):
        trace = np.ascontiguousarray(trace, dtype=np.uint8)
        trace_off = np.ascontiguousarray(trace_off, dtype=np.int64)
        tlen = np.ascontiguousarray(tlen, dtype=np.int32)
        self._keep_tr = [trace, trace_off, tlen]
        self._ck(self.lib.hinge_set_traces(self.h, _ptr(trace), int(trace.shape[0]), _ptr(trace_off), _ptr(tlen), int(tbytes), 0))

    def set_las_image(self, image: np.ndarray, win_base: np.ndarray, rec_rel: np.ndarray, tbytes: int = 1):
        """The part form straight from the .las image (hinge_set_las_image): win_base[(n_ovl + 63) // 64 + 1], rec_rel[n_ovl]
        (formats.las_image_table / formats.image_windows)."""
        image = np.ascontiguousarray(image, dtype=np.uint8)
        win_base = np.ascontiguousarray(win_base, dtype=np.int64)
        rec_rel = np.ascontiguousarray(rec_rel, dtype=np.uint32)
        self._keep_tr = [image, win_base, rec_rel]
        self._ck(self.lib.hinge_set_las_image(self.h, _ptr(image), int(image.shape[0]), _ptr(win_base), _ptr(rec_rel), int(tbytes), 0))

    def set_eff_reads(self, eff: np.ndarray):
        eff = np.ascontiguousarray(eff, dtype=np.int32)
        self._ck(self.lib.hinge_set_eff_reads(self.h, _ptr(eff)))

    def trim_classify(self, sel: np.ndarray, a_of: np.ndarray, aln_threshold: int, theta: int, theta2: int) -> np.ndarray:
        sel = np.ascontiguousarray(sel, dtype=np.int64)
        a_of = np.ascontiguousarray(a_of, dtype=np.int32)
        out = np.zeros((max(len(sel), 1), 10), np.int32)
        self._ck(self.lib.hinge_trim_classify(self.h, len(sel), _ptr(sel), _ptr(a_of), aln_threshold, theta, theta2, _ptr(out)))
        return out[:len(sel)]

    def trim_classify_types(self, sel: np.ndarray, a_of: np.ndarray, aln_threshold: int, theta: int, theta2: int) -> np.ndarray:
        sel = np.ascontiguousarray(sel, dtype=np.int64)
        a_of = np.ascontiguousarray(a_of, dtype=np.int32)
        out = np.zeros(max(len(sel), 1), np.uint8)
        self._ck(self.lib.hinge_trim_classify_types(self.h, len(sel), _ptr(sel), _ptr(a_of), aln_threshold, theta, theta2, _ptr(out)))
        return out[:len(sel)]

    def trim_classify_part(self, n_ovl: int, aln_threshold: int, theta: int, theta2: int) -> np.ndarray:
        """Match type of every overlap of the current part (storage order)."""
        out = np.zeros(max(int(n_ovl), 1), np.uint8)
        self._ck(self.lib.hinge_trim_classify_part(self.h, aln_threshold, theta, theta2, _ptr(out)))
        return out[:int(n_ovl)]

    def trim_classify_part_full(self, n_ovl: int, aln_threshold: int, theta: int, theta2: int) -> np.ndarray:
        """All ten classification fields of every overlap of the current part (storage order)."""
        out = np.zeros((max(int(n_ovl), 1), 10), np.int32)
        self._ck(self.lib.hinge_trim_classify_part_full(self.h, aln_threshold, theta, theta2, _ptr(out)))
        return out[:int(n_ovl)]

    def set_trim(self, trim: bool):
        self._ck(self.lib.hinge_set_trim(self.h, 1 if trim else 0))

    def matching_position(self, q_ovl: np.ndarray, q_pos: np.ndarray) -> np.ndarray:
        q_ovl = np.ascontiguousarray(q_ovl, dtype=np.int64)
        q_pos = np.ascontiguousarray(q_pos, dtype=np.int32)
        out = np.zeros(max(len(q_ovl), 1), np.int32)
        self._ck(self.lib.hinge_matching_position(self.h, len(q_ovl), _ptr(q_ovl), _ptr(q_pos), _ptr(out)))
        return out[:len(q_ovl)]

    def select_edges(self, read_active, off_fwd, off_bwd, match_rec, h_off, h_rec, k_off, k_rec, hinge_tolerance: int, hinge_slack: int):
        """hinge_select_edges: (chosen[2, n_reads], hinge_pos[2, n_reads], poison_hits[n_matches])."""
        read_active = np.ascontiguousarray(read_active, dtype=np.uint8)
        n = len(read_active)
        off_fwd, off_bwd, h_off, k_off = (np.ascontiguousarray(v, dtype=np.int64) for v in (off_fwd, off_bwd, h_off, k_off))
        match_rec = np.ascontiguousarray(match_rec, dtype=np.int32).reshape(-1, 9)
        h_rec = np.ascontiguousarray(h_rec, dtype=np.int32).reshape(-1, 3)
        k_rec = np.ascontiguousarray(k_rec, dtype=np.int32).reshape(-1, 2)
        chosen = np.zeros((2, n), np.int32)
        hpos = np.zeros((2, n), np.int32)
        poison = np.zeros(max(len(match_rec), 1), np.int32)
        self._ck(self.lib.hinge_select_edges(self.h, n, _ptr(read_active), len(match_rec), _ptr(off_fwd), _ptr(off_bwd), _ptr(match_rec) if len(match_rec) else None,
                                             _ptr(h_off), _ptr(h_rec) if len(h_rec) else None, _ptr(k_off), _ptr(k_rec) if len(k_rec) else None,
                                             int(hinge_tolerance), int(hinge_slack), _ptr(chosen), _ptr(hpos), _ptr(poison)))
        return chosen, hpos, poison[:len(match_rec)]

    def filter_mask_annotate_async(self, p: FilterParams):
        self._ck(self.lib.hinge_filter_mask_annotate_async(self.h, C.byref(p)))

    def filter_hinges_async(self, p: FilterParams):
        self._ck(self.lib.hinge_filter_hinges_async(self.h, C.byref(p)))

    def check(self):
        self._ck(self.lib.hinge_filter_check(self.h))

    def profile_enable(self, max_launches: int):
        self._ck(self.lib.hinge_profile_enable(self.h, int(max_launches)))

    def profile_select(self, names=None):
        """Record events only for the named kernels (None = all)."""
        k = self.lib.hinge_profile_kernels()
        mask = 0
        for i in range(k):
            if names is None or self.lib.hinge_profile_kernel_name(i).decode() in names:
                mask |= 1 << i
        self._ck(self.lib.hinge_profile_select(self.h, mask))

    def profile_report(self):
        k = self.lib.hinge_profile_kernels()
        ms = np.zeros(k, np.float64)
        cnt = np.zeros(k, np.int64)
        self._ck(self.lib.hinge_profile_report(self.h, _ptr(ms), _ptr(cnt)))
        return {self.lib.hinge_profile_kernel_name(i).decode(): (float(ms[i]), int(cnt[i])) for i in range(k)}

    def timer_start(self):
        self._ck(self.lib.hinge_timer_start(self.h))

    def timer_stop_ms(self) -> float:
        ms = C.c_float()
        self._ck(self.lib.hinge_timer_stop_ms(self.h, C.byref(ms)))
        return float(ms.value)


CNS_ALN_DTYPE = np.dtype([("aread", "<i4"), ("bread", "<i4"), ("comp", "<i4"), ("abpos", "<i4"), ("aepos", "<i4"), ("bbpos", "<i4"), ("bepos", "<i4"),
                          ("tlen", "<i4"), ("trace_off", "<i8")])   # hinge_cns_alignment


class CnsStats(C.Structure):
    _fields_ = [("sum_coverage", C.c_int64)] + [(n, C.c_int32) for n in ("contig_length", "good_bases", "insertions", "deletions", "low_coverage_bases", "consensus_length")]


class Consensus:
    """`hinge consensus` over the C ABI (hinge_consensus_*): the two DBs' bases go to the GPU once, run() takes the alignments
    (rows of a formats.LasRecords) that vote."""

    def __init__(self, ctx: Context, draft_db: str, read_db: str):
        from . import formats
        self.ctx = ctx
        self.n = []
        for which, name in enumerate((draft_db, read_db)):
            idx = formats.read_db_index(name)
            rlen = np.ascontiguousarray(idx["rlen"], dtype=np.int32)
            boff = np.ascontiguousarray(idx["boff"], dtype=np.int64)
            bps = np.fromfile(formats.db_paths(name)[2], dtype=np.uint8)
            ctx._ck(ctx.lib.hinge_consensus_set_db(ctx.h, which, len(rlen), _ptr(rlen), _ptr(boff), _ptr(bps), bps.size))
            self.n.append(len(rlen))
        self.n_aln = 0

    def run(self, las, picks) -> None:
        picks = np.asarray(picks, dtype=np.int64)
        tb = 1 if las.tspace <= 125 else 2
        tr16 = las.trace.astype(np.uint16) if tb == 1 else np.ascontiguousarray(las.trace).view("<u2")
        a = np.zeros(len(picks), dtype=CNS_ALN_DTYPE)
        r = las.rec[picks]
        a["aread"], a["bread"], a["comp"] = r["aread"], r["bread"], r["flags"] & 1
        a["abpos"], a["aepos"], a["bbpos"], a["bepos"], a["tlen"] = r["abpos"], r["aepos"], r["bbpos"], r["bepos"], r["tlen"]
        a["trace_off"] = las.trace_off[picks] // tb
        tr16 = np.ascontiguousarray(tr16)
        self.n_aln = len(picks)
        self.ctx._ck(self.ctx.lib.hinge_consensus_run(self.ctx.h, len(picks), _ptr(a), _ptr(tr16), tr16.size, las.tspace))

    def contig(self, c: int):
        n = C.c_int64(0)
        st = CnsStats()
        self.ctx._ck(self.ctx.lib.hinge_consensus_get_contig(self.ctx.h, c, None, 0, C.byref(n), C.byref(st)))
        buf = np.zeros(max(n.value, 1), dtype=np.uint8)
        self.ctx._ck(self.ctx.lib.hinge_consensus_get_contig(self.ctx.h, c, _ptr(buf), n.value, C.byref(n), C.byref(st)))
        return buf[:n.value].tobytes(), st

    def offsets(self) -> np.ndarray:
        out = np.zeros(max(self.n_aln, 1), dtype=np.int32)
        if self.n_aln:
            self.ctx._ck(self.ctx.lib.hinge_consensus_get_offsets(self.ctx.h, _ptr(out)))
        return out[:self.n_aln]

    def indels(self, k: int) -> np.ndarray:
        n = C.c_int64(0)
        self.ctx._ck(self.ctx.lib.hinge_consensus_get_indels(self.ctx.h, k, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=np.int32)
        if n.value:
            self.ctx._ck(self.ctx.lib.hinge_consensus_get_indels(self.ctx.h, k, _ptr(out), n.value, C.byref(n)))
        return out[:n.value]


class Draft:
    """`hinge draft`'s two GPU steps over the C ABI (hinge_draft_*): both DB slots hold the read DB."""

    RUNG_DTYPE = np.dtype([("read", "<i4"), ("strand", "<i4"), ("start", "<i4"), ("end", "<i4")])

    def __init__(self, ctx: Context, read_db: str):
        self.ctx = ctx
        self.cns = Consensus(ctx, read_db, read_db)

    def mappings(self, las, picks):
        """List of uint32 arrays, one per picked record: B bases in front of every A base's column (bit 31: a gap in B there)."""
        picks = np.asarray(picks, dtype=np.int64)
        tb = 1 if las.tspace <= 125 else 2
        tr16 = np.ascontiguousarray(las.trace.astype(np.uint16) if tb == 1 else np.ascontiguousarray(las.trace).view("<u2"))
        a = np.zeros(len(picks), dtype=CNS_ALN_DTYPE)
        r = las.rec[picks]
        a["aread"], a["bread"], a["comp"] = r["aread"], r["bread"], r["flags"] & 1
        a["abpos"], a["aepos"], a["bbpos"], a["bepos"], a["tlen"] = r["abpos"], r["aepos"], r["bbpos"], r["bepos"], r["tlen"]
        a["trace_off"] = las.trace_off[picks] // tb
        off = np.concatenate([[0], np.cumsum((r["aepos"] - r["abpos"]).astype(np.int64))]).astype(np.int64)
        out = np.zeros(max(int(off[-1]), 1), np.uint32)
        self.ctx._ck(self.ctx.lib.hinge_draft_mappings(self.ctx.h, len(picks), _ptr(a), _ptr(tr16), tr16.size, las.tspace, _ptr(off), _ptr(out)))
        return [out[off[i]:off[i + 1]] for i in range(len(picks))]

    def ladders(self, ladders, templates, band: int = 150):
        """ladders: list of lists of (read, strand, start, end); templates: the template member of each.  Returns the consensus strings."""
        n = len(ladders)
        rung_off = np.concatenate([[0], np.cumsum([len(x) for x in ladders])]).astype(np.int64)
        rungs = np.zeros(max(int(rung_off[-1]), 1), dtype=self.RUNG_DTYPE)
        k = 0
        for ld in ladders:
            for g in ld:
                rungs[k] = g
                k += 1
        tm = np.ascontiguousarray(templates, dtype=np.int32)
        slot = np.concatenate([[0], np.cumsum([2 * (ld[t][3] - ld[t][2] + 1) for ld, t in zip(ladders, templates)])]).astype(np.int64)
        out = np.zeros(max(int(slot[-1]), 1), np.uint8)
        lens = np.zeros(max(n, 1), np.int32)
        self.ctx._ck(self.ctx.lib.hinge_draft_ladders(self.ctx.h, n, _ptr(rung_off), _ptr(rungs), _ptr(tm), band, _ptr(slot), _ptr(out), _ptr(lens)))
        return [out[slot[i]:slot[i] + lens[i]].tobytes().decode() for i in range(n)]


def median_from_hist_batch(ctxs, p: FilterParams, hist_dev, row_stride: int) -> None:
    """hinge_filter_median_from_hist for several contexts (one device, one stream) in one launch: context k takes row k of hist_dev."""
    lib = load_library()
    arr = (_VP * len(ctxs))(*[c.h for c in ctxs])
    rc = lib.hinge_filter_median_from_hist_batch(arr, len(ctxs), C.byref(p), _VP(_ptr(hist_dev)), int(row_stride))
    if rc != HINGE_OK:
        raise HingeError(rc, lib.hinge_last_error(ctxs[0].h).decode())


def _batch_call(fn_name: str, ctxs, *args) -> None:
    lib = load_library()
    arr = (_VP * len(ctxs))(*[c.h for c in ctxs])
    rc = getattr(lib, fn_name)(arr, len(ctxs), *args)
    if rc != HINGE_OK:
        raise HingeError(rc, lib.hinge_last_error(ctxs[0].h).decode())


def median_batch(ctxs, p: FilterParams, hist_dev=None, row_stride: int = 0) -> None:
    """hinge_filter_median_batch: every context's own-range median in one launch (hist_dev: the histogram form, row k for context k)."""
    _batch_call("hinge_filter_median_batch", ctxs, C.byref(p), _VP(_ptr(hist_dev)) if hist_dev is not None else None, int(row_stride))


def sweep_batch_async(ctxs, p: FilterParams, hist_dev=None, row_stride: int = 0) -> None:
    """hinge_filter_sweep_batch_async: the one-sweep pass's prediction + sweep + verifying median of several resident parts."""
    _batch_call("hinge_filter_sweep_batch_async", ctxs, C.byref(p), _VP(_ptr(hist_dev)) if hist_dev is not None else None, int(row_stride))


def finish_batch_async(ctxs, p: FilterParams) -> None:
    """hinge_filter_finish_batch_async: the guard-band reads with the exact MIN_COV."""
    _batch_call("hinge_filter_finish_batch_async", ctxs, C.byref(p))


def hinges_batch_async(ctxs, p: FilterParams) -> None:
    """hinge_filter_hinges_batch_async: hinge calling of several resident parts, one launch per kernel."""
    _batch_call("hinge_filter_hinges_batch_async", ctxs, C.byref(p))


def span16_pad() -> int:
    return int(load_library().hinge_span16_pad())


def pack_spans(row_ptr: np.ndarray, a_span: np.ndarray, rlen: np.ndarray):
    """What an ingest hands to set_pileups_packed besides the columns: (span16 or None, max_pile, spans_in_range).
    numpy restatement of the per-record work of hinge_amd/host/host_common.h LasPart::load."""
    row_ptr = np.asarray(row_ptr, dtype=np.int64)
    a_span = np.asarray(a_span, dtype=np.int32).reshape(-1, 2)
    n = int(a_span.shape[0])
    counts = np.diff(row_ptr)
    max_pile = int(counts.max()) if len(counts) else 0
    a_of = np.repeat(np.arange(len(counts), dtype=np.int64), counts)
    rl = np.asarray(rlen, dtype=np.int64)[a_of]
    in_range = bool(np.all((a_span[:, 0] >= 0) & (a_span[:, 1] >= 0) & (a_span[:, 0] <= rl) & (a_span[:, 1] <= rl))) if n else True
    span16 = None
    if n and in_range and int(np.max(rlen)) < 65536:
        span16 = np.zeros(n + span16_pad(), np.uint32)
        span16[:n] = a_span[:, 0].astype(np.uint32) | (a_span[:, 1].astype(np.uint32) << np.uint32(16))
    return span16, min(max_pile, 0x7FFFFFFF), in_range


def pile_bins(row_ptr: np.ndarray, a_span: np.ndarray, rlen: np.ndarray, reso: int = 40) -> np.ndarray:
    """What an ingest hands to set_pile_bins: per read of the block the bins of its plain coverage profile (0 for an empty pile-up,
    -1 for a coordinate outside [0, rlen] or 65 536+ overlaps).  numpy restatement of LasPart::finish_facts (host_common.h)."""
    row_ptr = np.asarray(row_ptr, dtype=np.int64)
    a_span = np.asarray(a_span, dtype=np.int32).reshape(-1, 2)
    counts = np.diff(row_ptr)
    nr = len(counts)
    out = np.zeros(nr, np.int32)
    if a_span.shape[0] == 0:
        return out
    end = int(row_ptr[-1])                    # (row_ptr may be a slice of a larger table with absolute offsets)
    mx = np.maximum(a_span[:end, 0], a_span[:end, 1]).astype(np.int64)
    mn = np.minimum(a_span[:end, 0], a_span[:end, 1]).astype(np.int64)
    if end == int(row_ptr[0]):
        return out
    has = counts > 0
    starts = row_ptr[:-1][has]
    rmx = np.maximum.reduceat(mx, starts)
    rmn = np.minimum.reduceat(mn, starts)
    rl = np.asarray(rlen, dtype=np.int64)[:nr][has]
    ok = (rmn >= 0) & (rmx <= rl) & (counts[has] < 65536)
    out[has] = np.where(ok, rmx // reso + 2, -1).astype(np.int32)
    return out


def pick_pairs(row_ptr, a_span, b_span, b_flag, lo: int, hi: int, accept_a=None, accept_b=None, self_before=None,
               two_matches: bool = True, n_sorts: int = 2):
    """hinge_pick_pairs: (sel, a_of) - the overlaps the reference classifies for the reads [lo, hi), in its own order."""
    lib = load_library()
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
    a_span = np.ascontiguousarray(a_span, dtype=np.int32)
    b_span = np.ascontiguousarray(b_span, dtype=np.int32)
    b_flag = np.ascontiguousarray(b_flag, dtype=np.uint32)
    n = len(row_ptr) - 1
    u8 = lambda v: None if v is None else np.ascontiguousarray(v, dtype=np.uint8)
    accept_a, accept_b = u8(accept_a), u8(accept_b)
    sb = None if self_before is None else np.ascontiguousarray(self_before, dtype=np.int32)
    args = (n, _ptr(row_ptr), _ptr(a_span), _ptr(b_span), _ptr(b_flag), _ptr(sb), _ptr(accept_a), _ptr(accept_b), int(lo), int(hi), 1 if two_matches else 0, int(n_sorts))
    cnt = lib.hinge_pick_pairs(*args, None, None, 0)
    if cnt < 0:
        raise HingeError(int(cnt), "hinge_pick_pairs: malformed pile-up arrays")
    sel = np.zeros(max(cnt, 1), np.int64)
    a_of = np.zeros(max(cnt, 1), np.int32)
    lib.hinge_pick_pairs(*args, _ptr(sel), _ptr(a_of), cnt)
    return sel[:cnt], a_of[:cnt]


def sort_order_desc(keys: np.ndarray, n_sorts: int = 1) -> np.ndarray:
    """hinge_sort_order_desc: the order std::sort(compare_overlap), run n_sorts times, leaves elements with these keys in."""
    keys = np.ascontiguousarray(keys, dtype=np.int64)
    perm = np.zeros(max(len(keys), 1), np.int32)
    rc = load_library().hinge_sort_order_desc(len(keys), _ptr(keys), int(n_sorts), _ptr(perm))
    if rc != 0:
        raise HingeError(rc, "hinge_sort_order_desc")
    return perm[:len(keys)]


MT_BCOVERA = 3   # match type "B covers A" (LAInterface.h:30-33)


def resolve_containment(active: np.ndarray, pairs: np.ndarray):
    """hinge_resolve_containment: active (uint8 [n_reads], modified in place) -> maximal-read mask; returns the
    container printed for every removed read (-1 elsewhere)."""
    lib = load_library()
    assert active.dtype == np.uint8 and active.flags.c_contiguous
    pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    containing = np.empty(len(active), np.int32)
    rc = lib.hinge_resolve_containment(len(active), _ptr(active), len(pairs), _ptr(pairs), _ptr(containing))
    if rc != 0:
        raise HingeError(rc, "hinge_resolve_containment: malformed candidate list")
    return containing
