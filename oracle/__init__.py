"""TEST INFRASTRUCTURE - ctypes loaders for the CPU oracle and the compiled-reference shim.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def _load(path):
    return ctypes.CDLL(path) if os.path.exists(path) else None


def oracle_lib():
    lib = _load(os.path.join(_HERE, "libhinge_oracle.so"))
    if lib is None:
        raise RuntimeError("oracle/libhinge_oracle.so missing: run `make -C oracle`")
    c = ctypes
    lib.oracle_filter.argtypes = [c.c_char_p, c.c_char_p, c.c_int, c.c_char_p, c.c_char_p, c.c_char_p]
    lib.oracle_maximal.argtypes = [c.c_char_p, c.c_char_p, c.c_int, c.c_char_p, c.c_char_p]
    lib.oracle_layout.argtypes = [c.c_char_p, c.c_char_p, c.c_int, c.c_char_p, c.c_char_p, c.c_char_p]
    lib.oracle_probe_means.argtypes = [c.POINTER(c.c_int), c.c_long, c.POINTER(c.c_int)]
    lib.oracle_probe_means.restype = c.c_long
    _common(lib, "oracle")
    return lib


def ref_lib():
    """The reference's own compiled library code (None when oracle/_ref was never built)."""
    lib = _load(os.path.join(_HERE, "_ref", "libhinge_ref.so"))
    if lib is not None:
        _common(lib, "ref")
    return lib


def _common(lib, pre):
    c = ctypes
    ip = c.POINTER(c.c_int)
    u16p = c.POINTER(c.c_uint16)
    f = getattr(lib, pre + "_profile_coverage")
    f.argtypes = [c.c_int, ip, ip, c.c_int, c.c_int, ip, c.c_int]
    f.restype = c.c_int
    f = getattr(lib, pre + "_process_alignment")
    f.argtypes = [ip, u16p, c.c_int, c.c_int, c.c_int, c.c_int, ip]
    f.restype = None
    f = getattr(lib, pre + "_matching_position")
    f.argtypes = [c.c_int] * 5 + [u16p, c.c_int, c.c_int]
    f.restype = c.c_int
    f = getattr(lib, pre + "_sort_perm")
    f.argtypes = [c.c_int, ip, c.c_int, ip]
    f.restype = None
    f = getattr(lib, pre + "_load_las")
    f.argtypes = [c.c_char_p, c.c_char_p, ip, c.c_long]
    f.restype = c.c_long
    for nm, rt, dt in (("_ini_int", c.c_long, c.c_long), ("_ini_bool", c.c_int, c.c_int), ("_ini_real", c.c_double, c.c_double)):
        f = getattr(lib, pre + nm)
        f.argtypes = [c.c_char_p, c.c_char_p, c.c_char_p, dt]
        f.restype = rt
    f = getattr(lib, pre + "_ini_error")
    f.argtypes = [c.c_char_p]
    f.restype = c.c_int
