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
