#!/usr/bin/env python
"""tests/golden/draft_falcon_golden.json: seeded ladders and what the REFERENCE's own falcon code (lib/DW_banded.c, lib/falcon.c,
compiled unmodified into oracle/_ref/libhinge_ref.so) makes of them through ref_falcon_ladder (the marshalling of
draft.cpp:597-691), plus the two gapped rows of its aligner for the first member.  Data only: inputs and outputs.

Run in the build container:  python tests/golden/make_draft_golden.py"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import draft_common as dc  # noqa: E402


def main():
    ref = oracle.ref_lib()
    assert ref is not None, "build oracle/_ref first (make -C oracle)"
    dc.bind_ref(ref)
    rng = np.random.default_rng(20260930)
    cases = []
    for case in range(80):
        mem, mx = dc.random_ladder(rng, case)
        n, cns = dc.ladder_call(ref.ref_falcon_ladder, mem, mx)
        cap = 3 * max(len(m) for m in mem) + 100
        q, t = ctypes.create_string_buffer(cap), ctypes.create_string_buffer(cap)
        a = ref.ref_falcon_align(mem[0].encode(), mem[mx].encode(), 150, q, t, cap)
        cases.append({"members": mem, "mx": mx, "cns": cns, "aln_len": int(a), "q": q.value.decode(), "t": t.value.decode()})
        assert n == len(cns)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "draft_falcon_golden.json")
    with open(path, "w") as f:
        json.dump(cases, f)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
