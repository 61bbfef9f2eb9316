#!/usr/bin/env python
"""Golden outputs of the REFERENCE's own `consensus` program (oracle/_ref/consensus: src/consensus/consensus.cpp + the reference's
library sources, compiled unmodified by oracle/Makefile) on the seeded synthetic inputs of hinge_amd/synth_consensus.py
-> tests/golden/consensus_golden.json (sha256 of the FASTA and of stdout per configuration, plus a digest of the inputs) and
tests/golden/consensus_cns_tiny.fasta (the smallest case in full).  Run in the build container, where /root/reference exists:

    python tests/golden/make_consensus_golden.py
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

NAMES = ["cns_tiny", "cns_small", "cns_noisy", "cns_clean", "cns_twobyte", "cns_midsize"]


def sha_bytes(b):
    return hashlib.sha256(b).hexdigest()


def input_digest(d):
    import numpy as np
    h = hashlib.sha256()
    for c in d.contigs + d.reads:
        h.update(np.ascontiguousarray(c).tobytes())
    h.update(d.rec.tobytes())
    h.update(np.ascontiguousarray(d.trace).tobytes())
    return h.hexdigest()


def main():
    from hinge_amd import synth_consensus as sc
    ref = os.path.join(ROOT, "oracle", "_ref", "consensus")
    assert os.path.exists(ref), "oracle/_ref/consensus missing: make -C oracle (needs /root/reference)"
    out = {}
    for name in NAMES:
        d = sc.generate(sc.CONFIGS[name])
        tmp = tempfile.mkdtemp(prefix="cns_golden_")
        try:
            sc.write_dataset(d, tmp)
            r = subprocess.run([ref, "draft", "reads", "draft.reads.las", "ref.fasta", "nominal.ini"], cwd=tmp, stdout=subprocess.PIPE, check=True)
            fasta = open(os.path.join(tmp, "ref.fasta"), "rb").read()
            out[name] = {"input_sha256": input_digest(d), "fasta_sha256": sha_bytes(fasta), "stdout_sha256": sha_bytes(r.stdout),
                         "contigs": len(d.contigs), "reads": len(d.reads), "alignments": int(d.n_alignments), "fasta_bytes": len(fasta)}
            if name == "cns_tiny":
                with open(os.path.join(HERE, "consensus_cns_tiny.fasta"), "wb") as f:
                    f.write(fasta)
            print(name, out[name])
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    with open(os.path.join(HERE, "consensus_golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
