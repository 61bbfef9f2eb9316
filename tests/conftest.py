import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NOMINAL_INI = """[filter]
length_threshold = 1000;
quality_threshold = 0.23;
n_iter = 3; // filter iteration
aln_threshold = 1000;
min_cov = 5;
cut_off = 300;
theta = 300;
use_qv = true;

[running]
n_proc = 12;

[layout]
hinge_slack = 1000
min_connected_component_size = 8
"""


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle
    return oracle.oracle_lib()


@pytest.fixture(scope="session")
def ref_lib():
    import oracle
    lib = oracle.ref_lib()
    if lib is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    return lib


def write_ini(path, extra_filter="", extra_layout=""):
    """nominal.ini with overrides.  A key that nominal.ini already has is REPLACED in place: the
    reference's INIReader joins duplicate keys with a newline and strtol then reads the first value
    (src/lib/INIReader.cpp:76-79), so an appended duplicate would change nothing."""
    lines = NOMINAL_INI.split("\n")

    def apply(section, extra):
        lo = lines.index("[%s]" % section)
        hi = next((i for i in range(lo + 1, len(lines)) if lines[i].startswith("[")), len(lines))
        for ov in [x for x in extra.split("\n") if x.strip()]:
            key = ov.split("=")[0].strip()
            hit = [i for i in range(lo + 1, hi) if lines[i].split("=")[0].strip() == key]
            if hit:
                lines[hit[0]] = ov
            else:
                lines.insert(hi, ov)
                hi += 1

    apply("filter", extra_filter)
    apply("layout", extra_layout)
    with open(path, "w") as f:
        f.write("\n".join(lines))
    return path


@pytest.fixture(scope="session")
def datasets(tmp_path_factory):
    """Synthetic DB + .las datasets written once per session: name -> (dir, SynthData)."""
    from hinge_amd import synth
    root = tmp_path_factory.mktemp("synth")
    cache = {}

    def get(name):
        if name not in cache:
            d = synth.generate(synth.CONFIGS[name])
            wd = os.path.join(str(root), name)
            synth.write_dataset(d, wd, "G")
            write_ini(os.path.join(wd, "nominal.ini"))
            cache[name] = (wd, d)
        return cache[name]

    return get


def free_port():
    """A TCP port nobody listens on right now, for a test's torch.distributed rendezvous on 127.0.0.1.  (Ports derived from the
    pid collided: half of the old choices lay in the kernel's ephemeral range, where any outgoing connection of the box may sit -
    `EADDRINUSE` once in a few hundred runs, and with `pytest -x` the end of the suite.)"""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_in(wd, fn, *args):
    old = os.getcwd()
    os.chdir(wd)
    try:
        return fn(*args)
    finally:
        os.chdir(old)


def clone_dataset(src, dst):
    """Copy the input files of a dataset directory (not stage outputs) into dst."""
    os.makedirs(dst, exist_ok=True)
    for f in os.listdir(src):
        if f.endswith((".las", ".db", ".ini")) or f.startswith(".G."):
            shutil.copy(os.path.join(src, f), os.path.join(dst, f))
    return dst
