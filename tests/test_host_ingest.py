"""CPU: the executables' threaded .las ingest (hinge_amd/host/host_common.h, LasPart::load) against the numpy reader,
for 1 thread (sequential record walk) and many (guessed piece starts, verified against the chain)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from hinge_amd import formats

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="session")
def ingest_dump(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("host") / "ingest_dump")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-o", exe, os.path.join(ROOT, "tests", "host", "ingest_dump.cpp"), "-lz"], check=True)
    return exe


def _dump(exe, db, las, out, threads, how=None):
    r = subprocess.run([exe, db, las, out], env=dict(os.environ, HINGE_THREADS=str(threads)), stdout=subprocess.PIPE)
    if how is not None:
        how.append(r.stdout.decode().strip())
    return r.returncode


def _read_dump(path):
    raw = open(path, "rb").read()
    hdr = np.frombuffer(raw, np.int64, 4)
    pos = 32
    cols = []
    for dt in (np.int64, np.int32, np.int32, np.uint32, np.int64, np.int32, np.int64, np.int32, np.int64, np.int32, np.int32, np.uint32, np.int64, np.int64, np.int64, np.uint32):
        n = int(np.frombuffer(raw, np.int64, 1, pos)[0])
        pos += 8
        cols.append(np.frombuffer(raw, dt, n, pos).copy())
        pos += n * np.dtype(dt).itemsize
    assert pos == len(raw)
    return hdr, cols


@pytest.mark.parametrize("name,las", [("tiny", "G.las"), ("tiny_mlas", "G.2.las"), ("ties", "G.las"), ("long_reads", "G.las"), ("tspace200", "G.las"), ("edges", "G.las")])
def test_ingest_matches_numpy_reader(datasets, ingest_dump, tmp_path, name, las):
    wd, d = datasets(name)
    db, lasp = os.path.join(wd, "G"), os.path.join(wd, las)
    recs = formats.read_las(lasp)
    pile = formats.pileups_from_las(recs, d.rlen)
    ref = None
    for threads in (1, 3, 32):
        out = str(tmp_path / ("dump%d.bin" % threads))
        how = []
        assert _dump(ingest_dump, db, lasp, out, threads, how) == 0
        assert how == ["sequential" if threads == 1 else "pieces"], how
        hdr, c = _read_dump(out)
        row_ptr, a_span, b_span, b_flag, trace_off, tlen, rec_row_ptr, rec_b, rec_kept, self_a, self_span, span16, facts, img_ok, win_base, rec_rel = c
        # the image windows of hinge_set_las_image (LasPart::build_image_table) == the numpy helper the GPU tests use
        if pile.n_ovl:
            want_wb, want_rel = formats.las_image_table(recs, pile)
            assert img_ok.tolist() == [1]
            np.testing.assert_array_equal(win_base, want_wb)
            np.testing.assert_array_equal(rec_rel, want_rel)
        # what hinge_set_pileups_packed gets besides the columns: the numpy restatement (capi.pack_spans) must agree
        from hinge_amd import capi
        want16, want_pile, want_in_range = capi.pack_spans(pile.row_ptr, pile.a_span, d.rlen)
        assert facts.tolist() == [want_pile, int(want_in_range)]
        if want16 is None:
            assert len(span16) == 0 and int(d.rlen.max()) >= 65536
        else:
            np.testing.assert_array_equal(span16, want16)
        assert hdr[0] == recs.novl and hdr[1] == recs.tspace
        assert hdr[2] == recs.rec["aread"][0] and hdr[3] == recs.rec["aread"][-1]
        np.testing.assert_array_equal(row_ptr, pile.row_ptr)
        np.testing.assert_array_equal(a_span, pile.a_span.ravel())
        np.testing.assert_array_equal(b_span, pile.b_span.ravel())
        np.testing.assert_array_equal(b_flag, pile.b_flag)
        np.testing.assert_array_equal(tlen, recs.rec["tlen"][pile.las_index])
        np.testing.assert_array_equal(rec_b, recs.rec["bread"])
        kept = np.full(recs.novl, -1, np.int64)
        kept[pile.las_index] = np.arange(len(pile.las_index))
        np.testing.assert_array_equal(rec_kept, kept)
        np.testing.assert_array_equal(rec_row_ptr, np.concatenate([[0], np.cumsum(np.bincount(recs.rec["aread"], minlength=len(d.rlen)))]))
        np.testing.assert_array_equal(self_a, pile.self_a)
        np.testing.assert_array_equal(self_span, pile.self_span.ravel())
        # trace offsets: 12-byte header, 40-byte records, tlen * tbytes trace bytes each
        sizes = 40 + recs.rec["tlen"].astype(np.int64) * (1 if recs.tspace <= 125 else 2)
        starts = 12 + np.concatenate([[0], np.cumsum(sizes)[:-1]])
        np.testing.assert_array_equal(trace_off, starts[pile.las_index] + 40)
        if ref is None:
            ref = open(out, "rb").read()
        else:
            assert open(out, "rb").read() == ref, "result depends on the number of threads"


def test_fast_las_writer_writes_the_same_bytes(datasets, tmp_path):
    """hinge_amd/tools_c/synth_io.c (what write_dataset uses when it is built) == to_las_records + formats.write_las."""
    import filecmp
    from hinge_amd import synth
    if synth._synthio_lib() is None:
        pytest.skip("libhinge_synthio.so not built")
    for name in ("tiny", "edges", "chimera", "long_reads"):
        _, d = datasets(name)
        sel = np.nonzero(d.aread < d.n_reads // 2)[0]
        for k, s in enumerate((None, sel)):
            a, b = str(tmp_path / ("%s%d_c.las" % (name, k))), str(tmp_path / ("%s%d_np.las" % (name, k)))
            synth.write_las_file(d, a, s, fast=True)
            synth.write_las_file(d, b, s, fast=False)
            assert filecmp.cmp(a, b, shallow=False), (name, k)


def test_ingest_rejects_damaged_files(datasets, ingest_dump, tmp_path):
    wd, d = datasets("tiny")
    db = os.path.join(wd, "G")
    raw = open(os.path.join(wd, "G.las"), "rb").read()
    out = str(tmp_path / "o.bin")
    cases = {}
    cases["truncated"] = raw[: len(raw) * 2 // 3]
    cases["count_too_large"] = np.int64(np.frombuffer(raw, np.int64, 1)[0] + 5).tobytes() + raw[8:]
    mid = 12 + (len(raw) - 12) // 2
    cases["garbage_in_the_middle"] = raw[:mid] + bytes(4096) + raw[mid + 4096:]
    for name, blob in cases.items():
        p = str(tmp_path / (name + ".las"))
        open(p, "wb").write(blob)
        codes = {t: _dump(ingest_dump, db, p, out, t) for t in (1, 32)}
        assert codes[1] == codes[32], (name, codes)
        assert codes[1] != 0 or name == "garbage_in_the_middle", (name, codes)
        if codes[1] == 0:   # whatever the sequential walk makes of the garbage, the threaded walk must make the same of it
            a = str(tmp_path / "a.bin")
            b = str(tmp_path / "b.bin")
            assert _dump(ingest_dump, db, p, a, 1) == 0 and _dump(ingest_dump, db, p, b, 32) == 0
            assert open(a, "rb").read() == open(b, "rb").read()
    # records not sorted by A read: -2
    recs = formats.read_las(os.path.join(wd, "G.las"))
    import copy
    r2 = copy.copy(recs)
    r2.rec = recs.rec.copy()
    n = recs.novl
    r2.rec["aread"][n // 2] = recs.rec["aread"][-1]
    p = str(tmp_path / "unsorted.las")
    formats.write_las(p, r2)
    for t in (1, 32):
        assert _dump(ingest_dump, db, p, out, t) == 254


def test_fasta_and_paf_ingest_matches_the_oracle(datasets, ingest_dump, oracle_lib, tmp_path):
    """The executables' FASTA / PAF readers against the oracle's (which is pinned to the reference's kseq.h / paf.c)."""
    import ctypes
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_oracle_pinned import FASTA_CORNERS, FASTA_CORNERS_GOLDEN, _awkward_fasta_and_paf
    ip = ctypes.POINTER(ctypes.c_int)
    for name, blob in FASTA_CORNERS.items():
        p = str(tmp_path / (name + ".fa"))
        open(p, "wb").write(blob)
        out = str(tmp_path / "len.bin")
        assert subprocess.run([ingest_dump, "--fasta", p, out]).returncode == 0
        assert np.fromfile(out, np.int32).tolist() == FASTA_CORNERS_GOLDEN[name], name
    from hinge_amd import synth
    _, d = datasets("chimera")
    wd = str(tmp_path / "paf")
    synth.write_paf_dataset(d, wd, "G", gz=True)
    fa_c, paf_c = _awkward_fasta_and_paf(tmp_path)
    oracle_lib.oracle_load_las.restype = ctypes.c_long
    for fa, paf, n_rec in ((os.path.join(wd, "G.fasta.gz"), os.path.join(wd, "G.paf.gz"), d.novl), (fa_c, paf_c, 5)):
        x = np.zeros((n_rec + 2, 8), np.int32)
        assert oracle_lib.oracle_load_las(b"fasta:" + fa.encode(), b"paf:" + paf.encode(), x.ctypes.data_as(ip), ctypes.c_long(len(x))) == n_rec
        x = x[:n_rec]
        out = str(tmp_path / "paf.bin")
        assert subprocess.run([ingest_dump, "--paf", fa, paf, out], stdout=subprocess.DEVNULL).returncode == 0
        hdr, c = _read_dump(out)
        row_ptr, a_span, b_span, b_flag, trace_off, tlen, rec_row_ptr, rec_b, rec_kept, self_a, self_span, span16, facts, img_ok, _, _ = c
        assert img_ok.tolist() == [0]                          # (PAF input has no .las image: maximal keeps hinge_set_traces)
        assert hdr[0] == n_rec and hdr[2] == x[0, 0] and hdr[3] == x[-1, 0]
        order = np.argsort(x[:, 0], kind="stable")           # the reference files every line under its A read, in file order
        xs = x[order]
        keep = xs[:, 0] != xs[:, 1]
        np.testing.assert_array_equal(a_span.reshape(-1, 2), xs[keep][:, 2:4])
        np.testing.assert_array_equal(b_span.reshape(-1, 2), xs[keep][:, 4:6])
        np.testing.assert_array_equal(b_flag, xs[keep][:, 1].astype(np.uint32) | (xs[keep][:, 6].astype(np.uint32) << np.uint32(31)))
        np.testing.assert_array_equal(rec_b, xs[:, 1])
        assert (tlen == 0).all()
        sf = x[x[:, 0] == x[:, 1]]                            # self overlaps stay in file order
        np.testing.assert_array_equal(self_a, sf[:, 0])
        np.testing.assert_array_equal(self_span.reshape(-1, 4), sf[:, 2:6])
