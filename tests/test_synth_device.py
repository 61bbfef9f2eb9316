"""The device-side pile-up generator of config 5 (hinge_amd/synth_device.py), run on the CPU at a small size: the columns it
builds are a valid pile-up set in the layout of include/hinge_hip.h, and a sub-block of it round-trips through the on-disk
formats into the CPU oracle."""
import dataclasses
import os

import numpy as np

from conftest import run_in, write_ini


def _small():
    from hinge_amd import synth, synth_device
    spec = dataclasses.replace(synth.CONFIGS["cfg5_share"], genome_len=600_000, n_repeat_families=2)
    return synth_device.generate_pileups(spec, "cpu", span16_pad=256)


def test_device_generator_builds_valid_pileups():
    p = _small()
    rp = p.row_ptr.numpy()
    assert rp[0] == 0 and rp[-1] == p.n_ovl and (np.diff(rp) >= 0).all() and p.n_ovl > 100 * p.n_reads
    a = np.repeat(np.arange(p.n_reads), np.diff(rp))
    bf = p.b_flag.numpy().view(np.uint32)
    b = (bf & np.uint32(0x7FFFFFFF)).astype(np.int64)
    comp = (bf >> np.uint32(31)).astype(np.int64)
    asp, bsp = p.a_span.numpy(), p.b_span.numpy()
    assert (a != b).all()
    assert (asp[:, 0] >= 0).all() and (asp[:, 1] <= p.rlen[a]).all() and (asp[:, 1] - asp[:, 0] >= 100).all()
    assert (bsp[:, 0] >= 0).all() and (bsp[:, 1] <= p.rlen[b]).all() and (bsp[:, 1] - bsp[:, 0] >= 100).all()
    key = (a << 42) | (b << 22) | (comp << 21) | asp[:, 0]          # LAsort order
    assert (np.diff(key) >= 0).all()
    pairs = set(zip(a.tolist(), b.tolist()))
    assert all((y, x) in pairs for x, y in list(pairs)[:5000])     # overlaps come in both directions
    assert p.spans_in_range and p.max_pile == int(np.diff(rp).max())
    s16 = p.span16.numpy().view(np.uint32)[:p.n_ovl]
    assert np.array_equal(s16 & 0xFFFF, asp[:, 0].astype(np.uint32)) and np.array_equal(s16 >> 16, asp[:, 1].astype(np.uint32))
    # both strands, and repeat-induced records (a pair with two or more records, or a distant partner) exist
    assert 0.3 < comp.mean() < 0.7


def test_sub_block_runs_through_the_oracle(oracle_lib, tmp_path):
    from hinge_amd import synth, synth_device
    p = _small()
    d = synth_device.extract_block(p, 200, 900)
    assert d.aread.min() == 200 and d.aread.max() == 899 and d.n_reads == p.n_reads
    wd = str(tmp_path / "blk")
    synth.write_dataset(d, wd, "G", write_bases=False)
    write_ini(os.path.join(wd, "nominal.ini"))
    assert run_in(wd, oracle_lib.oracle_filter, b"G", b"G.las", 0, b"G", b"nominal.ini", b"") == 0
    assert sum(1 for _ in open(os.path.join(wd, "G.mas"))) == 700
