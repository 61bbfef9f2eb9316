"""BASELINE config 5 (100 Mb genome, 100x, mean 7 kb): one rank's share, >= 1.25e8 overlaps, generated on the
device (hinge_amd/synth_device.py) and run through stats -> median -> mask + annotate -> hinges.  No oracle can hold this
many overlaps in seconds, so the full-size pass is checked through size-independent properties, and a sub-block of the very
same pile-ups goes through the CPU oracle byte for byte.  HINGE_CFG5_GENOME=<bp> shrinks the share (default 52 Mb)."""
import dataclasses
import filecmp
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, run_in, write_ini

pytestmark = pytest.mark.gpu
HINGE = os.path.join(ROOT, "hinge_amd", "bin", "hinge")


def _pass(ctx, P, lo, hi):
    ctx.filter_stats(P)
    est = ctx.filter_median(P, lo, hi, fetch=True)
    ctx.filter_mask_annotate(P)
    return est


def test_cfg5_share_against_the_oracle_digest():
    """The WHOLE one-rank share of BASELINE config 5 (52 Mb at 100x: 742 857 reads, 1.33e8 overlaps) against the CPU oracle AT
    SIZE: tests/golden/cfg5_share_digest.json holds sha256 digests of the oracle's mask / cmask rows, repeat annotations and
    hinge rows (tests/golden/make_cfg5_digest.py: 800 s in the build container - 290 s of generating and writing the 16 GB .las,
    510 s of single-thread oracle - which the GPU box does not spend).  The pile-ups are regenerated with torch's CPU generator
    (its stream does not depend on the machine; the input is digested too), moved to the GPU and run through the one-sweep
    pass + hinge calling."""
    import json
    import sys
    import time
    import torch
    from hinge_amd import capi, synth, synth_device
    from hinge_amd.config import default_filter_params
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_cfg5_digest as mk
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "cfg5_share_digest.json")))
    spec = dataclasses.replace(synth.CONFIGS["cfg5_share"], genome_len=want["genome"])
    t0 = time.time()
    p = synth_device.generate_pileups(spec, "cpu", span16_pad=0)
    n, m = p.n_reads, p.n_ovl
    assert (n, m) == (want["reads"], want["overlaps"])
    assert mk.input_digest(p) == want["input_sha256"], "the CPU generator no longer produces the pile-ups the digests were made on (torch %s there, %s here)" % (want["torch"], torch.__version__)
    print("cfg5 share generated on the host in %.0f s" % (time.time() - t0))
    dev = torch.device("cuda", 0)
    row_ptr, a_span, b_span, b_flag = (t.to(dev) for t in (p.row_ptr, p.a_span, p.b_span, p.b_flag))
    span16, max_pile, in_range = capi.pack_spans(p.row_ptr.numpy(), p.a_span.numpy(), p.rlen)
    span16 = torch.from_numpy(span16.view(np.int32)).to(dev)
    P = default_filter_params()
    ctx = capi.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_reads(p.rlen, None)
    ctx.set_pileups_packed(0, n - 1, row_ptr, a_span, b_span, b_flag, span16, max_pile, in_range, n_ovl=m, on_device=True)
    ctx.set_min_cov(P.min_cov)
    ctx.filter_sweep(P, fetch=True)
    ctx.filter_hinges(P)
    mask, cmask, _ = ctx.get_masks()
    off, pos, typ, ish = ctx.get_annotations()
    got = mk.result_digests(n, mask, cmask, off, pos, typ, ish)
    for key in ("n_annotations", "n_hinges", "mask", "cmask", "repeat", "hinges"):
        assert got[key] == want[key], "%s differs from the oracle's at full size" % key
    ctx.close()


def test_cfg5_share_full_size(oracle_lib, tmp_path):
    import torch
    from hinge_amd import capi, synth, synth_device
    from hinge_amd.config import default_filter_params
    genome = int(os.environ.get("HINGE_CFG5_GENOME", "52000000"))
    spec = dataclasses.replace(synth.CONFIGS["cfg5_share"], genome_len=genome)
    dev = torch.device("cuda", 0)
    p = synth_device.generate_pileups(spec, dev, span16_pad=capi.span16_pad())
    n, m = p.n_reads, p.n_ovl
    if genome >= 52_000_000:
        assert m >= 125_000_000, m
    assert p.span16 is not None and p.spans_in_range

    P = default_filter_params()
    ctx = capi.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_reads(p.rlen, None)
    ctx.set_pileups_packed(0, n - 1, p.row_ptr, p.a_span, p.b_span, p.b_flag, p.span16, p.max_pile, p.spans_in_range, n_ovl=m, on_device=True)
    ctx.coverage_out(True)
    ctx.set_min_cov(P.min_cov)
    est = _pass(ctx, P, 0, n - 1)
    ctx.filter_hinges(P)
    mask, cmask, _ = ctx.get_masks()
    off, pos, typ, ish = ctx.get_annotations()
    nb, cov = ctx.get_coverage()
    assert int(off[-1]) > 0 and int(ish.sum()) > 0, "no annotations / hinges: the comparison would be vacuous"

    # ---- property 1: K1's totals against the closed form, summed by torch over all overlaps ------------------------
    #   sum_k cov[k] of a read = sum_o (bin(aepos) - bin(abpos)),  bins K = bin(max aepos) + 1,  bin(v) = v // 40 + 1
    a_of = torch.repeat_interleave(torch.arange(n, device=dev), p.row_ptr[1:] - p.row_ptr[:-1])
    per_ovl = (p.a_span[:, 1].to(torch.int64) // 40) - (p.a_span[:, 0].to(torch.int64) // 40)
    per_read = torch.zeros(n, dtype=torch.int64, device=dev).index_add_(0, a_of, per_ovl)
    mx = torch.zeros(n, dtype=torch.int64, device=dev).scatter_reduce_(0, a_of, p.a_span[:, 1].to(torch.int64), "amax", include_self=True)
    counts = (p.row_ptr[1:] - p.row_ptr[:-1])
    K = torch.where(counts > 0, mx // 40 + 2, torch.zeros_like(mx))
    long_reads = torch.from_numpy(p.rlen >= 5000).to(dev)
    assert est.total_cov == int(per_read[long_reads].sum().item())
    assert est.num_slot == int(K[long_reads].sum().item())
    assert est.n_long == int(long_reads.sum().item())
    mean = torch.div(per_read, K.clamp(min=1), rounding_mode="trunc")[long_reads]
    assert est.cov_est == int(torch.sort(mean).values[mean.numel() // 2].item())       # nth_element(n / 2), filter.cpp:660

    # ---- property 2: the bins K2 stored, read by read: count and sum ------------------------------------------------
    assert np.array_equal(nb.astype(np.int64), K.cpu().numpy())
    seg = np.concatenate([[0], np.cumsum(nb.astype(np.int64))])
    sums = np.add.reduceat(np.concatenate([cov.astype(np.int64), [0]]), seg[:-1]) * (nb > 0)
    assert np.array_equal(sums, per_read.cpu().numpy())

    # ---- property 3: a split into two parts with the coverage estimate fixed (ec) gives the same masks, annotations, hinges
    P2 = default_filter_params()
    P2.est_cov = int(est.cov_est)
    h = n // 2
    ctx2 = capi.Context(0)
    ctx2.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx2.set_reads(p.rlen, None)
    got_mask = np.zeros_like(mask)
    got = {}
    for (lo, hi, with_hinges) in ((0, h - 1, False), (h, n - 1, True), (0, h - 1, True)):
        ctx2.set_pileups_packed(lo, hi, p.row_ptr, p.a_span, p.b_span, p.b_flag, p.span16, p.max_pile, p.spans_in_range, n_ovl=m, on_device=True)
        ctx2.set_min_cov(P.min_cov)
        _pass(ctx2, P2, lo, hi)
        if with_hinges:   # every mask exists by now (the table is the context's, it survives set_pileups)
            ctx2.filter_hinges(P2)
            got_mask[lo:hi + 1] = ctx2.get_masks()[0]
            got[lo] = ctx2.get_annotations()
    assert np.array_equal(got_mask, mask)
    off_a, pos_a, typ_a, ish_a = got[0]
    off_b, pos_b, typ_b, ish_b = got[h]
    assert np.array_equal(np.concatenate([np.diff(off_a), np.diff(off_b)]), np.diff(off))
    assert np.array_equal(np.concatenate([pos_a, pos_b]), pos) and np.array_equal(np.concatenate([typ_a, typ_b]), typ)
    assert np.array_equal(np.concatenate([ish_a, ish_b]), ish)
    ctx2.close()
    ctx.close()

    # ---- a sub-block of the same pile-ups, written as a real DB + .las: `hinge filter` vs the CPU oracle, byte for byte ----
    # (a block with annotations in it: take the 2500 reads around the read with the most annotations)
    busiest = int(np.argmax(np.diff(off)))
    r0 = max(0, min(n - 2500, busiest - 1250))
    d = synth_device.extract_block(p, r0, r0 + 2500)
    del p
    torch.cuda.empty_cache()
    wd = str(tmp_path / "blk")
    synth.write_dataset(d, wd, "G", write_bases=False)
    write_ini(os.path.join(wd, "nominal.ini"))
    assert run_in(wd, oracle_lib.oracle_filter, b"G", b"G.las", 0, b"O", b"nominal.ini", b"") == 0
    r = subprocess.run([HINGE, "filter", "--db", "G", "--las", "G.las", "-x", "H", "--config", "nominal.ini"], cwd=wd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    bad = [s for s in (".mas", ".cmas", ".repeat.txt", ".hinges.txt", ".coverage.txt") if not filecmp.cmp(os.path.join(wd, "O" + s), os.path.join(wd, "H" + s), shallow=False)]
    assert not bad, bad
    assert sum((len(l.split()) - 1) // 2 for l in open(os.path.join(wd, "O.repeat.txt"))) > 0
