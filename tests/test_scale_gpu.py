"""Size-independent property at a size where byte offsets pass 2^32: K independent copies of one data set, laid out as one
part, must give K copies of the same masks / annotations / hinges (every copy has the same median, so the pass over the
concatenation is the pass over one copy, K times)."""
import dataclasses

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_replicated_part_beyond_4gib():
    import torch
    from hinge_amd import capi, synth
    from hinge_amd.config import default_filter_params
    spec = dataclasses.replace(synth.CONFIGS["cfg2_ecoli160"], genome_len=500_000, n_repeat_families=1, repeat_copies=(3, 3))
    d = synth.generate(spec)
    pile = synth.to_pileups(d)
    K = int(np.ceil((4.5 * 2 ** 30) / (pile.n_ovl * 8)))          # a_span alone > 4 GiB
    n1, m1 = d.n_reads, pile.n_ovl
    dev = torch.device("cuda", 0)
    a = torch.from_numpy(pile.a_span).to(dev).repeat(K, 1)
    b = torch.from_numpy(pile.b_span).to(dev).repeat(K, 1)
    f1 = torch.from_numpy(pile.b_flag.view(np.int32).astype(np.int64) & 0xFFFFFFFF).to(dev)
    copy = torch.arange(K, device=dev, dtype=torch.int64).repeat_interleave(m1)
    f = ((f1.repeat(K) & 0x7FFFFFFF) + copy * n1) | (f1.repeat(K) & 0x80000000)
    f = f.to(torch.int64).bitwise_and(0xFFFFFFFF)
    f = torch.where(f >= 2 ** 31, f - 2 ** 32, f).to(torch.int32)
    del copy, f1
    rp1 = torch.from_numpy(pile.row_ptr[:-1]).to(dev)
    row_ptr = torch.cat([(rp1 + c * m1) for c in range(K)] + [torch.tensor([K * m1], dtype=torch.int64, device=dev)])
    rlen = np.tile(d.rlen, K).astype(np.int32)
    assert a.numel() * 4 > 2 ** 32 and row_ptr[-1].item() == K * m1

    P = default_filter_params()
    ctx = capi.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_reads(rlen, None)
    ctx.set_pileups(0, K * n1 - 1, row_ptr, a, b, f, n_ovl=K * m1, on_device=True)
    ctx.set_min_cov(P.min_cov)
    ctx.filter_stats(P)
    est = ctx.filter_median(P, 0, K * n1 - 1, fetch=True)
    ctx.filter_mask_annotate(P)
    ctx.filter_hinges(P)
    mask, cmask, flags = ctx.get_masks()
    off, pos, typ, ish = ctx.get_annotations()

    # the same pass over ONE copy
    ctx1 = capi.Context(0)
    ctx1.set_reads(d.rlen.astype(np.int32), None)
    ctx1.set_pileups(0, n1 - 1, pile.row_ptr, pile.a_span, pile.b_span, pile.b_flag, n_ovl=m1, on_device=False)
    ctx1.set_min_cov(P.min_cov)
    ctx1.filter_stats(P)
    est1 = ctx1.filter_median(P, 0, n1 - 1, fetch=True)
    ctx1.filter_mask_annotate(P)
    ctx1.filter_hinges(P)
    mask1, cmask1, flags1 = ctx1.get_masks()
    off1, pos1, typ1, ish1 = ctx1.get_annotations()

    assert est.cov_est == est1.cov_est and est.n_long == K * est1.n_long
    assert np.array_equal(mask.reshape(K, n1, 2), np.broadcast_to(mask1.reshape(1, n1, 2), (K, n1, 2)))
    assert np.array_equal(cmask.reshape(K, n1, 2), np.broadcast_to(cmask1.reshape(1, n1, 2), (K, n1, 2)))
    cnt = np.diff(off).reshape(K, n1)
    assert np.array_equal(cnt, np.broadcast_to(np.diff(off1).reshape(1, n1), (K, n1)))
    na1 = int(off1[-1])
    assert na1 > 0 and int(ish1.sum()) > 0
    assert np.array_equal(pos.reshape(K, na1), np.broadcast_to(pos1.reshape(1, na1), (K, na1)))
    assert np.array_equal(typ.reshape(K, na1), np.broadcast_to(typ1.reshape(1, na1), (K, na1)))
    assert np.array_equal(ish.reshape(K, na1), np.broadcast_to(ish1.reshape(1, na1), (K, na1)))
