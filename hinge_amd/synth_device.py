"""Device-side pile-up generator for the large configurations (BASELINE config 5: 100 Mb genome, 100x, mean 7 kb).

SURVEY.md 8(d) prescribes that config 5 is generated directly on the device: a 10^8..10^9-record `.las` would be tens of
GB on disk and minutes of numpy.  This module restates hinge_amd.synth's overlap model (true interval intersections of at
least `min_ovl` bases between reads sampled uniformly on both strands, end-point jitter, small B-side indels, plus the
repeat-induced cross-copy alignments) with torch ops, so the pile-up columns of `hinge_set_pileups_packed` are built where
they are used.  The repeat-induced records (a few hundred thousand) come from the numpy code of hinge_amd.synth itself and
are merged in.  Test / bench infrastructure: the product never imports it.

The generator is seeded and deterministic per (spec, seed, device type); it is NOT the same random stream as
hinge_amd.synth.generate.  `extract_block` turns a contiguous range of A reads back into a hinge_amd.synth.SynthData
(host arrays), so a sub-block can be written as a real DB + .las and run through the CPU oracle.
"""
from __future__ import annotations

import dataclasses
from typing import Optional

import numpy as np
import torch

from . import synth


@dataclasses.dataclass
class DevicePileups:
    spec: synth.SynthSpec
    n_reads: int
    rlen: np.ndarray                 # int32 [n_reads] (host: the C ABI takes the read table from the host)
    row_ptr: torch.Tensor            # int64 [n_reads + 1]
    a_span: torch.Tensor             # int32 [n, 2]
    b_span: torch.Tensor             # int32 [n, 2]
    b_flag: torch.Tensor             # int32 [n] (bits of uint32 bread | comp << 31)
    span16: Optional[torch.Tensor]   # int32 [n + pad] (bits of abpos | aepos << 16) or None
    max_pile: int
    spans_in_range: bool

    @property
    def n_ovl(self) -> int:
        return int(self.b_flag.shape[0])


def _randint(gen, low, high_excl, size, device):
    return torch.randint(low, high_excl, (size,), generator=gen, device=device, dtype=torch.int64)


def generate_pileups(spec: synth.SynthSpec, device, span16_pad: int = 0, chunk_reads: int = 1 << 16) -> DevicePileups:
    """Pile-ups of one block (all reads of the spec's genome) as device tensors in the layout of include/hinge_hip.h."""
    assert spec.chimera_frac == 0 and spec.tie_quantum == 0 and spec.short_reads == 0 and spec.orphan_reads == 0 \
        and spec.self_overlap_reads == 0 and spec.orphan_ends == 0, "the device generator restates the plain model only"
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(spec.seed))
    rng = np.random.default_rng(spec.seed)          # host stream: repeat placement + the repeat-induced records
    G = spec.genome_len
    fam_len, copies = synth.plant_repeats(spec, rng)

    mean_len = (spec.len_min + spec.len_max) / 2 if spec.len_dist == "uniform" else spec.len_mean
    n = max(4, int(round(G * spec.coverage / mean_len)))
    if spec.len_dist == "uniform":
        lens = _randint(gen, spec.len_min, spec.len_max + 1, n, dev)
    else:
        mu = float(np.log(spec.len_mean) - 0.5 * spec.len_sigma ** 2)
        z = torch.randn(n, generator=gen, device=dev, dtype=torch.float64)
        lens = torch.exp(mu + spec.len_sigma * z).clamp(spec.len_min, spec.len_max).to(torch.int64)
    lens = torch.minimum(lens, torch.tensor(G // 2, device=dev))
    starts = (torch.rand(n, generator=gen, device=dev, dtype=torch.float64) * (G - lens + 1).to(torch.float64)).to(torch.int64)
    starts = torch.minimum(starts, G - lens)
    strand = torch.where(torch.rand(n, generator=gen, device=dev) < 0.5, 1, -1).to(torch.int64)
    # read ids are independent of the genome position (starts are i.i.d.), like a shuffled DB

    # sorted-by-start view for the sweep
    order = torch.argsort(starts, stable=True)
    s0 = starts[order]
    s1 = s0 + lens[order]
    hi = torch.searchsorted(s0, s1 - spec.min_ovl, right=True)
    lo = torch.arange(n, device=dev) + 1
    cnt = (hi - lo).clamp(min=0)
    csum = torch.cumsum(cnt, 0)

    cols = {k: [] for k in ("a", "b", "ab", "ae", "bb", "be", "comp")}
    jit, ind = spec.end_jitter, spec.indel_max
    min_keep = max(spec.min_ovl - 2 * jit, 200)

    def emit(x, y, lo_g, hi_g):
        """Directed records x -> y for pairs whose genome intersection is [lo_g, hi_g) (before jitter)."""
        if jit > 0:
            lo_g = lo_g + _randint(gen, 0, jit + 1, lo_g.shape[0], dev)
            hi_g = hi_g - _randint(gen, 0, jit + 1, hi_g.shape[0], dev)
        ok = hi_g - lo_g >= min_keep
        x, y, lo_g, hi_g = x[ok], y[ok], lo_g[ok], hi_g[ok]
        sx, ex, stx = starts[x], starts[x] + lens[x], strand[x]
        sy, ey, sty = starts[y], starts[y] + lens[y], strand[y]
        ab = torch.where(stx > 0, lo_g - sx, ex - hi_g)
        ae = torch.where(stx > 0, hi_g - sx, ex - lo_g)
        bb = torch.where(sty > 0, lo_g - sy, ey - hi_g)
        be = torch.where(sty > 0, hi_g - sy, ey - lo_g)
        if ind > 0:
            d = _randint(gen, 0, ind + 1, bb.shape[0], dev)
            side = torch.rand(bb.shape[0], generator=gen, device=dev) < 0.5
            bb = torch.where(side, bb + d, bb)
            be = torch.where(side, be, be - d)
        good = (ab >= 0) & (ae <= lens[x]) & (bb >= 0) & (be <= lens[y]) & (ae - ab >= 100) & (be - bb >= 100)
        cols["a"].append(x[good].to(torch.int32)); cols["b"].append(y[good].to(torch.int32))
        cols["ab"].append(ab[good].to(torch.int32)); cols["ae"].append(ae[good].to(torch.int32))
        cols["bb"].append(bb[good].to(torch.int32)); cols["be"].append(be[good].to(torch.int32))
        cols["comp"].append((stx != sty)[good].to(torch.int32))

    # the pair expansion is done for chunks of sorted reads so that the temporaries stay bounded
    for c0 in range(0, n, chunk_reads):
        c1 = min(n, c0 + chunk_reads)
        cc = cnt[c0:c1]
        tot = int(cc.sum().item())
        if tot == 0:
            continue
        i_idx = torch.repeat_interleave(torch.arange(c0, c1, device=dev), cc)
        first = csum[c0:c1] - cc - (csum[c0 - 1] if c0 > 0 else 0)
        j_idx = torch.arange(tot, device=dev) - torch.repeat_interleave(first, cc) + i_idx + 1
        lo_g = s0[j_idx]
        hi_g = torch.minimum(s1[i_idx], s1[j_idx])
        ok = hi_g - lo_g >= spec.min_ovl
        i_idx, j_idx, lo_g, hi_g = order[i_idx[ok]], order[j_idx[ok]], lo_g[ok], hi_g[ok]
        emit(i_idx, j_idx, lo_g, hi_g)
        emit(j_idx, i_idx, lo_g, hi_g)

    # repeat-induced records: the numpy generator's own code on the host copy of the read table
    if fam_len:
        h_starts, h_lens, h_strand = starts.cpu().numpy(), lens.cpu().numpy(), strand.cpu().numpy()
        rec = synth._Records(spec, rng)
        synth.repeat_records(rec, fam_len, copies, np.arange(n, dtype=np.int64), np.zeros(n, np.int64), h_starts, h_starts + h_lens, h_strand)
        if rec.a:
            ra, rb = np.concatenate(rec.a), np.concatenate(rec.b)
            rab, rae, rbb, rbe = (np.concatenate(v) for v in (rec.ab, rec.ae, rec.bb, rec.be))
            rcomp = np.concatenate(rec.comp)
            ok = (rab >= 0) & (rae <= h_lens[ra]) & (rbb >= 0) & (rbe <= h_lens[rb]) & (rae - rab >= 100) & (rbe - rbb >= 100) & (ra != rb)
            for k, v in (("a", ra), ("b", rb), ("ab", rab), ("ae", rae), ("bb", rbb), ("be", rbe), ("comp", rcomp)):
                cols[k].append(torch.from_numpy(np.ascontiguousarray(v[ok].astype(np.int32))).to(dev))

    a = torch.cat(cols["a"]); b = torch.cat(cols["b"]); comp = torch.cat(cols["comp"])
    ab = torch.cat(cols["ab"]); ae = torch.cat(cols["ae"]); bb = torch.cat(cols["bb"]); be = torch.cat(cols["be"])
    del cols
    # LAsort order: (aread, bread, comp, abpos)
    kb = int(n).bit_length()
    pb = int(lens.max().item()).bit_length() + 1
    assert 2 * kb + 1 + pb <= 62
    key = ((((a.to(torch.int64) << kb) | b.to(torch.int64)) << 1) | comp.to(torch.int64)) << pb | ab.to(torch.int64)
    perm = torch.argsort(key, stable=True)
    del key
    a, b, comp, ab, ae, bb, be = (v[perm] for v in (a, b, comp, ab, ae, bb, be))
    del perm
    counts = torch.bincount(a.to(torch.int64), minlength=n)
    row_ptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    row_ptr[1:] = torch.cumsum(counts, 0)
    m = int(a.shape[0])
    a_span = torch.stack([ab, ae], dim=1).contiguous()
    b_span = torch.stack([bb, be], dim=1).contiguous()
    b_flag = (b.to(torch.int64) | (comp.to(torch.int64) << 31))
    b_flag = torch.where(b_flag >= 2 ** 31, b_flag - 2 ** 32, b_flag).to(torch.int32)
    rl_a = lens[a.to(torch.int64)]
    in_range = bool(((ab >= 0) & (ae >= 0) & (ab <= rl_a) & (ae <= rl_a)).all().item()) if m else True
    span16 = None
    if m and in_range and int(lens.max().item()) < 65536:
        span16 = torch.zeros(m + span16_pad, dtype=torch.int32, device=dev)
        v = ab.to(torch.int64) | (ae.to(torch.int64) << 16)
        span16[:m] = torch.where(v >= 2 ** 31, v - 2 ** 32, v).to(torch.int32)
    return DevicePileups(spec=spec, n_reads=n, rlen=lens.to(torch.int32).cpu().numpy(), row_ptr=row_ptr, a_span=a_span, b_span=b_span,
                         b_flag=b_flag, span16=span16, max_pile=int(counts.max().item()) if m else 0, spans_in_range=in_range)


def extract_block(p: DevicePileups, r0: int, r1: int, tspace: int = 100) -> synth.SynthData:
    """The records of A reads [r0, r1) as a host SynthData over ALL reads of the DB (write it with synth.write_dataset: a DB
    of every read and a .las that holds only this block, as one `name.k.las` of a --mlas run would)."""
    lo, hi = int(p.row_ptr[r0].item()), int(p.row_ptr[r1].item())
    counts = (p.row_ptr[r0 + 1:r1 + 1] - p.row_ptr[r0:r1]).cpu().numpy()
    aread = np.repeat(np.arange(r0, r1, dtype=np.int32), counts)
    bf = p.b_flag[lo:hi].cpu().numpy().view(np.uint32)
    a_span = p.a_span[lo:hi].cpu().numpy()
    b_span = p.b_span[lo:hi].cpu().numpy()
    spec = dataclasses.replace(p.spec, tspace=tspace, n_blocks=1)
    return synth.SynthData(spec=spec, rlen=p.rlen.copy(), aread=aread, bread=(bf & np.uint32(0x7FFFFFFF)).astype(np.int32),
                           comp=(bf >> np.uint32(31)).astype(np.uint8), ab=a_span[:, 0].copy(), ae=a_span[:, 1].copy(),
                           bb=b_span[:, 0].copy(), be=b_span[:, 1].copy(), block_first=[0, p.n_reads], qv=None)
