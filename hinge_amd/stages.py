"""Python driver of the `hinge filter` stage over the C ABI (used by tests, bench.py and the
multi-GPU orchestrator; the installed command-line tools are the C++ programs in hinge_amd/host/).

Same inputs, same output files, same part loop as src/filter/filter.cpp:474-1111; all arithmetic on
the pile-ups happens in the HIP kernels behind include/hinge_hip.h.  Host work here is file I/O only.
"""
from __future__ import annotations

import os
from typing import List, Optional

import numpy as np

from . import formats
from .capi import Context, FilterParams, HingeError
from .config import IniFile, filter_params


def las_list(las_base: str, mlas: bool) -> List[str]:
    """filter.cpp:228-278 (+ glob() at :35-63)."""
    if mlas:
        out, i = [], 1
        while os.path.exists("%s.%d.las" % (las_base, i)):
            out.append("%s.%d.las" % (las_base, i))
            i += 1
        return out
    return [las_base if las_base.endswith(".las") else las_base + ".las"]


def qv_masks(qv: List[np.ndarray], tspace: int) -> np.ndarray:
    """filter.cpp:309-312,340-369: longest run of good (<40) segments; the last segment always
    breaks a run."""
    out = np.zeros((len(qv), 2), np.int32)
    for i, q in enumerate(qv):
        good = (np.asarray(q) < 40)
        s = e = 0
        mx = maxs = maxe = 0
        n = len(good)
        for j in range(n):
            if good[j] and j < n - 1:
                e += 1
            else:
                if e - s > mx:
                    maxe, maxs, mx = e, s, e - s
                s = e = j + 1
        out[i] = (maxs * tspace, maxe * tspace)
    return out


def self_match_reads(p: formats.Pileups, rlen: np.ndarray) -> set:
    """filter.cpp:552-561 (float32 accumulation in record order)."""
    acc = {}
    for a, sp in zip(p.self_a, p.self_span):
        c = acc.get(int(a), np.float32(0.0))
        c = np.float32(c + np.float32(int(sp[1]) - int(sp[0])))
        c = np.float32(c + np.float32(int(sp[3]) - int(sp[2])))
        acc[int(a)] = c
    out = set()
    for a, c in acc.items():
        cov = np.float32(c / np.float32(rlen[a]))
        if cov > 4.5 and rlen[a] > 10000:
            out.add(a)
    return out


def run_filter(db_name: str, las_base: str, prefix: str, config: str, mlas: bool = False, device: int = 0,
               write_coverage: bool = True, force_exact: bool = False, ctx: Optional[Context] = None, packed: bool = False) -> int:
    """`hinge filter --db DB --las LAS [--mlas] -x PREFIX --config INI`.  Returns the exit code.
    packed = True takes the route of the executables (hinge_set_pileups_packed with the ingest's span copy and facts, the
    coverage bins from K2); False the plain hinge_set_pileups (device sweep k_pileup_facts) and hinge_filter_coverage_bins."""
    try:
        idx = formats.read_db_index(db_name)
    except OSError:
        return 1
    rlen = idx["rlen"]
    n_read = len(rlen)
    qv = formats.read_qual_track(db_name)
    has_qv = qv is not None
    names = las_list(las_base, mlas)
    if not names:
        return 1
    first = formats.read_las(names[0])
    qvm = qv_masks(qv, first.tspace) if has_qv else None
    ini = IniFile(config)
    if ini.error < 0:
        return 1
    P: FilterParams = filter_params(ini, has_qv)

    own_ctx = ctx is None
    if own_ctx:
        ctx = Context(device)
    ctx.set_reads(rlen, qvm)
    ctx.force_exact(force_exact)
    ctx.set_min_cov(P.min_cov)

    f_cov = open(prefix + ".coverage.txt", "w")
    open(prefix + ".homologous.txt", "w").close()
    f_rep = open(prefix + ".repeat.txt", "w")
    open(prefix + ".filtered.fasta", "w").close()
    f_hg = open(prefix + ".hinges.txt", "w")
    f_mask = open(prefix + ".mas", "w")
    f_cmask = open(prefix + ".cmas", "w")
    f_covflag = open(prefix + ".cov.flag", "w")
    f_selfflag = open(prefix + ".self.flag", "w")
    rc = 0
    try:
        for part, name in enumerate(names):
            recs = first if part == 0 else formats.read_las(name)
            if recs.novl == 0:
                rc = 1
                break
            pile = formats.pileups_from_las(recs, rlen)
            r_begin = int(recs.rec["aread"][0])
            r_end = int(recs.rec["aread"][-1])
            if packed:
                from .capi import pack_spans
                span16, max_pile, in_range = pack_spans(pile.row_ptr, pile.a_span, rlen)
                ctx.set_pileups_packed(r_begin, r_end, pile.row_ptr, pile.a_span, pile.b_span, pile.b_flag, span16, max_pile, in_range)
                if P.reso == 40:
                    from .capi import pile_bins
                    ctx.set_pile_bins(pile_bins(pile.row_ptr[r_begin:r_end + 2], pile.a_span, rlen[r_begin:r_end + 1], 40), 40)
            else:
                ctx.set_pileups(r_begin, r_end, pile.row_ptr, pile.a_span, pile.b_span, pile.b_flag)
            ctx.coverage_out(packed and write_coverage)
            if packed:      # the route of the executables: the one-sweep pass (statistics, median, masks, annotations)
                ctx.filter_sweep(P, fetch=True)
            else:
                ctx.filter_stats(P)
                ctx.filter_median(P, r_begin, r_end, fetch=True)
                ctx.filter_mask_annotate(P)
            ctx.filter_hinges(P)
            mask, cmask, flags = ctx.get_masks()
            off, pos, typ, ish = ctx.get_annotations()
            if write_coverage:
                nb, cov = ctx.get_coverage() if packed else ctx.coverage_bins(r_begin, r_end, P.reso, 0)
                o = 0
                for k, i in enumerate(range(r_begin, r_end + 1)):
                    c = cov[o:o + nb[k]]
                    o += nb[k]
                    f_cov.write("read %d " % i + "".join("%d,%d " % (P.reso * j, v) for j, v in enumerate(c)) + "\n")
            selfm = self_match_reads(pile, rlen) if P.delete_telomere else set()
            for k, i in enumerate(range(r_begin, r_end + 1)):
                if P.delete_telomere:
                    if flags[k] & 1:
                        f_covflag.write("%d\n" % i)
                    if i in selfm:
                        f_selfflag.write("%d\n" % i)
                f_cmask.write("%d %d %d\n" % (i, cmask[k, 0], cmask[k, 1]))
                f_mask.write("%d %d %d\n" % (i, mask[k, 0], mask[k, 1]))
            open("debug.txt", "w").close()
            if f_rep is not None:      # closed after the first part, filter.cpp:1086
                for k, i in enumerate(range(r_begin, r_end + 1)):
                    s, e = off[k], off[k + 1]
                    f_rep.write("%d " % i + "".join("%d %d " % (pos[t], typ[t]) for t in range(s, e)) + "\n")
                f_rep.close()
                f_rep = None
            for k, i in enumerate(range(r_begin, r_end)):   # excludes r_end, filter.cpp:1091
                s, e = off[k], off[k + 1]
                f_hg.write("%d " % i + "".join("%d %d " % (pos[t], typ[t]) for t in range(s, e) if ish[t]) + "\n")
    except HingeError as ex:
        rc = 1 if ex.code == -4 else 2
        if rc == 2:
            raise
    finally:
        for f in (f_cov, f_rep, f_hg, f_mask, f_cmask, f_covflag, f_selfflag):
            if f is not None:
                f.close()
        if own_ctx:
            ctx.close()
    return rc


def _read_mas(path: str, n_read: int) -> np.ndarray:
    eff = np.zeros((n_read, 2), np.int64)
    for line in open(path):
        tok = line.split()
        if len(tok) >= 3:
            eff[int(tok[0])] = (int(tok[1]), int(tok[2]))
    return eff


def run_maximal(db_name: str, las_base: str, prefix: str, config: str, mlas: bool = False, device: int = 0) -> int:
    """`hinge maximal --db DB --las LAS [--mlas] -x PREFIX --config INI`: writes PREFIX.max and PREFIX.contained.txt
    (maximal.cpp:517-531, 780-878; the `.coverage.txt` rewrite of :659-685 is left to `hinge filter`'s file).  One process, one
    GPU, part after part: the single-rank case of dist.ShardedMaximal."""
    import torch
    from . import capi
    from .dist import BlockTable, Exchange, HipMaximalBackend, ShardedMaximal
    try:
        rlen = formats.read_db_index(db_name)["rlen"]
    except OSError:
        return 1
    n = len(rlen)
    names = las_list(las_base, mlas)
    ini = IniFile(config)
    if not names or ini.error < 0:
        return 1
    g = ini.get_int
    eff = _read_mas(prefix + ".mas", n)
    ctx = Context(device)
    active = None
    contained_lines, max_lines = [], []
    try:
        for name in names:
            recs = formats.read_las(name)
            if recs.novl == 0:
                return 1
            pile = formats.pileups_from_las(recs, rlen)
            r_begin, r_end = int(recs.rec["aread"][0]), int(recs.rec["aread"][-1])
            tb = 1 if recs.tspace <= formats.TRACE_XOVR else 2
            be = HipMaximalBackend(ctx, rlen, eff, r_begin, r_end, pile.row_ptr, pile.a_span, pile.b_span, pile.b_flag, recs.trace,
                                   recs.trace_off[:-1][pile.las_index], recs.rec["tlen"][pile.las_index], tb,
                                   g("filter", "length_threshold", -1), g("filter", "aln_threshold", -1), g("filter", "theta", -1), g("filter", "theta2", 0),
                                   bool(g("layout", "use_two_matches", 1)), torch.device("cuda", device), self_before=pile.self_before)   # (maximal.cpp:471 reads it from [layout])
            if active is not None:      # the sequential --mlas loop: a read removed by an earlier part stays removed
                be.active0 = (be.active0.astype(bool) & active.astype(bool)).astype(np.uint8)
            job = ShardedMaximal(be, Exchange(BlockTable([0, n]), torch.device("cuda", device)))
            active = job.step()
            contained_lines += ["%d\t%d" % (i, c) for i, c in enumerate(job.containing) if c >= 0]
            max_lines += ["%d" % i for i in range(r_begin, r_end + 1) if active[i]]
    finally:
        ctx.close()
    open(prefix + ".contained.txt", "w").write("".join(l + "\n" for l in contained_lines))
    open(prefix + ".max", "w").write("".join(l + "\n" for l in max_lines))
    return 0


def run_layout(db_name: str, las_name: str, prefix: str, out: str, config: str, device: int = 0) -> int:
    """`hinge layout --db DB --las LAS -x PREFIX --config INI -o OUT` for one merged .las: writes OUT.edges.hinges, .edges.hinges2,
    .edges.skipped, .deadends.txt, .hinge.list, .killed.hinges, .hgraph and PREFIX.garbage.txt (hinging.cpp:616-2148; the debug dumps
    and the pure-greedy .edges.greedy / .1 / .2 of :1724-1860 are the C++ executable's).  The single-rank case of dist.ShardedLayout."""
    import torch
    from . import layout as L
    from .dist import BlockTable, Exchange, HipLayoutBackend, ShardedLayout
    try:
        rlen = formats.read_db_index(db_name)["rlen"]
    except OSError:
        return 1
    n = len(rlen)
    ini = IniFile(config)
    if ini.error < 0:
        return 1
    P = L.LayoutParams.from_ini(ini)
    eff = _read_mas(prefix + ".mas", n)
    maximal = np.zeros(n, bool)
    for line in open(prefix + ".max"):
        if line.strip():
            maximal[int(line)] = True
    repeats = L.read_pairs_file(prefix + ".repeat.txt", n)
    hinges = L.read_pairs_file(prefix + ".hinges.txt", n)
    recs = formats.read_las(las_name if las_name.endswith(".las") else las_name + ".las")
    if recs.novl == 0:
        return 1
    pile = formats.pileups_from_las(recs, rlen)
    tb = 1 if recs.tspace <= formats.TRACE_XOVR else 2
    ctx = Context(device)
    try:
        be = HipLayoutBackend(ctx, rlen, eff, pile, recs.trace, recs.trace_off[:-1][pile.las_index], recs.rec["tlen"][pile.las_index], tb)
        files = ShardedLayout(be, Exchange(BlockTable([0, n]), torch.device("cuda", device)), P, eff, maximal, repeats, hinges).step()
    finally:
        ctx.close()
    for suffix, lines in files.items():
        base = prefix if suffix in (".garbage.txt", ".killed.hinges") else out      # hinging.cpp:659, 1201 write these two under --prefix
        open(base + suffix, "w").write("".join(l + "\n" for l in lines))
    return 0
