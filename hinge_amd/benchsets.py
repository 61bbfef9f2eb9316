"""The read sets bench.py passes over, and the digest its results are checked with.

N = 1: part p of the GPU is BASELINE config 2 restated with generator seed `base.seed + 17 p` (part 0 = the configuration's own
data set), every B id inside the part.

N >= 2 (weak scaling): the ranks work in TEAMS of two.  Team q generates, for part p, ONE data set of twice config 2's genome
(2 x 4.6 Mb at 160x, twice the repeat families, DBsplit into 2 blocks); rank 2 q + k holds block k: the same 86-87 k reads and
~26 M overlaps per GPU as at N = 1, but about half of every pile-up's B reads now live in the OTHER rank's block, as they do
in any DBsplit data set.  Hinge calling reads `maskvec[B]` (filter.cpp:883-890), so what a rank computes depends on the masks
its team mate computed: exchange 2 is consumed, and exchange 1 (the global median over all N blocks -> MIN_COV) as well.
Teams, not one N-block genome, keep the set-up cost per rank and the CPU oracle's memory independent of N.

Expected results: tests/golden/bench_expect.json (tools/make_bench_expect.py, CPU oracle): per (N, part, rank) the number of
hinges and `digest()` of the (read index inside the block, position, type) rows.
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass

import numpy as np

from . import synth

TEAM = 2
DIGEST_MOD = (1 << 61) - 1


def supported_world(world: int, scaling: str = "weak") -> bool:
    return world == 1 or (world > 1 and (scaling == "strong" or world % TEAM == 0))


def part_spec(base: synth.SynthSpec, world: int, rank: int, part: int, scaling: str = "weak"):
    """(generator spec, block of it this rank holds).  scaling = "strong": ONE config-2 data set per part - the very reads and
    overlaps of the N = 1 run - DBsplit into `world` blocks, rank r holds block r (merged-las semantics: the union of the ranks'
    results is the N = 1 result, which is what bench.py asserts)."""
    if world == 1:
        return dataclasses.replace(base, n_blocks=1, seed=base.seed + 17 * part), 0
    if scaling == "strong":
        return dataclasses.replace(base, n_blocks=world, seed=base.seed + 17 * part), rank
    assert supported_world(world), "bench data sets are defined for 1 GPU or an even number of GPUs"
    q = rank // TEAM
    spec = dataclasses.replace(base, genome_len=TEAM * base.genome_len, n_blocks=TEAM, n_repeat_families=TEAM * base.n_repeat_families,
                               seed=base.seed + 1000 * (q + 1) + 17 * part)
    return spec, rank % TEAM


@dataclass
class RankPart:
    """One rank's block of one part, B reads named as (rank that holds the B read's block, index inside that block)."""
    rlen: np.ndarray          # int32 [n_reads] of the own block
    row_ptr: np.ndarray       # int64 [n_reads + 1], 0-based over the own block's overlaps (self-overlaps dropped)
    a_span: np.ndarray        # int32 [n, 2]
    b_span: np.ndarray        # int32 [n, 2] (B's forward strand)
    b_owner: np.ndarray       # int32 [n]
    b_local: np.ndarray       # int32 [n]
    comp: np.ndarray          # uint32 [n], 0 / 1
    last_a: int               # A read (index inside the block) of the block's last .las record
    n_records: int            # .las records of the block, self-overlaps included

    @property
    def n_reads(self) -> int:
        return int(self.rlen.shape[0])

    @property
    def n_ovl(self) -> int:
        return int(self.comp.shape[0])


def rank_part(base: synth.SynthSpec, world: int, rank: int, part: int, data: synth.SynthData = None, scaling: str = "weak") -> RankPart:
    spec, k = part_spec(base, world, rank, part, scaling)
    d = synth.generate(spec) if data is None else data
    pile = synth.to_pileups(d)
    bf = np.asarray(d.block_first, dtype=np.int64)
    lo, hi = int(bf[k]), int(bf[k + 1])
    s, e = int(pile.row_ptr[lo]), int(pile.row_ptr[hi])
    b = (pile.b_flag[s:e] & np.uint32(0x7FFFFFFF)).astype(np.int64)
    kb = np.searchsorted(bf, b, side="right") - 1                     # block of every B read
    first_rank = (rank // TEAM) * TEAM if (world > 1 and scaling != "strong") else 0
    r0, r1 = np.searchsorted(d.aread, [lo, hi])
    return RankPart(rlen=np.ascontiguousarray(d.rlen[lo:hi], dtype=np.int32),
                    row_ptr=(pile.row_ptr[lo:hi + 1] - s).astype(np.int64),
                    a_span=np.ascontiguousarray(pile.a_span[s:e]), b_span=np.ascontiguousarray(pile.b_span[s:e]),
                    b_owner=(first_rank + kb).astype(np.int32), b_local=(b - bf[kb]).astype(np.int32),
                    comp=(pile.b_flag[s:e] >> np.uint32(31)).astype(np.uint32),
                    last_a=int(d.aread[r1 - 1]) - lo, n_records=int(r1 - r0))


def digest(rows) -> int:
    """Order-independent digest of (read index inside the block, position, type) int rows (a few thousand per part: exact
    Python integers)."""
    rows = np.asarray(rows, dtype=np.int64).reshape(-1, 3)
    total = 0
    for r, pos, typ in rows.tolist():
        h = ((r * 1_000_003 + pos) * 7 + (typ + 3)) * 2_654_435_761 % DIGEST_MOD
        total += (h * h) % 1_000_000_007 + h                           # not linear in the fields
    return total % DIGEST_MOD
