"""bench.py's read sets (hinge_amd/benchsets.py), CPU side: the strong-scaling split of ONE data set into N blocks is the N = 1
data set - same reads, same overlaps - with every B read named by (owner rank, index inside the owner's block)."""
import dataclasses

import numpy as np
import pytest

from hinge_amd import benchsets, synth

BASE = dataclasses.replace(synth.CONFIGS["cfg2_ecoli160"], genome_len=120_000, coverage=30, n_repeat_families=2)


@pytest.mark.parametrize("world", [2, 3, 4])
def test_strong_split_is_the_single_block_data_set(world):
    one = benchsets.rank_part(BASE, 1, 0, 1)
    parts = [benchsets.rank_part(BASE, world, r, 1, scaling="strong") for r in range(world)]
    first = np.concatenate([[0], np.cumsum([p.n_reads for p in parts])])
    assert first[-1] == one.n_reads and sum(p.n_ovl for p in parts) == one.n_ovl and sum(p.n_records for p in parts) == one.n_records
    assert np.array_equal(np.concatenate([p.rlen for p in parts]), one.rlen)
    assert np.array_equal(np.concatenate([p.a_span for p in parts]), one.a_span)
    assert np.array_equal(np.concatenate([p.b_span for p in parts]), one.b_span)
    assert np.array_equal(np.concatenate([p.comp for p in parts]), one.comp)
    # B reads: (owner, local) back to the data set's own ids
    b_global = np.concatenate([first[p.b_owner] + p.b_local for p in parts])
    assert np.array_equal(b_global, one.b_local)                    # (world 1: owner 0, local = global)
    for r, p in enumerate(parts):
        assert p.b_owner.min() >= 0 and p.b_owner.max() < world
        assert np.all(p.b_local < np.asarray([q.n_reads for q in parts])[p.b_owner])
        assert np.array_equal(p.row_ptr, one.row_ptr[first[r]:first[r + 1] + 1] - one.row_ptr[first[r]])
    assert any((p.b_owner != r).any() for r, p in enumerate(parts)), "no pile-up crosses a block: the exchanges would not be consumed"


def test_weak_worlds_and_strong_worlds():
    assert benchsets.supported_world(1) and benchsets.supported_world(2) and not benchsets.supported_world(3)
    assert benchsets.supported_world(3, "strong") and benchsets.supported_world(8, "strong")
    spec, k = benchsets.part_spec(BASE, 4, 2, 3, "strong")
    assert (spec.n_blocks, k, spec.seed, spec.genome_len) == (4, 2, BASE.seed + 51, BASE.genome_len)
    spec, k = benchsets.part_spec(BASE, 4, 2, 3)
    assert (spec.n_blocks, k, spec.genome_len) == (2, 0, 2 * BASE.genome_len)
