// (A, B) grouping of one read's overlap records the way maximal.cpp:615-654,789-849 and
// hinging.cpp:478-602 walk them.
//
// The reference keeps `std::unordered_map<int, std::vector<LOverlap*>> idx_ab[A]`, fills it in record
// order and ITERATES it; for every B it std::sort()s the pair's overlaps by compare_overlap and takes the
// first (and second, if use_two_matches) for ProcessAlignment.  The iteration order of the unordered_map
// and the tie order of std::sort both reach the outputs (the order matches_forward is
// filled in, the "last containing read" of .contained.txt), so the host uses the very same containers
// of the same libstdc++ here.  Only the selected overlaps (at most two per pair) go to the GPU.
#pragma once
#include <algorithm>
#include <memory_resource>
#include <unordered_map>
#include <vector>
#include "host_common.h"

namespace hh {

struct PairPick {
    int b;            // B read
    int64_t pick[2];  // kept-overlap index of the best / second best (-1: none, or a self-overlap)
};

// Records rec_row_ptr[a] .. rec_row_ptr[a+1] of read a -> picks in unordered_map iteration order.
// `accept(b)` filters B reads before insertion (layout keeps active x active only); n_sorts = how many
// times the reference sorts a pair's vector before reading [0] and [1] (2 in maximal, 1 in layout).
template <typename Accept>
inline void pick_pairs(const LasPart& las, int a, bool two_matches, int n_sorts, Accept accept, std::vector<PairPick>& out) {
    out.clear();
    const int64_t r0 = las.rec_row_ptr[(size_t)a], r1 = las.rec_row_ptr[(size_t)a + 1];
    if (r0 == r1) return;
    // Same container, hash, rehash policy and insertion sequence as the reference (so the same iteration order); only the
    // memory comes from a per-thread bump arena that is released once per read instead of ~4 malloc/free per record.
    // (the buffer is owned by the thread: release() rewinds to it without touching the upstream allocator)
    thread_local std::vector<char> arena_buf(1 << 20);
    thread_local std::pmr::monotonic_buffer_resource arena(arena_buf.data(), arena_buf.size());
    struct Release { std::pmr::monotonic_buffer_resource& a; ~Release() { a.release(); } };
    Release release_after_groups{arena};   // declared BEFORE `groups`: destroyed after it
    std::pmr::unordered_map<int, std::pmr::vector<int64_t>> groups(&arena);   // B -> record indices, in record order
    for (int64_t j = r0; j < r1; j++) {
        const int b = las.rec_b[(size_t)j];
        if (!accept(b)) continue;
        groups[b] = std::pmr::vector<int64_t>(&arena);      // same insertion sequence as the reference's first loop
    }
    for (int64_t j = r0; j < r1; j++) {
        const int b = las.rec_b[(size_t)j];
        if (!accept(b)) continue;
        groups[b].push_back(j);
    }
    auto length_of = [&](int64_t j) -> long long {   // compare_overlap key of a non-self record
        const int64_t k = las.rec_kept[(size_t)j];
        return (long long)(las.a_span[(size_t)k * 2 + 1] - las.a_span[(size_t)k * 2]) + (las.b_span[(size_t)k * 2 + 1] - las.b_span[(size_t)k * 2]);
    };
    for (auto it = groups.begin(); it != groups.end(); ++it) {
        std::pmr::vector<int64_t>& v = it->second;
        if (it->first == a) {   // the (A, A) pair: its overlaps are inactive, ProcessAlignment makes them NOT_ACTIVE and nothing
                                // reads them again; the key itself had to be in the map for the iteration order of the others
            PairPick p{it->first, {-1, -1}};
            out.push_back(p);
            continue;
        }
        if (v.size() > 1)
            for (int s = 0; s < n_sorts; s++)
                std::sort(v.begin(), v.end(), [&](int64_t x, int64_t y) { return length_of(x) > length_of(y); });   // compare_overlap
        PairPick p;
        p.b = it->first;
        p.pick[0] = v.size() > 0 ? las.rec_kept[(size_t)v[0]] : -1;
        p.pick[1] = (v.size() > 1 && two_matches) ? las.rec_kept[(size_t)v[1]] : -1;
        out.push_back(p);
    }
}

// result row of hinge_trim_classify
struct Classified {
    int32_t eff_ab, eff_ae, eff_bb, eff_be, type, active, weight, length, start_idx, end_idx;
};
enum { MT_FORWARD = 0, MT_BACKWARD = 1, MT_ACOVERB = 2, MT_BCOVERA = 3, MT_UNDEFINED = 4, MT_INTERNAL = 5, MT_NOT_ACTIVE = 6,
       MT_FORWARD_INTERNAL = 12, MT_BACKWARD_INTERNAL = 13 };   // src/include/LAInterface.h:30-33

}  // namespace hh
