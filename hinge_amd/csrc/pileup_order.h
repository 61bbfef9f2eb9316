// Wavefront-parallel, exact replay of libstdc++'s std::sort (introsort, see stdsort_emul.h) for one
// list held in LDS.  Used twice per hinge decision:
//   * std::sort(idx_pileup[i].begin(), end(), compare_overlap)      src/filter/filter.cpp:565-567
//   * std::sort(read_other_ends.begin(), end(), pairAscend/Descend) src/filter/filter.cpp:914,1010
//
// Why this exists: both comparators look at one field only, std::sort is not stable, and the order of
// equal-key elements feeds the hinge scan.  A lane-serial replay costs ~n log n dependent LDS round
// trips; this version keeps libstdc++'s exact sequence of partitions but executes each partition with
// the whole wavefront:
//
//   __unguarded_partition(first, last, pivot) walks `first` up to the next element with !comp(x, pivot)
//   ("left stopper") and `last` down to the next element with !comp(pivot, x) ("right stopper"), swaps
//   them and repeats while first < last.  Stoppers are a property of the ORIGINAL segment content (each
//   pointer sees a position at most once before the pointers cross, and a swapped-in element is itself
//   a stopper for the other pointer), so with LS[k] / RS[k] = k-th left / right stopper position:
//       swap k happens        <=>  LS[k] < RS[k]  <=>  #right stoppers right of LS[k]  >=  k
//       returned cut          =    min(LS[t+1], RS[t])         (t = number of swaps)
//   which is two ballot/prefix sweeps over the segment and one parallel swap pass.
//
//   The final insertion sort (threshold 16) is a stable sort; after the introsort loop every leaf
//   segment (<= 16 elements, or a heap-sorted one) only holds keys that do not precede those of the
//   segments before it, so it reduces to a stable sort inside each leaf.
//
// The sort is DESCENDING on key: comp(x, y) = key[x] > key[y].  (Ascending sorts pass negated keys;
// the comparison outcomes are identical.)
#pragma once
#include <hip/hip_runtime.h>
#include "stdsort_emul.h"

namespace hinge {

constexpr int PO_CAP = 4096;   // longest list sorted in LDS

struct WaveSortLds {
    int key[PO_CAP];               // sort key by element index (never permuted)
    int perm[PO_CAP];              // perm[p] = element at position p
    unsigned short pl[PO_CAP];     // left stopper positions by rank (scratch); on return: position of every element
    unsigned short pr[PO_CAP];     // right stopper positions by rank (scratch)
    unsigned short seglo[PO_CAP];  // leaf segment of every position
    unsigned short seghi[PO_CAP];
    int stk_first[64], stk_last[64], stk_depth[64];
};

// Runs on ONE wavefront (all 64 lanes active).  On return o.pl[e] = position of element e in
// std::sort(order, comp) of the list 0..n-1.
__device__ inline void wave_std_sort_desc(WaveSortLds& o, int n, int lane) {
    for (int p = lane; p < n; p += 64) o.perm[p] = p;
    if (n <= 0) return;
    hinge_sort::KeyCmp cmp{o.key, 1};
    int sp = 0;
    if (lane == 0) { o.stk_first[0] = 0; o.stk_last[0] = n; o.stk_depth[0] = hinge_sort::floor_log2((unsigned)n) * 2; }
    sp = 1;
    while (sp > 0) {
        --sp;
        int first = o.stk_first[sp], last = o.stk_last[sp], depth = o.stk_depth[sp];
        bool leaf_done = false;
        while (last - first > 16) {
            if (depth == 0) {
                if (lane == 0) hinge_sort::heapsort_(o.perm, first, last, cmp);
                for (int p = first + lane; p < last; p += 64) { o.seglo[p] = (unsigned short)p; o.seghi[p] = (unsigned short)(p + 1); }
                leaf_done = true;
                break;
            }
            --depth;
            const int mid = first + (last - first) / 2;
            if (lane == 0) hinge_sort::move_median_to_first_(o.perm, first, first + 1, mid, last - 1, cmp);
            const int pivot = o.key[o.perm[first]];
            const int lo = first + 1, hi = last;
            // sweep 1: number of right stoppers (key >= pivot)
            int totalR = 0;
            for (int base = lo; base < hi; base += 64) {
                const int p = base + lane;
                const bool isR = (p < hi) && (o.key[o.perm[p]] >= pivot);
                totalR += __popcll(__ballot(isR));
            }
            // sweep 2: stopper ranks, positions by rank, number of swaps
            int prefL = 0, prefR = 0, t = 0;
            const unsigned long long lmask = (1ull << lane) - 1ull;
            for (int base = lo; base < hi; base += 64) {
                const int p = base + lane;
                int x = 0;
                if (p < hi) x = o.key[o.perm[p]];
                const bool isL = (p < hi) && (x <= pivot);   // !comp(x, pivot)
                const bool isR = (p < hi) && (x >= pivot);   // !comp(pivot, x)
                const unsigned long long balL = __ballot(isL), balR = __ballot(isR);
                const int rankL = prefL + __popcll(balL & lmask) + 1;
                const int rIncl = prefR + __popcll(balR & lmask) + (isR ? 1 : 0);
                bool swp = false;
                if (isL) {
                    o.pl[rankL - 1] = (unsigned short)p;
                    swp = (totalR - rIncl) >= rankL;
                }
                if (isR) o.pr[totalR - rIncl] = (unsigned short)p;   // rank from the right, 0-based
                t += __popcll(__ballot(swp));
                prefL += __popcll(balL);
                prefR += __popcll(balR);
            }
            const int totalL = prefL;
            int cut = 0x7fffffff;
            if (t < totalL) cut = o.pl[t];
            if (t >= 1) cut = min(cut, (int)o.pr[t - 1]);
            for (int k = lane; k < t; k += 64) {
                const int a = o.pl[k], b = o.pr[k];
                const int tmp = o.perm[a]; o.perm[a] = o.perm[b]; o.perm[b] = tmp;
            }
            // recurse right (deferred on the stack), loop left
            if (lane == 0) { o.stk_first[sp] = cut; o.stk_last[sp] = last; o.stk_depth[sp] = depth; }
            ++sp;
            last = cut;
        }
        if (!leaf_done)
            for (int p = first + lane; p < last; p += 64) { o.seglo[p] = (unsigned short)first; o.seghi[p] = (unsigned short)last; }
    }
    // final insertion sort == stable sort inside each leaf; positions by element go to pr, then pl
    for (int base = 0; base < n; base += 64) {
        const int p = base + lane;
        if (p < n) {
            const int slo = o.seglo[p], shi = o.seghi[p];
            const int el = o.perm[p];
            const int x = o.key[el];
            int r = slo;
            for (int q = slo; q < shi; ++q) {
                const int y = o.key[o.perm[q]];
                r += (y > x) || (y == x && q < p);
            }
            o.pr[el] = (unsigned short)r;
        }
    }
    for (int p = lane; p < n; p += 64) o.pl[p] = o.pr[p];
}

}  // namespace hinge
