// Workgroup-parallel, exact replay of libstdc++'s std::sort (introsort, see stdsort_emul.h) for one
// list held in LDS.  Used twice per hinge decision that depends on tie order:
//   * std::sort(idx_pileup[i].begin(), end(), compare_overlap)      src/filter/filter.cpp:565-567
//   * std::sort(read_other_ends.begin(), end(), pairAscend/Descend) src/filter/filter.cpp:914,1010
//
// Why this exists: both comparators look at one field only, std::sort is not stable, and the order of
// equal-key elements feeds the hinge scan.  A lane-serial replay costs ~n log n dependent LDS round
// trips; this version keeps libstdc++'s exact sequence of partitions but spreads the work:
//
//   * The introsort recursion tree is walked level by level.  The segments of one level are disjoint,
//     so the four wavefronts of the workgroup partition different segments at the same time (the order
//     in which disjoint segments are processed cannot change the result).
//   * One partition is done by a whole wavefront.  __unguarded_partition(first, last, pivot) walks
//     `first` up to the next element with !comp(x, pivot) ("left stopper") and `last` down to the next
//     element with !comp(pivot, x) ("right stopper"), swaps them and repeats while first < last.
//     Stoppers are a property of the ORIGINAL segment content (each pointer sees a position at most once
//     before the pointers cross, and a swapped-in element is itself a stopper for the other pointer), so
//     with LS[k] / RS[k] = k-th left / right stopper position:
//         swap k happens        <=>  LS[k] < RS[k]         (a monotone prefix k = 1..t)
//         returned cut          =    min(LS[t+1], RS[t])
//     i.e. one ballot/prefix sweep over the segment, a count over the stopper lists, a parallel swap pass.
//   * The final insertion sort (threshold 16) is a stable sort; after the introsort loop every leaf
//     segment only holds keys that do not precede those of the segments before it, so it reduces to a
//     stable sort inside each leaf (a rank count over <= 16 neighbours per element).
//   * The depth-limit heapsort fallback is replayed serially by one lane (never seen on real pile-ups).
//
// The sort is DESCENDING on key: comp(x, y) = key[x] > key[y].  (Ascending sorts pass negated keys;
// the comparison outcomes are identical.)
#pragma once
#include <hip/hip_runtime.h>
#include "stdsort_emul.h"

namespace hinge {

constexpr int PO_CAP = 4096;            // longest list sorted in LDS
constexpr int PO_CAP_SMALL = 2048;      // the half-size instance: two workgroups of k_hinge_call per CU

template <int CAP>
struct WaveSortLdsT {
    static constexpr int SEG_CAP = CAP / 8;   // segments alive on one level (each is > 16 long)
    int key[CAP];                  // sort key by element index (never permuted)
    int perm[CAP];                 // perm[p] = element at position p
    unsigned short pl[CAP];        // left stopper positions by rank (scratch); on return: position of every element
    unsigned short pr[CAP];        // right stopper positions by rank from the left (scratch)
    unsigned short seglo[CAP];     // leaf segment of every position
    unsigned short seghi[CAP];
    unsigned short seg_first[2][SEG_CAP], seg_last[2][SEG_CAP];
    unsigned char seg_depth[2][SEG_CAP];
    int seg_cnt[2];
};
typedef WaveSortLdsT<PO_CAP> WaveSortLds;

template <typename WS>
__device__ __forceinline__ void mark_leaf(WS& o, int first, int last, int lane) {
    for (int p = first + lane; p < last; p += 64) { o.seglo[p] = (unsigned short)first; o.seghi[p] = (unsigned short)last; }
}

// Call from ALL threads of a 256-thread workgroup (contains __syncthreads).  On return
// o.pl[e] = position of element e in std::sort(order, comp) of the list 0..n-1.
template <typename WS>
__device__ inline void block_std_sort_desc(WS& o, int n, int tid) {
    const int lane = tid & 63;
    const int wib = tid >> 6;
    const unsigned long long lmask = (1ull << lane) - 1ull;
    for (int p = tid; p < n; p += 256) o.perm[p] = p;
    if (tid == 0) {
        o.seg_cnt[0] = 0;
        o.seg_cnt[1] = 0;
        if (n > 16) {
            o.seg_first[0][0] = 0;
            o.seg_last[0][0] = (unsigned short)n;
            o.seg_depth[0][0] = (unsigned char)(hinge_sort::floor_log2((unsigned)n) * 2);
            o.seg_cnt[0] = 1;
        }
    }
    if (n <= 16 && wib == 0) mark_leaf(o, 0, n, lane);
    __syncthreads();
    hinge_sort::KeyCmp cmp{o.key, 1};
    int cur = 0;
    while (true) {
        const int nc = o.seg_cnt[cur];
        if (nc == 0) break;
        const int nxt = cur ^ 1;
        for (int j = wib; j < nc; j += 4) {
            const int first = o.seg_first[cur][j], last = o.seg_last[cur][j];
            int depth = o.seg_depth[cur][j];
            if (depth == 0) {   // __partial_sort(first, last, last): heap sort, serial
                if (lane == 0) hinge_sort::heapsort_(o.perm, first, last, cmp);
                for (int p = first + lane; p < last; p += 64) { o.seglo[p] = (unsigned short)p; o.seghi[p] = (unsigned short)(p + 1); }
                continue;
            }
            --depth;
            // __move_median_to_first(first, first+1, mid, last-1): lanes 0..2 fetch the three candidates
            const int mid = first + (last - first) / 2;
            const int cpos = lane == 0 ? first + 1 : (lane == 1 ? mid : last - 1);
            int cel = 0, ckey = 0;
            if (lane < 3) { cel = o.perm[cpos]; ckey = o.key[cel]; }
            const int ka = __builtin_amdgcn_readlane(ckey, 0), kb = __builtin_amdgcn_readlane(ckey, 1), kc = __builtin_amdgcn_readlane(ckey, 2);
            int msel;   // which candidate goes to `first`: 0 = a (first+1), 1 = b (mid), 2 = c (last-1); comp(x,y) = x > y
            if (ka > kb) { if (kb > kc) msel = 1; else if (ka > kc) msel = 2; else msel = 0; }
            else if (ka > kc) msel = 0;
            else if (kb > kc) msel = 2;
            else msel = 1;
            const int mpos = msel == 0 ? first + 1 : (msel == 1 ? mid : last - 1);
            const int mel = __builtin_amdgcn_readlane(cel, msel);
            const int pivot = msel == 0 ? ka : (msel == 1 ? kb : kc);
            if (lane == 0) { const int fe = o.perm[first]; o.perm[first] = mel; o.perm[mpos] = fe; }
            const int lo = first + 1, hi = last;
            // one sweep: stopper positions by rank from the left, stored at [lo + rank)
            int totalL = 0, totalR = 0;
            for (int base = lo; base < hi; base += 64) {
                const int p = base + lane;
                int x = 0;
                if (p < hi) x = o.key[o.perm[p]];
                const bool isL = (p < hi) && (x <= pivot);   // !comp(x, pivot)
                const bool isR = (p < hi) && (x >= pivot);   // !comp(pivot, x)
                const unsigned long long balL = __ballot(isL), balR = __ballot(isR);
                if (isL) o.pl[lo + totalL + __popcll(balL & lmask)] = (unsigned short)p;
                if (isR) o.pr[lo + totalR + __popcll(balR & lmask)] = (unsigned short)p;
                totalL += __popcll(balL);
                totalR += __popcll(balR);
            }
            // LS[k] = pl[lo+k-1], RS[k] = pr[lo+totalR-k]; swaps are the prefix of k with LS[k] < RS[k]
            const int kmax = min(totalL, totalR);
            int t = 0;
            for (int base = 0; base < kmax; base += 64) {
                const int k = base + lane;   // 0-based
                const bool sw = (k < kmax) && (o.pl[lo + k] < o.pr[lo + totalR - 1 - k]);
                const unsigned long long b = __ballot(sw);
                t += __popcll(b);
                if (b != ~0ull) break;       // monotone: the first false ends the prefix
            }
            int cut = 0x7fffffff;
            if (t < totalL) cut = o.pl[lo + t];
            if (t >= 1) cut = min(cut, (int)o.pr[lo + totalR - t]);
            for (int k = lane; k < t; k += 64) {
                const int a = o.pl[lo + k], b = o.pr[lo + totalR - 1 - k];
                const int tmp = o.perm[a]; o.perm[a] = o.perm[b]; o.perm[b] = tmp;
            }
            // children: [first, cut) (the loop's continuation) and [cut, last) (the recursive call)
            const bool big_l = cut - first > 16, big_r = last - cut > 16;
            if (!big_l) mark_leaf(o, first, cut, lane);
            if (!big_r) mark_leaf(o, cut, last, lane);
            if (lane == 0 && (big_l || big_r)) {
                int slot = atomicAdd(&o.seg_cnt[nxt], (int)big_l + (int)big_r);
                if (big_l) { o.seg_first[nxt][slot] = (unsigned short)first; o.seg_last[nxt][slot] = (unsigned short)cut; o.seg_depth[nxt][slot] = (unsigned char)depth; ++slot; }
                if (big_r) { o.seg_first[nxt][slot] = (unsigned short)cut; o.seg_last[nxt][slot] = (unsigned short)last; o.seg_depth[nxt][slot] = (unsigned char)depth; }
            }
        }
        __syncthreads();
        if (tid == 0) o.seg_cnt[cur] = 0;
        cur = nxt;
        __syncthreads();
    }
    // final insertion sort == stable sort inside each leaf; positions by element -> pr -> pl
    for (int p = tid; p < n; p += 256) {
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
    __syncthreads();
    for (int p = tid; p < n; p += 256) o.pl[p] = o.pr[p];
    __syncthreads();
}

}  // namespace hinge
