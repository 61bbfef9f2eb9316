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
constexpr int PO_CAP_MINI = 1024;       // the quarter-size instance behind k_hinge_call_light: four per CU

// LEAN (round 6): no key array - the keys come from a functor (block_std_sort_desc_keys) and live inside the packed words only
template <int CAP, bool LEAN = false>
struct WaveSortLdsT {
    static constexpr int SEG_CAP = CAP / 8;   // segments alive on one level (each is > 16 long)
    int key[LEAN ? 1 : CAP];       // sort key by element index (never permuted)
    int perm[CAP];                 // perm[p] = element at position p
    unsigned short pl[CAP];        // left stopper positions by rank (scratch); on return: position of every element
    unsigned short pr[CAP];        // right stopper positions by rank from the left (scratch)
    unsigned short seglo[CAP];     // leaf segment of every position
    unsigned short seghi[CAP];
    unsigned short seg_first[2][SEG_CAP], seg_last[2][SEG_CAP];
    unsigned char seg_depth[2][SEG_CAP];
    int seg_cnt[2];
    // round 5: segments of 17 .. 64 elements leave the level-by-level walk and are finished by ONE wavefront in registers
    unsigned short small_first[SEG_CAP], small_last[SEG_CAP];
    unsigned char small_depth[SEG_CAP];
    int small_cnt;
    int krange[2][4];              // per wavefront: smallest / largest key (the packed form needs their span)
};
typedef WaveSortLdsT<PO_CAP> WaveSortLds;

template <typename WS>
__device__ __forceinline__ void mark_leaf(WS& o, int first, int last, int lane) {
    for (int p = first + lane; p < last; p += 64) { o.seglo[p] = (unsigned short)first; o.seghi[p] = (unsigned short)last; }
}

// position of the (k + 1)-th set bit of m counted from bit 0 (k < popcount(m))
__device__ __forceinline__ int nth_bit_up(unsigned long long m, int k) {
    int pos = 0;
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) {
        const int c = __popcll((m >> pos) & ((1ull << w) - 1ull));
        if (k >= c) { k -= c; pos += w; }
    }
    return pos;
}
__device__ __forceinline__ int nth_bit_down(unsigned long long m, int k) { return 63 - nth_bit_up(__brevll(m), k); }   // ... counted from bit 63

// The rest of libstdc++'s introsort loop for ONE segment [F, Lst) of 17 .. 64 elements, by one wavefront, in registers: lane l holds
// position F + l (element id + key).  All sub-segments that still need a partition are partitioned IN THE SAME instruction stream -
// they are disjoint lane ranges, every lane works with the ballot bits of its own range - so the loop runs once per recursion
// LEVEL of the segment (two or three times), not once per partition, and touches LDS only to load and to store.
//   median of three   three shuffles from per-lane source lanes (first + 1, mid, last - 1)
//   stoppers          key <= pivot / key >= pivot as ballots; the k-th swap pairs the k-th left stopper from below with the k-th right
//                     stopper from above and happens iff LS[k] < RS[k]: for a left stopper of rank k that is "more than k right
//                     stoppers above me", for a right stopper (rank k from above) "more than k left stoppers below me" - two
//                     popcounts; the partner's lane is the k-th set bit of the other ballot (nth_bit_*), the exchange one shuffle
//   cut               min(LS[t], RS[t - 1]), t = number of swaps, from the same ballots
// A sub-segment whose depth budget is spent takes libstdc++'s heap sort, serially (never seen on pile-ups), as in the walk above.
template <typename WS>
__device__ inline void wave_small_sort_wide(WS& o, int F, int Lst, int depth0, int lane, const hinge_sort::KeyCmp& cmp) {
    const int m = Lst - F;
    int el = 0, key = 0;
    if (lane < m) { el = o.perm[F + lane]; key = o.key[el]; }
    int sf = lane < m ? 0 : lane, sl = lane < m ? m : lane;      // this lane's sub-segment, in lanes
    int dep = depth0;
    const unsigned long long below = (1ull << lane) - 1ull, above = lane == 63 ? 0ull : ~((2ull << lane) - 1ull);
    while (true) {
        bool act = sl - sf > 16;
        if (!__any(act)) break;
        if (__any(act && dep == 0)) {
            // __partial_sort(first, last, last) of the exhausted sub-segments: through LDS, one lane
            if (lane < m) o.perm[F + lane] = el;
            __threadfence_block();
            unsigned long long heads = __ballot(act && dep == 0 && lane == sf);
            while (heads) {
                const int h = (int)__builtin_ctzll(heads);
                const int f = __shfl(sf, h), l = __shfl(sl, h);
                if (lane == 0) hinge_sort::heapsort_(o.perm, F + f, F + l, cmp);
                heads &= heads - 1ull;
            }
            __threadfence_block();
            if (act && dep == 0) { el = o.perm[F + lane]; key = o.key[el]; sf = lane; sl = lane + 1; act = false; }
            if (!__any(act)) break;
        }
        if (act) dep -= 1;
        // __move_median_to_first(first, first + 1, mid, last - 1)
        const int mid = sf + (sl - sf) / 2;
        const int ka = __shfl(key, act ? sf + 1 : lane), kb = __shfl(key, act ? mid : lane), kc = __shfl(key, act ? sl - 1 : lane);
        int msel;   // comp(x, y) = x > y
        if (ka > kb) { if (kb > kc) msel = 1; else if (ka > kc) msel = 2; else msel = 0; }
        else if (ka > kc) msel = 0;
        else if (kb > kc) msel = 2;
        else msel = 1;
        const int mpos = msel == 0 ? sf + 1 : (msel == 1 ? mid : sl - 1);
        const int pivot = msel == 0 ? ka : (msel == 1 ? kb : kc);
        {
            int src = lane;
            if (act) { if (lane == sf) src = mpos; else if (lane == mpos) src = sf; }
            key = __shfl(key, src); el = __shfl(el, src);
        }
        // __unguarded_partition(first + 1, last, pivot at first)
        const bool inr = act && lane > sf;                    // (lane < sl holds for every lane of the sub-segment)
        const bool isL = inr && key <= pivot, isR = inr && key >= pivot;
        const unsigned long long M = act ? (((sl >= 64) ? ~0ull : ((1ull << sl) - 1ull)) & ~((2ull << sf) - 1ull)) : 0ull;
        const unsigned long long Ls = __ballot(isL) & M, Rs = __ballot(isR) & M;
        const int rankL = __popcll(Ls & below), rankR = __popcll(Rs & above);
        const bool swL = isL && __popcll(Rs & above) > rankL;
        const bool swR = isR && __popcll(Ls & below) > rankR;
        const int t = __popcll(__ballot(swL) & M);
        {
            int src = lane;
            if (swL) src = nth_bit_down(Rs, rankL);
            else if (swR) src = nth_bit_up(Ls, rankR);
            key = __shfl(key, src); el = __shfl(el, src);
        }
        if (act) {
            int cut = 0x7fffffff;
            if (t < __popcll(Ls)) cut = nth_bit_up(Ls, t);
            if (t >= 1) cut = min(cut, nth_bit_down(Rs, t - 1));
            if (lane < cut) sl = cut; else sf = cut;
        }
    }
    if (lane < m) {
        o.perm[F + lane] = el;
        o.seglo[F + lane] = (unsigned short)(F + sf);
        o.seghi[F + lane] = (unsigned short)(F + sl);
    }
}
constexpr int PO_SMALL_SEG = 64;        // segments up to this size are finished by wave_small_sort

// The replay with key and element in separate arrays (any 32-bit keys): what block_std_sort_desc falls back to when the keys span
// 2^20 or more.  Call from ALL threads of a 256-thread workgroup (contains __syncthreads).  On return
// o.pl[e] = position of element e in std::sort(order, comp) of the list 0..n-1.
template <typename WS>
__device__ inline void block_std_sort_desc_wide(WS& o, int n, int tid) {
    const int lane = tid & 63;
    const int wib = tid >> 6;
    const unsigned long long lmask = (1ull << lane) - 1ull;
    for (int p = tid; p < n; p += 256) o.perm[p] = p;
    if (tid == 0) {
        o.seg_cnt[0] = 0;
        o.seg_cnt[1] = 0;
        o.small_cnt = 0;
        if (n > 16 && n <= PO_SMALL_SEG) {
            o.small_first[0] = 0; o.small_last[0] = (unsigned short)n; o.small_depth[0] = (unsigned char)(hinge_sort::floor_log2((unsigned)n) * 2);
            o.small_cnt = 1;
        } else if (n > 16) {
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
            const bool big_l = cut - first > PO_SMALL_SEG, big_r = last - cut > PO_SMALL_SEG;
            const bool sm_l = !big_l && cut - first > 16, sm_r = !big_r && last - cut > 16;
            if (!big_l && !sm_l) mark_leaf(o, first, cut, lane);
            if (!big_r && !sm_r) mark_leaf(o, cut, last, lane);
            if (lane == 0 && (sm_l || sm_r)) {
                int slot = atomicAdd(&o.small_cnt, (int)sm_l + (int)sm_r);
                if (sm_l) { o.small_first[slot] = (unsigned short)first; o.small_last[slot] = (unsigned short)cut; o.small_depth[slot] = (unsigned char)depth; ++slot; }
                if (sm_r) { o.small_first[slot] = (unsigned short)cut; o.small_last[slot] = (unsigned short)last; o.small_depth[slot] = (unsigned char)depth; }
            }
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
    // the segments of 17 .. 64 elements: each finished by one wavefront in registers
    {
        const int ns = o.small_cnt;
        for (int j = wib; j < ns; j += 4) wave_small_sort_wide(o, o.small_first[j], o.small_last[j], o.small_depth[j], lane, cmp);
    }
    __syncthreads();
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


// ------------------------------------------------------------------------------------------------
// The same replay with key and element in ONE LDS word (round 5): pk[p] = (key - kmin) << PK_SHIFT | element.  Every sweep of the
// walk above reads `key[perm[p]]` - two dependent LDS reads, the second one a gather with bank conflicts - for every position and
// level; a packed word is one stride-1 read, a swap moves one word, and the register-phase shuffles carry one value instead of two.
// The comparisons are those of the keys: comp(x, y) = (x >> PK_SHIFT) > (y >> PK_SHIFT) - the element bits never take part, so equal
// keys are as equal as before.  Usable when the keys span less than 2^(32 - PK_SHIFT) (pile-up keys are length sums, supporter keys
// other ends: both far below 2^20); block_std_sort_desc checks and falls back to the wide form otherwise.
// ------------------------------------------------------------------------------------------------
constexpr int PK_SHIFT = 12;
static_assert(PO_CAP <= (1 << PK_SHIFT), "an element index fits the low bits of a packed word");

template <typename WS>
__device__ inline void wave_small_sort_packed(WS& o, unsigned* __restrict__ pk, int F, int Lst, int depth0, int lane) {
    const int m = Lst - F;
    unsigned v = 0u;
    if (lane < m) v = pk[F + lane];
    int sf = lane < m ? 0 : lane, sl = lane < m ? m : lane;      // this lane's sub-segment, in lanes
    int dep = depth0;
    const unsigned long long below = (1ull << lane) - 1ull, above = lane == 63 ? 0ull : ~((2ull << lane) - 1ull);
    const hinge_sort::PackedCmp cmp{PK_SHIFT};
    while (true) {
        bool act = sl - sf > 16;
        if (!__any(act)) break;
        if (__any(act && dep == 0)) {
            // __partial_sort(first, last, last) of the exhausted sub-segments: through LDS, one lane
            if (lane < m) pk[F + lane] = v;
            __threadfence_block();
            unsigned long long heads = __ballot(act && dep == 0 && lane == sf);
            while (heads) {
                const int h = (int)__builtin_ctzll(heads);
                const int f = __shfl(sf, h), l = __shfl(sl, h);
                if (lane == 0) hinge_sort::heapsort_(pk, F + f, F + l, cmp);
                heads &= heads - 1ull;
            }
            __threadfence_block();
            if (act && dep == 0) { v = pk[F + lane]; sf = lane; sl = lane + 1; act = false; }
            if (!__any(act)) break;
        }
        if (act) dep -= 1;
        // __move_median_to_first(first, first + 1, mid, last - 1)
        const int mid = sf + (sl - sf) / 2;
        const int key = (int)(v >> PK_SHIFT);
        const int ka = __shfl(key, act ? sf + 1 : lane), kb = __shfl(key, act ? mid : lane), kc = __shfl(key, act ? sl - 1 : lane);
        int msel;   // comp(x, y) = x > y
        if (ka > kb) { if (kb > kc) msel = 1; else if (ka > kc) msel = 2; else msel = 0; }
        else if (ka > kc) msel = 0;
        else if (kb > kc) msel = 2;
        else msel = 1;
        const int mpos = msel == 0 ? sf + 1 : (msel == 1 ? mid : sl - 1);
        const int pivot = msel == 0 ? ka : (msel == 1 ? kb : kc);
        {
            int src = lane;
            if (act) { if (lane == sf) src = mpos; else if (lane == mpos) src = sf; }
            v = (unsigned)__shfl((int)v, src);
        }
        // __unguarded_partition(first + 1, last, pivot at first)
        const int k2 = (int)(v >> PK_SHIFT);
        const bool inr = act && lane > sf;                    // (lane < sl holds for every lane of the sub-segment)
        const bool isL = inr && k2 <= pivot, isR = inr && k2 >= pivot;
        const unsigned long long M = act ? (((sl >= 64) ? ~0ull : ((1ull << sl) - 1ull)) & ~((2ull << sf) - 1ull)) : 0ull;
        const unsigned long long Ls = __ballot(isL) & M, Rs = __ballot(isR) & M;
        const int rankL = __popcll(Ls & below), rankR = __popcll(Rs & above);
        const bool swL = isL && __popcll(Rs & above) > rankL;
        const bool swR = isR && __popcll(Ls & below) > rankR;
        const int t = __popcll(__ballot(swL) & M);
        {
            int src = lane;
            if (swL) src = nth_bit_down(Rs, rankL);
            else if (swR) src = nth_bit_up(Ls, rankR);
            v = (unsigned)__shfl((int)v, src);
        }
        if (act) {
            int cut = 0x7fffffff;
            if (t < __popcll(Ls)) cut = nth_bit_up(Ls, t);
            if (t >= 1) cut = min(cut, nth_bit_down(Rs, t - 1));
            if (lane < cut) sl = cut; else sf = cut;
        }
    }
    if (lane < m) {
        pk[F + lane] = v;
        o.seglo[F + lane] = (unsigned short)(F + sf);
        o.seghi[F + lane] = (unsigned short)(F + sl);
    }
}

template <typename WS, typename KeyOf>
__device__ inline void block_std_sort_desc_packed(WS& o, int n, int tid, int kmin, KeyOf key_of) {
    const int lane = tid & 63;
    const int wib = tid >> 6;
    const unsigned long long lmask = (1ull << lane) - 1ull;
    unsigned* __restrict__ pk = reinterpret_cast<unsigned*>(o.perm);       // pk[p] = (key - kmin) << PK_SHIFT | element at position p
    for (int p = tid; p < n; p += 256) pk[p] = ((unsigned)(key_of(p) - kmin) << PK_SHIFT) | (unsigned)p;
    if (tid == 0) {
        o.seg_cnt[0] = 0;
        o.seg_cnt[1] = 0;
        o.small_cnt = 0;
        if (n > 16 && n <= PO_SMALL_SEG) {
            o.small_first[0] = 0; o.small_last[0] = (unsigned short)n; o.small_depth[0] = (unsigned char)(hinge_sort::floor_log2((unsigned)n) * 2);
            o.small_cnt = 1;
        } else if (n > 16) {
            o.seg_first[0][0] = 0;
            o.seg_last[0][0] = (unsigned short)n;
            o.seg_depth[0][0] = (unsigned char)(hinge_sort::floor_log2((unsigned)n) * 2);
            o.seg_cnt[0] = 1;
        }
    }
    if (n <= 16 && wib == 0) mark_leaf(o, 0, n, lane);
    __syncthreads();
    const hinge_sort::PackedCmp cmp{PK_SHIFT};
    int cur = 0;
    while (true) {
        const int nc = o.seg_cnt[cur];
        if (nc == 0) break;
        const int nxt = cur ^ 1;
        for (int j = wib; j < nc; j += 4) {
            const int first = o.seg_first[cur][j], last = o.seg_last[cur][j];
            int depth = o.seg_depth[cur][j];
            if (depth == 0) {   // __partial_sort(first, last, last): heap sort, serial
                if (lane == 0) hinge_sort::heapsort_(pk, first, last, cmp);
                for (int p = first + lane; p < last; p += 64) { o.seglo[p] = (unsigned short)p; o.seghi[p] = (unsigned short)(p + 1); }
                continue;
            }
            --depth;
            // __move_median_to_first(first, first+1, mid, last-1): lanes 0..2 fetch the three candidates
            const int mid = first + (last - first) / 2;
            const int cpos = lane == 0 ? first + 1 : (lane == 1 ? mid : last - 1);
            unsigned cw = 0u;
            if (lane < 3) cw = pk[cpos];
            const int ckey = (int)(cw >> PK_SHIFT);
            const int ka = __builtin_amdgcn_readlane(ckey, 0), kb = __builtin_amdgcn_readlane(ckey, 1), kc = __builtin_amdgcn_readlane(ckey, 2);
            int msel;   // which candidate goes to `first`: 0 = a (first+1), 1 = b (mid), 2 = c (last-1); comp(x,y) = x > y
            if (ka > kb) { if (kb > kc) msel = 1; else if (ka > kc) msel = 2; else msel = 0; }
            else if (ka > kc) msel = 0;
            else if (kb > kc) msel = 2;
            else msel = 1;
            const int mpos = msel == 0 ? first + 1 : (msel == 1 ? mid : last - 1);
            const unsigned mw = (unsigned)__builtin_amdgcn_readlane((int)cw, msel);
            const int pivot = msel == 0 ? ka : (msel == 1 ? kb : kc);
            if (lane == 0) { const unsigned fw = pk[first]; pk[first] = mw; pk[mpos] = fw; }
            const int lo = first + 1, hi = last;
            // one sweep: stopper positions by rank from the left, stored at [lo + rank)
            int totalL = 0, totalR = 0;
            for (int base = lo; base < hi; base += 64) {
                const int p = base + lane;
                int x = 0;
                if (p < hi) x = (int)(pk[p] >> PK_SHIFT);
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
                const unsigned tmp = pk[a]; pk[a] = pk[b]; pk[b] = tmp;
            }
            // children: [first, cut) (the loop's continuation) and [cut, last) (the recursive call)
            const bool big_l = cut - first > PO_SMALL_SEG, big_r = last - cut > PO_SMALL_SEG;
            const bool sm_l = !big_l && cut - first > 16, sm_r = !big_r && last - cut > 16;
            if (!big_l && !sm_l) mark_leaf(o, first, cut, lane);
            if (!big_r && !sm_r) mark_leaf(o, cut, last, lane);
            if (lane == 0 && (sm_l || sm_r)) {
                int slot = atomicAdd(&o.small_cnt, (int)sm_l + (int)sm_r);
                if (sm_l) { o.small_first[slot] = (unsigned short)first; o.small_last[slot] = (unsigned short)cut; o.small_depth[slot] = (unsigned char)depth; ++slot; }
                if (sm_r) { o.small_first[slot] = (unsigned short)cut; o.small_last[slot] = (unsigned short)last; o.small_depth[slot] = (unsigned char)depth; }
            }
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
    // the segments of 17 .. 64 elements: each finished by one wavefront in registers
    {
        const int ns = o.small_cnt;
        for (int j = wib; j < ns; j += 4) wave_small_sort_packed(o, pk, o.small_first[j], o.small_last[j], o.small_depth[j], lane);
    }
    __syncthreads();
    // final insertion sort == stable sort inside each leaf; positions by element -> pr -> pl
    for (int p = tid; p < n; p += 256) {
        const int slo = o.seglo[p], shi = o.seghi[p];
        const unsigned w = pk[p];
        const unsigned x = w >> PK_SHIFT;
        int r = slo;
        for (int q = slo; q < shi; ++q) {
            const unsigned y = pk[q] >> PK_SHIFT;
            r += (y > x) || (y == x && q < p);
        }
        o.pr[w & ((1u << PK_SHIFT) - 1u)] = (unsigned short)r;
    }
    __syncthreads();
    for (int p = tid; p < n; p += 256) o.pl[p] = o.pr[p];
    __syncthreads();
}

// Call from ALL threads of a 256-thread workgroup (contains __syncthreads).  On return
// o.pl[e] = position of element e in std::sort(order, comp) of the list 0..n-1 (o.key[e] = its key; comp(x, y) = key[x] > key[y]).
template <typename WS>
__device__ inline void block_std_sort_desc(WS& o, int n, int tid) {
    // the keys' range decides the form (uniform: every thread reduces the same LDS words)
    int kmin = 0x7fffffff, kmax = (int)0x80000000;
    for (int p = tid; p < n; p += 256) { const int k = o.key[p]; kmin = min(kmin, k); kmax = max(kmax, k); }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { kmin = min(kmin, __shfl_xor(kmin, d)); kmax = max(kmax, __shfl_xor(kmax, d)); }
    if ((tid & 63) == 0) { o.krange[0][tid >> 6] = kmin; o.krange[1][tid >> 6] = kmax; }
    __syncthreads();
    kmin = min(min(o.krange[0][0], o.krange[0][1]), min(o.krange[0][2], o.krange[0][3]));
    kmax = max(max(o.krange[1][0], o.krange[1][1]), max(o.krange[1][2], o.krange[1][3]));
    __syncthreads();
#ifndef HINGE_SORT_WIDE
    if (n > 0 && (long long)kmax - (long long)kmin < (1ll << (32 - PK_SHIFT))) { block_std_sort_desc_packed(o, n, tid, kmin, [&](int p) { return o.key[p]; }); return; }
#endif
    block_std_sort_desc_wide(o, n, tid);
}

// The same for a list whose keys come from a functor (the LEAN work space has no key array): the keys are staged once in the stopper
// scratch (pl .. pr: CAP ints, free until the walk starts), their range is reduced, the packed words are built from the stage.
// Returns false - nothing sorted - when the keys span 2^(32 - PK_SHIFT) or more (the caller takes another route: never seen on
// length sums or other ends, both below 2^18).  Call from ALL threads of a 256-thread workgroup.
template <typename WS, typename KeyOf>
__device__ inline bool block_std_sort_desc_keys(WS& o, int n, int tid, KeyOf key_of) {
    static_assert(sizeof(o.pl) == sizeof(o.pr), "pl and pr: one stretch of CAP ints");
    int* __restrict__ stage = reinterpret_cast<int*>(o.pl);
    int kmin = 0x7fffffff, kmax = (int)0x80000000;
    for (int p = tid; p < n; p += 256) { const int k = key_of(p); stage[p] = k; kmin = min(kmin, k); kmax = max(kmax, k); }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { kmin = min(kmin, __shfl_xor(kmin, d)); kmax = max(kmax, __shfl_xor(kmax, d)); }
    if ((tid & 63) == 0) { o.krange[0][tid >> 6] = kmin; o.krange[1][tid >> 6] = kmax; }
    __syncthreads();
    kmin = min(min(o.krange[0][0], o.krange[0][1]), min(o.krange[0][2], o.krange[0][3]));
    kmax = max(max(o.krange[1][0], o.krange[1][1]), max(o.krange[1][2], o.krange[1][3]));
    __syncthreads();
    if (n > 0 && (long long)kmax - (long long)kmin >= (1ll << (32 - PK_SHIFT))) return false;
    block_std_sort_desc_packed(o, n, tid, kmin, [&](int p) { return stage[p]; });
    return true;
}

}  // namespace hinge
