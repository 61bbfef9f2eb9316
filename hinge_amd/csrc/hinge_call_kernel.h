// K3 of `hinge filter`: hinge calling (src/filter/filter.cpp:867-1068), one workgroup per work-list read,
// one pass over the pile-up per annotation of that read.
//
//   gather   all 256 threads stream (abpos, aepos); only overlaps whose near end falls in the annotation's
//            +-HINGE_TOLERANCE window load their B-side fields and gather mask[B] (filter.cpp:877-908).
//            Supporters are compacted into LDS in .las order.
//   order    the reference collects supporters in the order of the std::sort-ed pile-up
//            (filter.cpp:565-567).  For a short list with no confusable length tie that is simply
//            (length desc, .las order).  Otherwise the exact pile-up order is computed ONCE per read by
//            wave_std_sort_desc() and the supporters are compacted in that order.
//   sort     std::sort(pairAscend / pairDescend) of the supporters is replayed by the same routine.
//   scan     lane 0 walks the sorted list (it usually stops within the first ~10 elements).
//
// Pile-ups or supporter lists beyond PO_CAP go to k_hinge_exact (global scratch, serial).
// hinge_flag[anno slot] = 1 emit / 0 no.
#pragma once
#include "filter_kernels.h"
#include "pileup_order.h"

namespace hinge {

constexpr int HC_SMALL = 64;    // lists up to this size try the tie-free shortcut
constexpr int PRE_MAXA = 2;     // annotations covered by one count-only sweep (round 5: two instead of four - 1.2 annotations per work-list read; the registers buy a wavefront per SIMD)
// (the sort-free scan evaluation bins f - f[0] at 1 bp into 2 * CAP bins of LDS scratch)
constexpr int GATHER_LOADS = 8;       // pile-up loads a lane of k_hinge_call keeps in flight

// One undecided annotation, produced by k_hinge_count for k_hinge_call.  Both kernels cut a pile-up into the
// same four contiguous slices (one per wavefront), so the count sweep can hand the gather the slot at which
// every wavefront starts writing its supporters: the gather then needs no workgroup barrier per chunk.
struct alignas(16) HeavyItem {   // everything k_hinge_call needs, so that it starts streaming after ONE dependent load
    int read, anno;      // read id, annotation index within the read
    int base[3];         // supporters in slices 0, 0-1, 0-2 (slice 0 starts at slot 0)
    int sup, near_end;   // support; supporters that take the scan's first branch
    int n;               // pile-up size
    long long row;       // row_ptr[read]
    int mask_lo, mask_hi;
    int pos, type;       // the annotation
    unsigned slot;       // anno_off[read] + anno: index into hinge_flag
    int f0;              // smallest other end among the supporters (the origin of k_hinge_call_light's bins)
};
__device__ __forceinline__ int slice_len(int n) { return ((n + 4 * WAVE - 1) / (4 * WAVE)) * WAVE; }

// Everything the two hinge kernels read or write for ONE resident part.  A launch takes up to HINGE_BATCH_MAX of them (round 3):
// both kernels are chains of dependent look-ups over 1-2 % of the reads - ~19 us each however small the part - so the parts a
// GPU holds resident go through them in ONE launch each (workgroup b works for part b % n) and the chain is paid once.
struct HingePart {
    const int64_t* row_ptr; const int2* a_span; const int2* b_span; const unsigned* b_flag; const int2* mask;
    const int2* anno_buf; const unsigned* anno_off; const int* anno_cnt;
    const WorkItem* work_list; unsigned work_cap; const unsigned* work_shard;   // (sharded list: filter_kernels.h N_SHARD)
    unsigned char* hinge_flag;
    HeavyItem* heavy; unsigned* heavy_count; unsigned* heavy_count_big; unsigned heavy_cap;
    int2* exact_queue; unsigned* exact_count; unsigned exact_cap;
    int* status; unsigned* work_next; unsigned* work_next_big;
    unsigned* dbg;
    int force_exact;
    // second tier (round 4): what k_hinge_call_light could not settle, for k_hinge_call<CAP> (front / back by pile-up size as in `heavy`)
    HeavyItem* heavy2; unsigned* heavy2_count; unsigned* heavy2_count_big; unsigned* work_next_light;
    unsigned* work_next_small;     // round 5: the cursor of the PO_CAP_MINI instance
    // round 5: the part's 16|16 copies of a_span and b_span (both or neither; nullptr if a read has 65 536+ bases or a coordinate lies
    // outside [0, 65535]): k_hinge_count and k_hinge_call_light move 3 TB/s of pile-up columns - 12 instead of 20 bytes per overlap
    const unsigned* span16;
    const unsigned* bspan16;
    // round 6: the second-tier items chained BY READ, so that k_hinge_call<CAP> replays a read's pile-up sort once for all of its open
    // annotations (the order depends on the read only, filter.cpp:565-567).  k_hinge_call_light's pass_on links every item it appends
    // into its read's chain - rd_head[read] = stamp << 32 | newest item, rd_next2[item] = the one before it - and the FIRST item of a
    // read files the read in rl2 (front / back by pile-up size, like heavy2); k_hinge_call<CAP> then draws READS.  `stamp` changes with
    // every launch, so rd_head is never cleared.  rd_head == nullptr: items are drawn one by one as before.
    unsigned long long* rd_head; unsigned* rd_next2; unsigned* rl2; unsigned* rl2_count; unsigned* rl2_count_big; unsigned stamp;
};
constexpr unsigned CHAIN_END = 0xffffffffu;
constexpr int HEAVY_TIES = 0x40000000;   // in HeavyItem::anno of a second-tier item: k_hinge_call_light evaluated it and the tie order decides
                                         // (the same evaluation in k_hinge_call<CAP> would say the same: it goes straight to the replay)
constexpr int HINGE_BATCH_MAX = 8;
struct HingeBatch {
    HingePart part[HINGE_BATCH_MAX];
    int n;
};

// LEAN (round 6): the instance behind k_hinge_call_light - no key array, no pile-up-order copies of the supporters (an order list in
// sK instead), no sort-free evaluation (its items come flagged: the tie order decides) - 52 instead of 77 KiB at CAP = 2048: THREE
// workgroups per CU for a replay that is bound by the workgroups it has in flight
template <int CAP, bool LEAN = false>
struct HingeCallLdsT {
    WaveSortLdsT<CAP, LEAN> ws;
    unsigned short ppos[CAP];      // position of every overlap of the read in the sorted pile-up
    alignas(16) int sF[CAP];       // supporters in .las order: other end in scan-ascending form ...
    alignas(16) int sS[CAP];       // ... and the overhang on the far side; reused for the sorted lists
    unsigned short sK[CAP];        // ... and the local overlap index
    int sL[HC_SMALL];              // length sums of the first HC_SMALL supporters
    int wF[LEAN ? 1 : CAP];        // supporters in pile-up order
    int wS[LEAN ? 1 : CAP];
    int wcnt[2][WAVES_PER_BLOCK];
    int cnt;
    int need_order;
    int near_end;
    int ev_ucan, ev_bcan, ev_umust, ev_bmust, f0;   // sort-free scan: smallest f (bin) of a group with each property
    int sf_over, fmax;
    int wtot[2][WAVES_PER_BLOCK];
    unsigned next_item;
};

// ------------------------------------------------------------------------------------------------
// K3a: count-only sweep, one workgroup per work-list read, 40 bytes of LDS.  For every annotation: support and the
// number of supporters that take the scan's first branch.  These two numbers decide most annotations
// (support <= SUP: no hinge;  first-branch count > UNB: unbridged whatever the order, filter.cpp:920-931).
// Undecided annotations get hinge_flag = 2 and their read goes to the heavy list for k_hinge_call.
// ------------------------------------------------------------------------------------------------
// CW wavefronts per workgroup (= per work-list read): four, one slice each, or two with two consecutive slices each - half the
// wavefronts per read, twice the reads in flight (the kernel is bound by the reads it has in flight: a chain of round trips per read)
// PK: every part of the batch has the 16|16 copies of its span columns (a template parameter: as a run-time branch both load forms'
// registers were live at once - 158 VGPRs, three wavefronts per SIMD; the kernel lives on the reads it has in flight)
// (round 6: five wavefronts per SIMD by decree - amdgpu_waves_per_eu(5, 5): 96 VGPRs, 17 of them spilled into the per-overlap loop -
// takes 68 us where the compiler's own 113 VGPRs take 52.5; -DHINGE_COUNT_WAVES_ATTR=... to try another)
#ifndef HINGE_COUNT_WAVES_ATTR
#define HINGE_COUNT_WAVES_ATTR
#endif
template <int CW, bool PK>
__global__ __launch_bounds__(CW * WAVE) HINGE_COUNT_WAVES_ATTR void k_hinge_count(FilterDev P, HingeBatch B) {
    static_assert(CW == 2 || CW == 4, "two or four wavefronts per read");
    constexpr int SPW = WAVES_PER_BLOCK / CW;   // slices per wavefront
    const unsigned n_parts = (unsigned)B.n;
    const HingePart& A = B.part[blockIdx.x % n_parts];              // (uniform: scalar loads from the kernel arguments)
    const unsigned bx = blockIdx.x / n_parts, gx = gridDim.x / n_parts;
    const int64_t* __restrict__ row_ptr = A.row_ptr; const int2* __restrict__ a_span = A.a_span; const int2* __restrict__ b_span = A.b_span;
    const unsigned* __restrict__ b_flag = A.b_flag; const int2* __restrict__ mask = A.mask; const int2* __restrict__ anno_buf = A.anno_buf;
    const WorkItem* __restrict__ work_list = A.work_list; const unsigned* __restrict__ work_shard = A.work_shard;
    unsigned char* __restrict__ hinge_flag = A.hinge_flag; HeavyItem* __restrict__ heavy = A.heavy;
    unsigned* __restrict__ heavy_count = A.heavy_count; unsigned* __restrict__ heavy_count_big = A.heavy_count_big;
    const unsigned heavy_cap = A.heavy_cap; const int force_exact = A.force_exact; unsigned* __restrict__ dbg = A.dbg;
    const unsigned* __restrict__ span16 = A.span16; const unsigned* __restrict__ bspan16 = A.bspan16;
    (void)row_ptr;
    __shared__ int s_sup[PRE_MAXA][WAVES_PER_BLOCK], s_near[PRE_MAXA][WAVES_PER_BLOCK], s_minf[PRE_MAXA][WAVES_PER_BLOCK];
    const int tid = threadIdx.x;
    const int lane = lane_id();
    const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (bx >= gx) return;   // (a grid that is not a multiple of the parts: the surplus workgroups have no share)
    // The kernel is a chain of dependent memory round trips (~2 us each) per read, so every link counts: the first work item is
    // fetched together with the number of items (the list has work_cap slots: the read is legal whether or not the slot is in use)
    WorkItem wi_first;
    if (bx < A.work_cap) wi_first = work_list[bx];
    // the sharded work list: slot w is in use iff w / N_SHARD < count[w % N_SHARD]; lane l keeps shard l's count
    static_assert(N_SHARD == WAVE, "one shard count per lane");
    const unsigned my_count = work_shard[lane * SHARD_STRIDE];
    const unsigned nwork = min((unsigned)N_SHARD * (unsigned)wave_max((int)my_count), A.work_cap);
    for (unsigned w = bx; w < nwork; w += gx) {   // one workgroup per work-list read
#ifdef HINGE_TIMING
        const unsigned long long tc0 = wall_clock64();
#endif
        const WorkItem wi = wi_first;
        if (w + gx < nwork) wi_first = work_list[w + gx];   // the next item travels while this one is worked on
        if ((w >> 6) >= (unsigned)__builtin_amdgcn_readlane((int)my_count, (int)(w & (N_SHARD - 1)))) continue;   // (a hole of a shorter shard; uniform)
        const int i = wi.read;
        const int64_t s = wi.row, e = wi.row + wi.n;
        const int2 mk = make_int2(wi.mask_lo, wi.mask_hi);
        const unsigned off = wi.off;
        const int cnt = wi.cnt;
        const int q = slice_len((int)(e - s));
        const int64_t k_lo = s + (int64_t)wib * SPW * q, k_hi = min(e, k_lo + (int64_t)SPW * q);   // this wavefront's slice(s)
        const int64_t k_mid = k_lo + q;                                                                   // (SPW == 2: where its second slice begins)
        for (int a0 = 0; a0 < cnt; a0 += PRE_MAXA) {
            const int na = min(PRE_MAXA, cnt - a0);
            int apos[PRE_MAXA], atype[PRE_MAXA], csup[PRE_MAXA], cnear[PRE_MAXA], cminf[PRE_MAXA], csup_hi[PRE_MAXA];   // csup_hi: of csup, in the wavefront's second slice
#pragma unroll
            for (int a = 0; a < PRE_MAXA; a++) {
                const int2 an = a < na ? (a0 == 0 ? wi.anno[a] : anno_buf[off + a0 + a]) : make_int2(0, 0);
                apos[a] = an.x; atype[a] = an.y; csup[a] = 0; cnear[a] = 0; cminf[a] = INT_MAX; csup_hi[a] = 0;
            }
            for (int64_t k0 = k_lo; k0 < k_hi; k0 += GATHER_LOADS * WAVE) {
                // three dependent round trips per GATHER_LOADS * 64 overlaps (spans, B-side fields, mask[B]) instead of per 64
                // (round 6, PK: the span words stay PACKED in their registers - 16 instead of 32 for a batch - and are taken apart where
                // they are used: the kernel lives on the reads it has in flight, i.e. on wavefronts per SIMD)
                int2 avw[PK ? 1 : GATHER_LOADS], bsw[PK ? 1 : GATHER_LOADS], mb[GATHER_LOADS];
                unsigned svp[PK ? GATHER_LOADS : 1], bvp[PK ? GATHER_LOADS : 1];
                unsigned bf[GATHER_LOADS];
                auto AV = [&](int u) { return PK ? make_int2((int)(svp[u] & 0xffffu), (int)(svp[u] >> 16)) : avw[u]; };
                auto BS = [&](int u) { return PK ? make_int2((int)(bvp[u] & 0xffffu), (int)(bvp[u] >> 16)) : bsw[u]; };
                bool nearw[GATHER_LOADS];
                // (the B-side fields are loaded with the spans, needed or not: 12 more bytes per overlap of a work-list read - 1-2 % of
                // the part - buy one dependent round trip less per batch)
                if (PK) {   // the kernel moves TB/s of pile-up columns: the packed copies of the spans where the parts have them
#pragma unroll
                    for (int u = 0; u < GATHER_LOADS; u++) {
                        const int64_t k = k0 + u * WAVE + lane;
                        const bool in = k < k_hi;
                        svp[u] = in ? span16[k] : 0u;
                        bvp[u] = in ? bspan16[k] : 0u;
                        bf[u] = in ? b_flag[k] : 0u;
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < GATHER_LOADS; u++) {
                        const int64_t k = k0 + u * WAVE + lane;
                        const bool in = k < k_hi;
                        avw[u] = in ? a_span[k] : make_int2(0, 0);
                        bf[u] = in ? b_flag[k] : 0u;
                        bsw[u] = in ? b_span[k] : make_int2(0, 0);
                    }
                }
#pragma unroll
                for (int u = 0; u < GATHER_LOADS; u++) {
                    const int64_t k = k0 + u * WAVE + lane;
                    bool nr = false;
#pragma unroll
                    for (int a = 0; a < PRE_MAXA; a++) {
                        if (a >= na) break;
                        const int c = atype[a] == -1 ? AV(u).y : AV(u).x;
                        nr = nr || ((c > apos[a] - P.tol) && (c < apos[a] + P.tol));
                    }
                    nearw[u] = nr && k < k_hi;
                }
#pragma unroll
                for (int u = 0; u < GATHER_LOADS; u++) mb[u] = nearw[u] ? mask[bf[u] & 0x7fffffffu] : make_int2(0, 0);
#pragma unroll
                for (int u = 0; u < GATHER_LOADS; u++) {
                    if (!nearw[u]) continue;
                    int L, R;
                    overhangs(BS(u), (int)(bf[u] >> 31), mb[u], L, R);
#pragma unroll
                    for (int a = 0; a < PRE_MAXA; a++) {
                        if (a >= na) break;
                        const int c = atype[a] == -1 ? AV(u).y : AV(u).x;
                        if ((c > apos[a] - P.tol) && (c < apos[a] + P.tol)) {
                            const bool sup = atype[a] == -1 ? (R > P.theta) : (L > P.theta);
                            if (sup) {
                                csup[a]++;
                                if (SPW == 2) csup_hi[a] += (k0 + u * WAVE + lane >= k_mid);
                                const int f = atype[a] == -1 ? AV(u).x : -AV(u).y;
                                const int m0 = atype[a] == -1 ? mk.x : -mk.y;
                                cnear[a] += (f - m0 < P.bin_len);
                                cminf[a] = min(cminf[a], f);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < PRE_MAXA; a++) {
                if (a >= na) break;
                const int psup = wave_sum(csup[a]), pnear = wave_sum(cnear[a]), pminf = -wave_max(-cminf[a]);
                if (SPW == 1) { if (lane == 0) { s_sup[a][wib] = psup; s_near[a][wib] = pnear; s_minf[a][wib] = pminf; } }
                else {
                    const int phi = wave_sum(csup_hi[a]);
                    if (lane == 0) { s_sup[a][2 * wib] = psup - phi; s_sup[a][2 * wib + 1] = phi; s_near[a][2 * wib] = pnear; s_near[a][2 * wib + 1] = 0;
                                     s_minf[a][2 * wib] = pminf; s_minf[a][2 * wib + 1] = INT_MAX; }
                }
            }
            __syncthreads();
            if (tid < na) {
                const int b1 = s_sup[tid][0], b2 = b1 + s_sup[tid][1], b3 = b2 + s_sup[tid][2];
                const int psup = b3 + s_sup[tid][3];
                const int pnear = s_near[tid][0] + s_near[tid][1] + s_near[tid][2] + s_near[tid][3];
                int quick = 2;
                if (force_exact == 0) {
                    if (psup <= P.sup) quick = 0;                      // needs support >= SUP to be scanned and > SUP to be emitted
                    else if (P.unb >= 0 && pnear > P.unb) quick = 1;   // the first UNB+1 sorted supporters all take branch 1
                }
                hinge_flag[off + a0 + tid] = (unsigned char)quick;
                if (quick == 2) {
                    HeavyItem it;
                    it.read = i; it.anno = a0 + tid;
                    it.base[0] = b1; it.base[1] = b2; it.base[2] = b3;
                    it.sup = psup; it.near_end = pnear;
                    it.f0 = min(min(s_minf[tid][0], s_minf[tid][1]), min(s_minf[tid][2], s_minf[tid][3]));
                    it.n = (int)(e - s); it.row = s; it.mask_lo = mk.x; it.mask_hi = mk.y;
                    {   // (the annotation is in registers already - indexed by a lane-varying tid, so by selects - not a dependent re-load)
                        static_assert(PRE_MAXA == 2, "select chain below");
                        it.pos = tid == 0 ? apos[0] : apos[1];
                        it.type = tid == 0 ? atype[0] : atype[1];
                    }
                    it.slot = off + a0 + tid;
                    // pile-ups that fit the half-size instance of k_hinge_call from the front, the others from the back
                    if (it.n <= PO_CAP_SMALL) heavy[atomicAdd(heavy_count, 1u)] = it;
                    else heavy[heavy_cap - 1u - atomicAdd(heavy_count_big, 1u)] = it;
                } else if (dbg) atomicAdd(&dbg[quick ? 3 : 0], 1u);
            }
            __syncthreads();
        }
#ifdef HINGE_TIMING
        if (tid == 0 && dbg) { const unsigned dt = (unsigned)(wall_clock64() - tc0); atomicAdd(&dbg[4], dt); atomicAdd(&dbg[5], 1u); atomicMax(&dbg[15], dt); atomicMax(&dbg[7], (unsigned)(e - s)); }
#endif
    }
}

// ------------------------------------------------------------------------------------------------
// K3b (round 4): the ORDER-INDEPENDENT evaluation of an open annotation in 18 KiB of LDS, so that every open annotation of a pass
// is worked on at once.  k_hinge_call<2048> sizes its LDS for the exact replay of a pile-up's std::sort (72 KiB: two workgroups per
// CU, 512 items in flight) although on the E. coli restatement no item ever needs the replay; its items - pile-ups of 1 666 overlaps
// with 784 supporters on average, 11 us each - queue up 2.4 deep behind those 512 slots (one workgroup per CU: 78 us, two: 51).
// Same rules as k_hinge_call's "sort-free evaluation" below, word for word, on a leaner structure:
//   * the supporters are binned AS THEY ARE GATHERED (no supporter lists): the bins start at the smallest other end, which
//     k_hinge_count finds in its sweep and hands over in the item - known before the first load returns; one 32-bit word per 1-bp
//     bin: g2 | g3 << 10 | g0 << 20 (at most LIGHT_CAP = 1023 supporters);
//   * the bins are a counting sort of the other ends: the non-empty ones are compacted in bin order (thread t walks its sixteen
//     bins) with their inclusive prefixes cat-2/3 | all << 16 - at most as many groups as supporters, not 4096 bins to evaluate
//     (a first form that evaluated every bin against per-block prefixes spent 12 us per item on it, divergent and vector-bound);
//   * every group: g2 / g3 / g from its word, `before` from its prefix, W from the prefix of the last group within BIN - 1 of it
//     (a binary search over the compacted bins), the four properties, the smallest bin that has each.
// What it cannot settle - the tie order decides, more than LIGHT_CAP supporters, an other end beyond the LIGHT_BINS bins - goes on to
// the second-tier list (`heavy2`) for k_hinge_call<CAP>, untouched.  (A bitonic sort of the other ends instead of bins - no span
// limit - was built first: 20 us per item for the sort alone with every item in flight; the bins take 2.)
// ------------------------------------------------------------------------------------------------
constexpr int LIGHT_CAP = 1023;
constexpr int LIGHT_BINS = 4096;      // 1-bp bins behind m0 + BIN
#ifndef HINGE_LIGHT_LOADS
#define HINGE_LIGHT_LOADS 8
#endif
constexpr int LIGHT_LOADS = HINGE_LIGHT_LOADS;   // pile-up loads in flight per lane
struct alignas(32) HingeLightLds {
    alignas(32) unsigned cnt[LIGHT_BINS];          // g2 | g3 << 10 | g0 << 20 of the supporters with other end m0 + BIN + bin
    unsigned g_cnt[LIGHT_CAP + 1], g_pre[LIGHT_CAP + 1];   // the non-empty bins in bin order: their word, and the inclusive prefix cat-2/3 | all << 16
    unsigned short g_bin[LIGHT_CAP + 1];
    int wsum[2][WAVES_PER_BLOCK];
    int groups, over, ev_ucan, ev_bcan, ev_umust, ev_bmust;
    unsigned next_item;
};
__global__ __launch_bounds__(BLOCK) void k_hinge_call_light(FilterDev P, HingeBatch B) {
    const HingePart& A = B.part[blockIdx.x % (unsigned)B.n];
    const int2* __restrict__ a_span = A.a_span; const int2* __restrict__ b_span = A.b_span;
    const unsigned* __restrict__ b_flag = A.b_flag; const int2* __restrict__ mask = A.mask;
    const HeavyItem* __restrict__ heavy = A.heavy; HeavyItem* __restrict__ heavy2 = A.heavy2;
    unsigned char* __restrict__ hinge_flag = A.hinge_flag;
    const unsigned heavy_cap = A.heavy_cap;
    const unsigned* __restrict__ span16 = A.span16; const unsigned* __restrict__ bspan16 = A.bspan16;
    __shared__ HingeLightLds S;
    const int tid = threadIdx.x;
    const int lane = lane_id();
    const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned n_small = *A.heavy_count, n_big = *A.heavy_count_big, nwork = n_small + n_big;
#ifdef HINGE_TIMING
    const unsigned long long t_start = wall_clock64();
#endif
    auto pass_on = [&](const HeavyItem& it) {   // (one thread)
        const bool small = it.n <= PO_CAP_SMALL;
        const unsigned idx = small ? atomicAdd(A.heavy2_count, 1u) : heavy_cap - 1u - atomicAdd(A.heavy2_count_big, 1u);
        heavy2[idx] = it;
        if (A.rd_head) {   // the item joins its read's chain; the first item of a read files the read
            const unsigned long long old = atomicExch(&A.rd_head[it.read], ((unsigned long long)A.stamp << 32) | idx);
            if ((unsigned)(old >> 32) == A.stamp) A.rd_next2[idx] = (unsigned)old;
            else {
                A.rd_next2[idx] = CHAIN_END;
                A.rl2[small ? atomicAdd(A.rl2_count, 1u) : heavy_cap - 1u - atomicAdd(A.rl2_count_big, 1u)] = (unsigned)it.read;
            }
        }
    };
    {   // the bins are zero between items: cleared here once, and after an item only where it counted
        uint4* z = reinterpret_cast<uint4*>(S.cnt);
        for (int t = tid; t < LIGHT_BINS / 4; t += BLOCK) z[t] = make_uint4(0u, 0u, 0u, 0u);
    }
    auto g23_of = [](unsigned c) { return (int)((c & 1023u) + ((c >> 10) & 1023u)); };
    auto all_of = [](unsigned c) { return (int)((c & 1023u) + ((c >> 10) & 1023u) + (c >> 20)); };
    while (true) {
        __syncthreads();
        if (tid == 0) S.next_item = atomicAdd(A.work_next_light, 1u);
        __syncthreads();
        const unsigned w = S.next_item;
        if (w >= nwork) break;
#ifdef HINGE_TIMING
        const unsigned long long tm0 = wall_clock64();
#endif
        const HeavyItem item = heavy[w < n_small ? w : heavy_cap - 1u - (w - n_small)];
        const int sup = item.sup;
        if (sup <= P.sup || (P.unb >= 0 && item.near_end > P.unb)) {   // (k_hinge_count settles these itself; kept for symmetry with k_hinge_call)
            if (tid == 0) hinge_flag[item.slot] = sup <= P.sup ? 0 : 1;
            continue;
        }
        if (sup > LIGHT_CAP || P.bin_len < 1 || P.bin_len >= LIGHT_BINS) {
            if (tid == 0) pass_on(item);
            continue;
        }
        const int64_t s = item.row;
        const int n = item.n;
        const int64_t e = s + n;
        const int2 mk = make_int2(item.mask_lo, item.mask_hi);
        const int pos = item.pos, type = item.type;
        const int m0 = type == -1 ? mk.x : -mk.y;
        const int origin = item.f0;                               // other end of bin 0: the smallest among the supporters (k_hinge_count found it)
        if (tid == 0) { S.over = 0; S.ev_ucan = INT_MAX; S.ev_bcan = INT_MAX; S.ev_umust = INT_MAX; S.ev_bmust = INT_MAX; }
        __syncthreads();
        // ---- gather + bin: every wavefront streams its own slice of the pile-up (the slices of k_hinge_count: any cut would do) ----
        {
            const int q = slice_len(n);
            const int64_t k_lo = s + (int64_t)wib * q, k_hi = min(e, k_lo + q);
            bool over_l = false;
            for (int64_t k0 = k_lo; k0 < k_hi; k0 += LIGHT_LOADS * WAVE) {
                int2 av[LIGHT_LOADS], bs[LIGHT_LOADS], mb[LIGHT_LOADS];
                unsigned bf[LIGHT_LOADS];
                bool nearw[LIGHT_LOADS];
                if (span16) {   // (uniform) 12 instead of 20 bytes per overlap, as in k_hinge_count
#pragma unroll
                    for (int u = 0; u < LIGHT_LOADS; u++) {
                        const int64_t k = k0 + u * WAVE + lane;
                        const bool in = k < k_hi;
                        const unsigned sv = in ? span16[k] : 0u, bv = in ? bspan16[k] : 0u;
                        av[u] = make_int2((int)(sv & 0xffffu), (int)(sv >> 16));
                        bf[u] = in ? b_flag[k] : 0u;
                        bs[u] = make_int2((int)(bv & 0xffffu), (int)(bv >> 16));
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < LIGHT_LOADS; u++) {
                        const int64_t k = k0 + u * WAVE + lane;
                        const bool in = k < k_hi;
                        av[u] = in ? a_span[k] : make_int2(0, 0);
                        bf[u] = in ? b_flag[k] : 0u;
                        bs[u] = in ? b_span[k] : make_int2(0, 0);
                    }
                }
#pragma unroll
                for (int u = 0; u < LIGHT_LOADS; u++) {
                    const int64_t k = k0 + u * WAVE + lane;
                    const int c = type == -1 ? av[u].y : av[u].x;
                    nearw[u] = k < k_hi && (c > pos - P.tol) && (c < pos + P.tol);
                }
#pragma unroll
                for (int u = 0; u < LIGHT_LOADS; u++) mb[u] = nearw[u] ? mask[bf[u] & 0x7fffffffu] : make_int2(0, 0);
#pragma unroll
                for (int u = 0; u < LIGHT_LOADS; u++) {
                    if (!nearw[u]) continue;
                    int L, R;
                    overhangs(bs[u], (int)(bf[u] >> 31), mb[u], L, R);
                    bool sp;
                    int f, sec;
                    if (type == -1) { sp = R > P.theta; f = av[u].x; sec = L; }
                    else { sp = L > P.theta; f = -av[u].y; sec = R; }
                    if (!sp) continue;
                    if (f - m0 < P.bin_len) continue;              // first-branch prefix: in no group (k_hinge_count counted it: near_end)
                    const int rel = f - origin;                     // >= 0
                    if (rel >= LIGHT_BINS) over_l = true;
                    else atomicAdd(&S.cnt[rel], sec < P.theta ? 1u : sec > P.theta ? (1u << 10) : (1u << 20));
                }
            }
            if (__any(over_l) && lane == 0) S.over = 1;
        }
        __syncthreads();
#ifdef HINGE_TIMING
        const unsigned long long tm1 = wall_clock64();
        if (tid == 0 && A.dbg) { atomicAdd(&A.dbg[8], (unsigned)(tm1 - tm0)); atomicAdd(&A.dbg[9], (unsigned)n); atomicAdd(&A.dbg[10], 1u); atomicAdd(&A.dbg[12], (unsigned)sup); }
#endif
        if (S.over) {   // (uniform) an other end beyond the bins
            if (tid == 0) pass_on(item);
            uint4* z = reinterpret_cast<uint4*>(S.cnt);
            for (int t = tid; t < LIGHT_BINS / 4; t += BLOCK) z[t] = make_uint4(0u, 0u, 0u, 0u);
            continue;
        }
        // ---- the non-empty bins, in bin order, with their inclusive prefixes (cat-2/3 | all << 16): thread t owns bins 16 t .. 16 t + 15 ----
        // (the bins are a counting sort of the other ends; what follows works on the <= sup groups, not on 4096 bins)
        {
            const uint4* c4 = reinterpret_cast<const uint4*>(S.cnt) + 4 * tid;
            unsigned c[16];
#pragma unroll
            for (int h = 0; h < 4; h++) { const uint4 x = c4[h]; c[4 * h] = x.x; c[4 * h + 1] = x.y; c[4 * h + 2] = x.z; c[4 * h + 3] = x.w; }
            int nz = 0, sum = 0;
            if ((c[0] | c[1] | c[2] | c[3] | c[4] | c[5] | c[6] | c[7] | c[8] | c[9] | c[10] | c[11] | c[12] | c[13] | c[14] | c[15]) != 0u) {
#pragma unroll
                for (int h = 0; h < 16; h++)
                    if (c[h] != 0u) { nz++; sum += g23_of(c[h]) | (all_of(c[h]) << 16); S.cnt[16 * tid + h] = 0u; }   // (and the bin is clear for the next item)
            }
            const int nz_incl = wave_incl_scan(nz), sum_incl = wave_incl_scan(sum);      // (sum: both halves <= 1023)
            if (lane == WAVE - 1) { S.wsum[0][wib] = nz_incl; S.wsum[1][wib] = sum_incl; }
            __syncthreads();
            int at = nz_incl - nz, run = sum_incl - sum;
            for (int v = 0; v < wib; v++) { at += S.wsum[0][v]; run += S.wsum[1][v]; }
            if (tid == BLOCK - 1) S.groups = at + nz;
#pragma unroll
            for (int h = 0; h < 16; h++) {
                if (c[h] == 0u) continue;
                run += g23_of(c[h]) | (all_of(c[h]) << 16);
                S.g_bin[at] = (unsigned short)(16 * tid + h);
                S.g_cnt[at] = c[h];
                S.g_pre[at] = (unsigned)run;
                at++;
            }
        }
        __syncthreads();
#ifdef HINGE_TIMING
        const unsigned long long tm2 = wall_clock64();
        if (tid == 0 && A.dbg) atomicAdd(&A.dbg[13], (unsigned)(tm2 - tm1));
#endif
        // ---- every non-empty bin is a group ----
        {
            const int c1 = item.near_end, m = S.groups;
            int m_ucan = INT_MAX, m_bcan = INT_MAX, m_umust = INT_MAX, m_bmust = INT_MAX;
            for (int e = tid; e < m; e += BLOCK) {
                const int b = (int)S.g_bin[e];
                const unsigned c = S.g_cnt[e], pre = S.g_pre[e];
                const int g2 = (int)(c & 1023u), g3 = (int)((c >> 10) & 1023u), g = g2 + g3 + (int)(c >> 20);
                const int before = (int)(pre & 0xffffu) - (g2 + g3);
                // W = supporters with f < f' < f + BIN: up to the last group whose bin is <= b + BIN - 1
                int lo = e, hi = m - 1;                              // (the answer is in [e, m - 1])
                const int lim = b + P.bin_len - 1;
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((int)S.g_bin[mid] <= lim) lo = mid; else hi = mid - 1; }
                const int W = (int)(S.g_pre[lo] >> 16) - (int)(pre >> 16);
                const bool far = b > P.bin_len;                      // f - f[0] > BIN
                const bool ucan = far && g2 >= 1 && (c1 + before + g2 + g3 > P.unb);
                const bool bcan = g3 >= 1 && (g + W > P.pil);
                const bool umust = far && g2 >= 1 && (c1 + before + g2 > P.unb) && !bcan;
                const bool bmust = g3 >= 1 && (g3 + W > P.pil) && !ucan;
                if (ucan) m_ucan = min(m_ucan, b);
                if (bcan) m_bcan = min(m_bcan, b);
                if (umust) m_umust = min(m_umust, b);
                if (bmust) m_bmust = min(m_bmust, b);
            }
            m_ucan = -wave_max(-m_ucan); m_bcan = -wave_max(-m_bcan); m_umust = -wave_max(-m_umust); m_bmust = -wave_max(-m_bmust);
            if (lane == 0) {
                if (m_ucan != INT_MAX) atomicMin(&S.ev_ucan, m_ucan);
                if (m_bcan != INT_MAX) atomicMin(&S.ev_bcan, m_bcan);
                if (m_umust != INT_MAX) atomicMin(&S.ev_umust, m_umust);
                if (m_bmust != INT_MAX) atomicMin(&S.ev_bmust, m_bmust);
            }
        }
        __syncthreads();
#ifdef HINGE_TIMING
        if (tid == 0 && A.dbg) { atomicAdd(&A.dbg[14], (unsigned)(wall_clock64() - tm2)); atomicAdd(&A.dbg[11], (unsigned)(wall_clock64() - tm1)); atomicMax(&A.dbg[15], (unsigned)(wall_clock64() - tm0)); atomicMax(&A.dbg[7], (unsigned)n);
                                 atomicMax(&A.dbg[6], (unsigned)(wall_clock64() - t_start)); }
#endif
        if (tid == 0) {
            if (S.ev_ucan == INT_MAX || S.ev_bmust < S.ev_ucan) hinge_flag[item.slot] = 0;     // bridged
            else if (S.ev_umust < S.ev_bcan) hinge_flag[item.slot] = 1;                          // unbridged whatever the tie order
            else { HeavyItem it2 = item; it2.anno |= HEAVY_TIES; pass_on(it2); }                 // the tie order decides: exact replay
        }
    }
}

// CAP: capacity of the LDS lists (pile-up size, supporters) = half the number of 1-bp bins.  The host launches the
// PO_CAP_SMALL instance (72 KiB of LDS: two workgroups per CU) over the items whose pile-up fits it - they are appended from
// the front of `heavy` - and the PO_CAP instance (one workgroup per CU) over the rest, appended from the back (`from_back`).
template <int CAP, bool LEAN = false>
__global__ __launch_bounds__(BLOCK) void k_hinge_call(FilterDev P, HingeBatch B, int from_back, int tier2 /*1: the items k_hinge_call_light passed on*/,
                                                      int n_min, int n_max /*only items with n_min <= pile-up size <= n_max*/, int small_cursor) {
    const HingePart& A = B.part[blockIdx.x % (unsigned)B.n];
    const int64_t* __restrict__ row_ptr = A.row_ptr; const int2* __restrict__ a_span = A.a_span; const int2* __restrict__ b_span = A.b_span;
    const unsigned* __restrict__ b_flag = A.b_flag; const int2* __restrict__ mask = A.mask; const int2* __restrict__ anno_buf = A.anno_buf;
    const unsigned* __restrict__ anno_off = A.anno_off; const HeavyItem* __restrict__ heavy = tier2 ? A.heavy2 : A.heavy;
    const unsigned* __restrict__ heavy_count = tier2 ? (from_back ? A.heavy2_count_big : A.heavy2_count) : (from_back ? A.heavy_count_big : A.heavy_count);
    unsigned char* __restrict__ hinge_flag = A.hinge_flag; int2* __restrict__ exact_queue = A.exact_queue;
    unsigned* __restrict__ exact_count = A.exact_count; const unsigned exact_cap = A.exact_cap; const int force_exact = A.force_exact;
    int* __restrict__ status = A.status; unsigned* __restrict__ work_next = small_cursor ? A.work_next_small : from_back ? A.work_next_big : A.work_next;
    unsigned* __restrict__ dbg = A.dbg; const unsigned heavy_cap = A.heavy_cap;
    (void)row_ptr; (void)anno_off; (void)anno_buf;
    constexpr int SF_BINS = 2 * CAP;   // 1-bp bins of f - f[0] for the sort-free scan evaluation
    __shared__ HingeCallLdsT<CAP, LEAN> S;
    const int tid = threadIdx.x;
    const int lane = lane_id();
    const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned long long lmask = (1ull << lane) - 1ull;
    // tier2 == 2: ONE instance for both ends of the second-tier list (behind k_hinge_call_light little is left, and an empty launch
    // costs 5 us): first the back (the pile-ups only this instance holds), then the front
    const bool grouped = tier2 != 0 && A.rd_head != nullptr;     // the work list holds READS, every read's items hang on its chain
    const unsigned n_first = grouped ? *(from_back ? A.rl2_count_big : A.rl2_count) : *heavy_count;
    const unsigned n_other = tier2 == 2 ? (grouped ? *(from_back ? A.rl2_count : A.rl2_count_big) : *(from_back ? A.heavy2_count : A.heavy2_count_big)) : 0u;
    const unsigned nwork = n_first + n_other;
    while (true) {
        // dynamic work distribution: annotations differ by orders of magnitude in cost
        __syncthreads();
        if (tid == 0) S.next_item = atomicAdd(work_next, 1u);
        __syncthreads();
        const unsigned w = S.next_item;
        if (w >= nwork) break;
#ifdef HINGE_TIMING
        const unsigned long long tm0 = wall_clock64();
#endif
        const bool back = w < n_first ? from_back != 0 : from_back == 0;
        const unsigned wi_ = w < n_first ? w : w - n_first;
        unsigned chain = back ? heavy_cap - 1u - wi_ : wi_;
        if (grouped) chain = (unsigned)A.rd_head[A.rl2[chain]];
        bool order_valid = false;        // S.ppos holds this read's pile-up order (block-uniform)
        for (bool first_of_read = true; chain != CHAIN_END; chain = grouped ? A.rd_next2[chain] : CHAIN_END, first_of_read = false) {
        if (!first_of_read) __syncthreads();   // (the previous item's scan reads the lists the next gather overwrites)
        const HeavyItem item = heavy[chain];
        const int i = item.read, a = item.anno & ~HEAVY_TIES;
        const bool ties_known = (item.anno & HEAVY_TIES) != 0;
        const int64_t s = item.row;
        const int n = item.n;
        if (n < n_min || n > n_max) continue;     // another instance's item (block-uniform)
        const int64_t e = s + n;
        const int2 mk = make_int2(item.mask_lo, item.mask_hi);
        {
            const int pos = item.pos, type = item.type;
            if (tid == 0) {
                S.cnt = item.sup; S.need_order = 0; S.near_end = item.near_end;
                S.ev_ucan = INT_MAX; S.ev_bcan = INT_MAX; S.ev_umust = INT_MAX; S.ev_bmust = INT_MAX; S.f0 = INT_MAX; S.fmax = INT_MIN; S.sf_over = 0;
            }
            __syncthreads();
            const int m0 = type == -1 ? mk.x : -mk.y;
            int* bin23 = S.wF;                                   // [SF_BINS] g2 | g3 << 16   (wF..wS are contiguous)
            unsigned short* bin0 = reinterpret_cast<unsigned short*>(S.ws.key);    // [SF_BINS] g0
            unsigned short* p23 = S.ws.pl;                       // [SF_BINS] inclusive prefix of g2 + g3 (pl..pr)
            unsigned short* pall = S.ws.seglo;                   // [SF_BINS] inclusive prefix of g       (seglo..seghi)
            int fmin_w = INT_MAX, fmax_w = INT_MIN;              // range of the supporters' other ends (this wavefront's slice)
            // ---- gather: every wavefront streams its own slice and writes from the slot k_hinge_count computed ----
            {
                const int q = slice_len(n);
                const int64_t k_lo = s + (int64_t)wib * q, k_hi = min(e, k_lo + q);
                int slot0 = wib == 0 ? 0 : item.base[wib - 1];   // wave-uniform
                for (int64_t k0 = k_lo; k0 < k_hi; k0 += GATHER_LOADS * WAVE) {
                    // three dependent round trips per GATHER_LOADS * 64 overlaps: spans, B-side fields, mask[B]
                    int2 av[GATHER_LOADS], bs[GATHER_LOADS], mb[GATHER_LOADS];
                    unsigned bf[GATHER_LOADS];
                    bool nearw[GATHER_LOADS];
#pragma unroll
                    for (int u = 0; u < GATHER_LOADS; u++) {   // (B-side fields with the spans, needed or not: one round trip less)
                        const int64_t k = k0 + u * WAVE + lane;
                        const bool in = k < k_hi;
                        av[u] = in ? a_span[k] : make_int2(0, 0);
                        bf[u] = in ? b_flag[k] : 0u;
                        bs[u] = in ? b_span[k] : make_int2(0, 0);
                    }
                    if (!LEAN && k0 == k_lo) {   // the sort-free evaluation's bins are cleared while the first loads are in flight
                        int4* z23 = reinterpret_cast<int4*>(bin23);
                        int4* z0 = reinterpret_cast<int4*>(bin0);
                        for (int b = tid; b < SF_BINS / 4; b += BLOCK) z23[b] = make_int4(0, 0, 0, 0);
                        for (int b = tid; b < SF_BINS / 8; b += BLOCK) z0[b] = make_int4(0, 0, 0, 0);
                    }
#pragma unroll
                    for (int u = 0; u < GATHER_LOADS; u++) {
                        const int64_t k = k0 + u * WAVE + lane;
                        const int c = type == -1 ? av[u].y : av[u].x;
                        nearw[u] = k < k_hi && (c > pos - P.tol) && (c < pos + P.tol);
                    }
#pragma unroll
                    for (int u = 0; u < GATHER_LOADS; u++) mb[u] = nearw[u] ? mask[bf[u] & 0x7fffffffu] : make_int2(0, 0);
#pragma unroll
                    for (int u = 0; u < GATHER_LOADS; u++) {
                        if (k0 + u * WAVE >= k_hi) break;   // wave-uniform
                        const int64_t k = k0 + u * WAVE + lane;
                        bool sup = false;
                        int f = 0, sec = 0;
                        if (nearw[u]) {
                            int L, R;
                            overhangs(bs[u], (int)(bf[u] >> 31), mb[u], L, R);
                            if (type == -1) { sup = R > P.theta; f = av[u].x; sec = L; }
                            else { sup = L > P.theta; f = -av[u].y; sec = R; }
                        }
                        const unsigned long long bal = __ballot(sup);
                        if (sup) {
                            fmin_w = min(fmin_w, f);
                            fmax_w = max(fmax_w, f);
                            const int slot = slot0 + __popcll(bal & lmask);
                            if (slot < CAP) {
                                S.sF[slot] = f;
                                S.sS[slot] = sec;
                                S.sK[slot] = (unsigned short)(k - s);
                                if (slot < HC_SMALL) S.sL[slot] = av[u].y - av[u].x + bs[u].y - bs[u].x;
                            }
                        }
                        slot0 += __popcll(bal);
                    }
                }
                if (!LEAN && k_lo >= k_hi) {   // an empty slice still clears its share of the bins
                    int4* z23 = reinterpret_cast<int4*>(bin23);
                    int4* z0 = reinterpret_cast<int4*>(bin0);
                    for (int b = tid; b < SF_BINS / 4; b += BLOCK) z23[b] = make_int4(0, 0, 0, 0);
                    for (int b = tid; b < SF_BINS / 8; b += BLOCK) z0[b] = make_int4(0, 0, 0, 0);
                }
                fmin_w = -wave_max(-fmin_w);
                fmax_w = wave_max(fmax_w);
                if (lane == 0 && fmin_w != INT_MAX) { atomicMin(&S.f0, fmin_w); atomicMax(&S.fmax, fmax_w); }
            }
            __syncthreads();
#ifdef HINGE_TIMING
            const unsigned long long tm1 = wall_clock64();
            if (tid == 0 && dbg) { atomicAdd(&dbg[8], (unsigned)(tm1 - tm0)); atomicAdd(&dbg[9], (unsigned)n); atomicAdd(&dbg[10], 1u); }
#endif
            const int sup = S.cnt;
            // ---- decide the path (block-uniform) ----------------------------------------------------
            int action;   // 0: result 0, 1: resolve in LDS, 2: k_hinge_exact, 3: result 1 without sorting
            if (sup <= P.sup) action = 0;   // needs support >= SUP to be scanned and > SUP to be emitted
            else if (force_exact == 0 && P.unb >= 0 && S.near_end > P.unb) {
                // Sorted by the other end, the supporters with f - m0 < BIN form a prefix (the test is monotone
                // in f) and every one of them takes the scan's first branch: to_end reaches UNB + 1 at element
                // UNB and nothing can stop the scan earlier (considered == to_end <= UNB until then).  The hinge
                // is unbridged whatever the order inside that prefix is (filter.cpp:920-931 / 1019-1030).
                action = 3;
            }
            else if (sup > CAP || force_exact == 1) action = 2;
            else action = 1;
            bool need_order = false;
            if (!LEAN && action == 1 && force_exact == 0 && sup <= CAP && !ties_known) {
                // ---- sort-free evaluation of the scan (filter.cpp:932-963 / 1031-1062) -----------------
                // Past the first-branch prefix (c1 = near_end <= UNB elements) the scan walks the remaining
                // supporters by ascending f.  With cat = 2 (sec < TH), 3 (sec > TH), 0 (sec == TH):
                //   a cat-2 element stops it "unbridged" once considered > UNB and f - f[0] > BIN, where
                //     considered = c1 + #(cat 2/3 elements up to and including it);
                //   a cat-3 element stops it "bridged" when 1 + #(following elements with f' - f < BIN) > PIL.
                // Take the group G of elements with one value f: g2/g3/g0 members per category, g = |G|,
                // before = #(cat 2/3 elements with smaller f), W = #(f < f' < f + BIN), far = f - f[0] > BIN.
                // Whatever order std::sort leaves INSIDE the group:
                //   U_can  (an unbridged stop is possible)  = far, g2 >= 1, c1 + before + g2 + g3 > UNB
                //   B_can  (a bridged stop is possible)     = g3 >= 1, g + W > PIL        (a cat-3 member first)
                //   U_must (unbridged stop in every order)  = far, c1 + before + g2 > UNB (the last cat-2 member
                //                                             has seen all g2 of them), and not B_can
                //   B_must (bridged stop in every order)    = g3 >= 1, g3 + W > PIL (the first cat-3 member has at
                //                                             least the other g3 - 1 behind it), and not U_can
                // The scan's outcome is decided by the first group that stops it, so with F* = smallest f of a
                // group with that property:  no U_can anywhere -> bridged;  F(B_must) < F(U_can) -> bridged;
                // F(U_must) < F(B_can) -> unbridged.  Only otherwise does the tie order matter (exact replay).
                // Groups are found by binning f - f[0] at 1 bp (SF_BINS bins in LDS scratch that is idle at this
                // point); `before` and W are differences of prefix sums over the bins.  O(sup + SF_BINS).
                // Supporters further than SF_BINS bp from f[0] are rare; such a list takes the exact replay.
                const int f0 = S.f0, c1 = S.near_end;
                // only the bins the supporters actually span (rounded to the workgroup size) are scanned
                const long long span = (long long)S.fmax - f0 + 1;
                const int nb = span > SF_BINS ? SF_BINS : (int)((span + BLOCK - 1) / BLOCK) * BLOCK;
                if (span > SF_BINS && tid == 0) S.sf_over = 1;
                for (int t = tid; t < sup; t += BLOCK) {
                    const int ft = S.sF[t];
                    if (ft - m0 < P.bin_len) continue;   // first-branch prefix: never counted in `before` or W
                    const int rel = ft - f0;
                    if (rel >= nb) continue;             // only when sf_over
                    const int st = S.sS[t];
                    if (st < P.theta) atomicAdd(&bin23[rel], 1);
                    else if (st > P.theta) atomicAdd(&bin23[rel], 1 << 16);
                    else atomicAdd(reinterpret_cast<unsigned*>(&bin0[rel & ~1]), (rel & 1) ? (1u << 16) : 1u);
                }
                __syncthreads();
#ifdef HINGE_TIMING
                const unsigned long long tm2 = wall_clock64();
                if (tid == 0 && dbg) atomicAdd(&dbg[13], (unsigned)(tm2 - tm1));
#endif
                if (!S.sf_over) {
                    // workgroup inclusive scan: every wavefront owns nb/4 consecutive bins and walks them 64 at a time
                    // (stride-1 LDS accesses, DPP scan, carry in a register); the three cross-wave offsets are added at use
                    const int QW = nb / WAVES_PER_BLOCK;   // multiple of 64
                    {
                        int run = 0;   // packed carry: low 16 bits prefix of g2 + g3, high 16 bits prefix of g (both <= sup <= CAP)
                        for (int b = wib * QW + lane; b < (wib + 1) * QW; b += WAVE) {
                            const int v = bin23[b];
                            const int c23 = (v & 0xffff) + (v >> 16);
                            const int pk = wave_incl_scan(c23 | ((c23 + bin0[b]) << 16)) + run;
                            p23[b] = (unsigned short)(pk & 0xffff);
                            pall[b] = (unsigned short)((unsigned)pk >> 16);
                            run = wave_last(pk);
                        }
                        if (lane == 0) { S.wtot[0][wib] = run & 0xffff; S.wtot[1][wib] = (int)((unsigned)run >> 16); }
                    }
                    __syncthreads();
#ifdef HINGE_TIMING
                    if (tid == 0 && dbg) atomicAdd(&dbg[14], (unsigned)(wall_clock64() - tm2));
#endif
                    const int o23_1 = S.wtot[0][0], o23_2 = o23_1 + S.wtot[0][1], o23_3 = o23_2 + S.wtot[0][2];
                    const int oal_1 = S.wtot[1][0], oal_2 = oal_1 + S.wtot[1][1], oal_3 = oal_2 + S.wtot[1][2];
                    auto full23 = [&](int x) { return (int)p23[x] + (x >= 3 * QW ? o23_3 : x >= 2 * QW ? o23_2 : x >= QW ? o23_1 : 0); };
                    auto fullall = [&](int x) { return (int)pall[x] + (x >= 3 * QW ? oal_3 : x >= 2 * QW ? oal_2 : x >= QW ? oal_1 : 0); };
                    int m_ucan = INT_MAX, m_bcan = INT_MAX, m_umust = INT_MAX, m_bmust = INT_MAX;   // per-lane minima, one LDS atomic per wave
                    // every supporter evaluates ITS group (members of one group compute the same thing; the minima do not care):
                    // sup / 256 iterations instead of nb / 256
                    for (int t = tid; t < sup; t += BLOCK) {
                        const int ft = S.sF[t];
                        if (ft - m0 < P.bin_len) continue;   // first-branch prefix: not a group of the walk
                        const int b = ft - f0;
                        const int v = bin23[b];
                        const int g2 = v & 0xffff, g3 = v >> 16;
                        const int g = g2 + g3 + bin0[b];
                        const int before = full23(b) - (g2 + g3);
                        const int W = fullall(min(b + P.bin_len - 1, nb - 1)) - fullall(b);
                        const bool far = b > P.bin_len;          // f - f[0] > BIN
                        const bool ucan = far && g2 >= 1 && (c1 + before + g2 + g3 > P.unb);
                        const bool bcan = g3 >= 1 && (g + W > P.pil);
                        const bool umust = far && g2 >= 1 && (c1 + before + g2 > P.unb) && !bcan;
                        const bool bmust = g3 >= 1 && (g3 + W > P.pil) && !ucan;
                        if (ucan) m_ucan = min(m_ucan, b);
                        if (bcan) m_bcan = min(m_bcan, b);
                        if (umust) m_umust = min(m_umust, b);
                        if (bmust) m_bmust = min(m_bmust, b);
                    }
                    m_ucan = -wave_max(-m_ucan); m_bcan = -wave_max(-m_bcan); m_umust = -wave_max(-m_umust); m_bmust = -wave_max(-m_bmust);
                    if (lane == 0) {
                        if (m_ucan != INT_MAX) atomicMin(&S.ev_ucan, m_ucan);
                        if (m_bcan != INT_MAX) atomicMin(&S.ev_bcan, m_bcan);
                        if (m_umust != INT_MAX) atomicMin(&S.ev_umust, m_umust);
                        if (m_bmust != INT_MAX) atomicMin(&S.ev_bmust, m_bmust);
                    }
                    __syncthreads();
                    if (S.ev_ucan == INT_MAX) action = 0;
                    else if (S.ev_bmust < S.ev_ucan) action = 0;
                    else if (S.ev_umust < S.ev_bcan) action = 3;
                }
            }
            if (tid == 0 && dbg) { atomicAdd(&dbg[action], 1u); atomicMax(&dbg[6], (unsigned)sup); }
#ifdef HINGE_TIMING
            if (tid == 0 && dbg) { atomicAdd(&dbg[11], (unsigned)(wall_clock64() - tm1)); atomicAdd(&dbg[12], (unsigned)sup); }
#endif
            if (action == 1) {
                need_order = (force_exact == 2) || (sup > HC_SMALL);
                if (!need_order) {
                    // a length tie matters if the two could swap places in the final order: always when the
                    // supporter sort is an introsort (sup > 16), only for equal f when it is a stable
                    // insertion sort; never when the pile-up sort itself is stable (n <= 16)
                    if (wib == 0 && n > 16) {
                        bool tie = false;
                        for (int t = lane; t < sup; t += WAVE) {
                            const int Lt = S.sL[t], ft = S.sF[t];
                            for (int u = 0; u < sup; u++)
                                if (u != t && S.sL[u] == Lt && (sup > 16 || S.sF[u] == ft)) tie = true;
                        }
                        if (__any(tie) && lane == 0) S.need_order = 1;
                    }
                    __syncthreads();
                    need_order = S.need_order != 0;
                }
                if (need_order && n > CAP) action = 2;
            }
            if (action == 0 || action == 3) {
                if (tid == 0) hinge_flag[item.slot] = action == 3 ? 1 : 0;
                continue;
            }
            if (action == 2) {
                if (tid == 0) {
                    const unsigned q = atomicAdd(exact_count, 1u);
                    if (q < exact_cap) exact_queue[q] = make_int2(i, a);
                    else atomicOr(status, ST_QUEUE_CAP);
                }
                continue;
            }
            // ---- exact pile-up order, once per read ---------------------------------------------
#ifdef HINGE_TIMING
            const unsigned long long tr0 = wall_clock64();
#endif
            if (need_order && !order_valid) {
                order_valid = true;
                if (tid == 0 && dbg) atomicAdd(&dbg[4], 1u);
                bool sorted = true;
                if constexpr (LEAN) {
                    sorted = block_std_sort_desc_keys(S.ws, n, tid, [&](int p) {
                        const int2 av = a_span[s + p];
                        const int2 bs = b_span[s + p];
                        return av.y - av.x + bs.y - bs.x;              // compare_overlap key
                    });
                } else {
                    for (int64_t k = s + tid; k < e; k += BLOCK) {
                        const int2 av = a_span[k];
                        const int2 bs = b_span[k];
                        S.ws.key[k - s] = av.y - av.x + bs.y - bs.x;   // compare_overlap key
                    }
                    __syncthreads();
                    block_std_sort_desc(S.ws, n, tid);
                }
                if (!sorted) {   // (block-uniform; keys spanning 2^20: the serial exact kernel takes the annotation)
                    order_valid = false;
                    if (tid == 0) {
                        const unsigned q = atomicAdd(exact_count, 1u);
                        if (q < exact_cap) exact_queue[q] = make_int2(i, a);
                        else atomicOr(status, ST_QUEUE_CAP);
                    }
                    continue;
                }
                for (int p = tid; p < n; p += BLOCK) S.ppos[p] = S.ws.pl[p];
                __syncthreads();
            }
            if (tid == 0 && dbg && need_order) atomicAdd(&dbg[5], 1u);
#ifdef HINGE_TIMING
            const unsigned long long tr1 = wall_clock64();
#endif
            // ---- supporters in pile-up order -> wF / wS (one wave); LEAN: their ORDER -> sK (no copies) ----
            if (wib == 0) {
                if (need_order) {
                    unsigned short* slot_of = S.ws.seglo;   // scratch: pile-up position -> supporter + 1
                    for (int p = lane; p < n; p += WAVE) slot_of[p] = 0;
                    for (int t = lane; t < sup; t += WAVE) slot_of[S.ppos[S.sK[t]]] = (unsigned short)(t + 1);
                    int r = 0;
                    for (int base = 0; base < n; base += WAVE) {
                        const int p = base + lane;
                        const int v = p < n ? slot_of[p] : 0;
                        const unsigned long long bal = __ballot(v != 0);
                        if (v) {
                            const int dst = r + __popcll(bal & lmask);
                            if constexpr (LEAN) S.sK[dst] = (unsigned short)(v - 1);      // (sK was read into slot_of above: free)
                            else { S.wF[dst] = S.sF[v - 1]; S.wS[dst] = S.sS[v - 1]; }
                        }
                        r += __popcll(bal);
                    }
                } else {
                    int rank_of = 0;
                    for (int t = lane; t < sup; t += WAVE) {       // (sup <= HC_SMALL = 64: one pass)
                        const int Lt = S.sL[t];
                        int rank = 0;
                        for (int u = 0; u < sup; u++) {
                            const int Lu = S.sL[u];
                            rank += (Lu > Lt) || (Lu == Lt && u < t);   // .las order breaks (harmless) ties
                        }
                        rank_of = rank;
                        if constexpr (!LEAN) { S.wF[rank] = S.sF[t]; S.wS[rank] = S.sS[t]; }
                    }
                    if constexpr (LEAN) { if (lane < sup) S.sK[rank_of] = (unsigned short)lane; }
                }
            }
            __syncthreads();
#ifdef HINGE_TIMING
            const unsigned long long tr2 = wall_clock64();
#endif
            // ---- std::sort(pairAscend / pairDescend): ascending f == descending -f --------------------
            if constexpr (LEAN) {
                // the list in pile-up order is sK's order of sF: its keys through the order, the sorted list as an order again (seglo)
                const bool sorted = block_std_sort_desc_keys(S.ws, sup, tid, [&](int r) { return -S.sF[S.sK[r]]; });
                if (!sorted) {
                    if (tid == 0) {
                        const unsigned q = atomicAdd(exact_count, 1u);
                        if (q < exact_cap) exact_queue[q] = make_int2(i, a);
                        else atomicOr(status, ST_QUEUE_CAP);
                    }
                    continue;
                }
                for (int r = tid; r < sup; r += BLOCK) S.ws.seglo[S.ws.pl[r]] = S.sK[r];
                __syncthreads();
            } else {
                for (int t = tid; t < sup; t += BLOCK) S.ws.key[t] = -S.wF[t];
                __syncthreads();
                block_std_sort_desc(S.ws, sup, tid);
                for (int t = tid; t < sup; t += BLOCK) {
                    const int p = S.ws.pl[t];
                    S.sF[p] = S.wF[t];
                    S.sS[p] = S.wS[t];
                }
                __syncthreads();
            }
#ifdef HINGE_TIMING
            const unsigned long long tr3 = wall_clock64();
#endif
            if (tid == 0) {
                const int r = LEAN ? hinge_scan(S.sF, S.sS, S.ws.seglo, sup, m0, P.bin_len, P.theta, P.unb, P.pil)
                                   : hinge_scan(S.sF, S.sS, (const unsigned short*)nullptr, sup, m0, P.bin_len, P.theta, P.unb, P.pil);
                hinge_flag[item.slot] = r == 0 ? 1 : 0;   // emit iff not bridged (support > SUP holds)
#ifdef HINGE_TIMING
                if (dbg) {
                    const unsigned long long tr4 = wall_clock64();
                    atomicAdd(&dbg[16], (unsigned)(tr1 - tr0)); atomicAdd(&dbg[17], (unsigned)(tr2 - tr1)); atomicAdd(&dbg[18], (unsigned)(tr3 - tr2));
                    atomicAdd(&dbg[19], (unsigned)(tr4 - tr3)); atomicAdd(&dbg[20], 1u); atomicAdd(&dbg[21], (unsigned)n); atomicAdd(&dbg[22], (unsigned)sup);
                    atomicMax(&dbg[23], (unsigned)(tr4 - tm0));
                }
#endif
            }
        }
        }   // the read's chain
    }
}

}  // namespace hinge
