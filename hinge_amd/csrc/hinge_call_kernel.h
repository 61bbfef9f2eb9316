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
constexpr int PRE_MAXA = 4;     // annotations covered by the count-only sweep
constexpr int SF_BINS = 2 * PO_CAP;   // 1-bp bins of f - f[0] for the sort-free scan evaluation

struct HingeCallLds {
    WaveSortLds ws;
    unsigned short ppos[PO_CAP];   // position of every overlap of the read in the sorted pile-up
    alignas(16) int sF[PO_CAP];    // supporters in .las order: other end in scan-ascending form ...
    alignas(16) int sS[PO_CAP];    // ... and the overhang on the far side; reused for the sorted lists
    unsigned short sK[PO_CAP];     // ... and the local overlap index
    int sL[HC_SMALL];              // length sums of the first HC_SMALL supporters
    int wF[PO_CAP];                // supporters in pile-up order
    int wS[PO_CAP];
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
// K3a: count-only sweep, one wavefront per work-list read, no LDS.  For every annotation: support and the
// number of supporters that take the scan's first branch.  These two numbers decide most annotations
// (support <= SUP: no hinge;  first-branch count > UNB: unbridged whatever the order, filter.cpp:920-931).
// Undecided annotations get hinge_flag = 2 and their read goes to the heavy list for k_hinge_call.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_hinge_count(FilterDev P, const int64_t* __restrict__ row_ptr, const int2* __restrict__ a_span,
                                                       const int2* __restrict__ b_span, const unsigned* __restrict__ b_flag,
                                                       const int2* __restrict__ mask, const int2* __restrict__ anno_buf,
                                                       const unsigned* __restrict__ anno_off, const int* __restrict__ anno_cnt,
                                                       const int* __restrict__ work_list, const unsigned* __restrict__ counters,
                                                       unsigned char* __restrict__ hinge_flag, int* __restrict__ heavy_list,
                                                       unsigned* __restrict__ heavy_count, int force_exact, unsigned* __restrict__ dbg) {
    const int lane = lane_id();
    const unsigned wave = __builtin_amdgcn_readfirstlane((blockIdx.x * BLOCK + threadIdx.x) >> 6);
    const unsigned nwaves = (gridDim.x * BLOCK) >> 6;
    const unsigned nwork = counters[1];
    for (unsigned w = wave; w < nwork; w += nwaves) {
        const int i = work_list[w];
        const int64_t s = row_ptr[i], e = row_ptr[i + 1];
        const int2 mk = mask[i];
        const unsigned off = anno_off[i];
        const int cnt = anno_cnt[i];
        bool heavy = false;
        for (int a0 = 0; a0 < cnt; a0 += PRE_MAXA) {
            const int na = min(PRE_MAXA, cnt - a0);
            int apos[PRE_MAXA], atype[PRE_MAXA], csup[PRE_MAXA], cnear[PRE_MAXA];
#pragma unroll
            for (int a = 0; a < PRE_MAXA; a++) {
                const int2 an = a < na ? anno_buf[off + a0 + a] : make_int2(0, 0);
                apos[a] = an.x; atype[a] = an.y; csup[a] = 0; cnear[a] = 0;
            }
            for (int64_t k = s + lane; k < e; k += WAVE) {
                const int2 av = a_span[k];
                bool loaded = false;
                int L = 0, R = 0;
#pragma unroll
                for (int a = 0; a < PRE_MAXA; a++) {
                    if (a >= na) break;
                    const int c = atype[a] == -1 ? av.y : av.x;
                    if ((c > apos[a] - P.tol) && (c < apos[a] + P.tol)) {
                        if (!loaded) {
                            const unsigned bf = b_flag[k];
                            const int2 bs = b_span[k];
                            const int2 mb = mask[bf & 0x7fffffffu];
                            overhangs(bs, (int)(bf >> 31), mb, L, R);
                            loaded = true;
                        }
                        const bool sup = atype[a] == -1 ? (R > P.theta) : (L > P.theta);
                        if (sup) {
                            csup[a]++;
                            const int f = atype[a] == -1 ? av.x : -av.y;
                            const int m0 = atype[a] == -1 ? mk.x : -mk.y;
                            cnear[a] += (f - m0 < P.bin_len);
                        }
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < PRE_MAXA; a++) {
                if (a >= na) break;
                const int psup = wave_sum(csup[a]), pnear = wave_sum(cnear[a]);
                int quick = 2;
                if (force_exact == 0) {
                    if (psup <= P.sup) quick = 0;                      // needs support >= SUP to be scanned and > SUP to be emitted
                    else if (P.unb >= 0 && pnear > P.unb) quick = 1;   // the first UNB+1 sorted supporters all take branch 1
                }
                if (quick == 2) heavy = true;
                if (lane == 0) { hinge_flag[off + a0 + a] = (unsigned char)quick; if (dbg && quick != 2) atomicAdd(&dbg[quick ? 3 : 0], 1u); }
            }
        }
        if (heavy && lane == 0) heavy_list[atomicAdd(heavy_count, 1u)] = i;
    }
}

__global__ __launch_bounds__(BLOCK) void k_hinge_call(FilterDev P, const int64_t* __restrict__ row_ptr, const int2* __restrict__ a_span,
                                                      const int2* __restrict__ b_span, const unsigned* __restrict__ b_flag,
                                                      const int2* __restrict__ mask, const int2* __restrict__ anno_buf,
                                                      const unsigned* __restrict__ anno_off, const int* __restrict__ anno_cnt,
                                                      const int* __restrict__ heavy_list, const unsigned* __restrict__ heavy_count,
                                                      unsigned char* __restrict__ hinge_flag, int2* __restrict__ exact_queue,
                                                      unsigned* __restrict__ exact_count, unsigned exact_cap, int force_exact,
                                                      int* __restrict__ status, unsigned* __restrict__ work_next,
                                                      unsigned* __restrict__ dbg) {
    __shared__ HingeCallLds S;
    const int tid = threadIdx.x;
    const int lane = lane_id();
    const int wib = tid >> 6;
    const unsigned long long lmask = (1ull << lane) - 1ull;
    const unsigned nwork = *heavy_count;
    while (true) {
        // dynamic work distribution: reads differ by orders of magnitude in cost
        __syncthreads();
        if (tid == 0) S.next_item = atomicAdd(work_next, 1u);
        __syncthreads();
        const unsigned w = S.next_item;
        if (w >= nwork) break;
        const int i = heavy_list[w];
        const int64_t s = row_ptr[i], e = row_ptr[i + 1];
        const int n = (int)(e - s);
        const int2 mk = mask[i];
        const unsigned off = anno_off[i];
        const int cnt = anno_cnt[i];
        bool order_ready = false;   // block-uniform
        for (int a = 0; a < cnt; a++) {
            if (hinge_flag[off + a] != 2 && force_exact == 0) continue;   // decided by k_hinge_count (block-uniform)
            const int2 an = anno_buf[off + a];
            const int pos = an.x, type = an.y;
            __syncthreads();
            if (tid == 0) { S.cnt = 0; S.need_order = 0; S.near_end = 0; }
            const int m0 = type == -1 ? mk.x : -mk.y;
            __syncthreads();
            // ---- gather -----------------------------------------------------------------------------
            int par = 0;
            for (int64_t k0 = s; k0 < e; k0 += BLOCK, par ^= 1) {
                const int64_t k = k0 + tid;
                bool sup = false;
                int f = 0, sec = 0, Lsum = 0;
                if (k < e) {
                    const int2 av = a_span[k];
                    const int c = type == -1 ? av.y : av.x;
                    if ((c > pos - P.tol) && (c < pos + P.tol)) {
                        const unsigned bf = b_flag[k];
                        const int2 bs = b_span[k];
                        const int2 mb = mask[bf & 0x7fffffffu];
                        int L, R;
                        overhangs(bs, (int)(bf >> 31), mb, L, R);
                        if (type == -1) { sup = R > P.theta; f = av.x; sec = L; }
                        else { sup = L > P.theta; f = -av.y; sec = R; }
                        Lsum = av.y - av.x + bs.y - bs.x;
                    }
                }
                const unsigned long long bal = __ballot(sup);
                const unsigned long long baln = __ballot(sup && (f - m0 < P.bin_len));
                if (lane == 0) {
                    S.wcnt[par][wib] = __popcll(bal);
                    // supporters whose other end lies within HINGE_BIN_LENGTH of the mask end (scan branch 1)
                    const int nn = __popcll(baln);
                    if (nn) atomicAdd(&S.near_end, nn);
                }
                __syncthreads();
                if (sup) {
                    int slot = S.cnt + __popcll(bal & lmask);
                    for (int ww = 0; ww < wib; ww++) slot += S.wcnt[par][ww];
                    if (slot < PO_CAP) {
                        S.sF[slot] = f;
                        S.sS[slot] = sec;
                        S.sK[slot] = (unsigned short)(k - s);
                        if (slot < HC_SMALL) S.sL[slot] = Lsum;
                    }
                }
                __syncthreads();
                if (tid == 0) S.cnt += S.wcnt[par][0] + S.wcnt[par][1] + S.wcnt[par][2] + S.wcnt[par][3];
                // the next chunk writes the other wcnt buffer and syncs before cnt is read again
            }
            __syncthreads();
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
            else if (sup > PO_CAP || force_exact == 1) action = 2;
            else action = 1;
            bool need_order = false;
            if (action == 1 && force_exact == 0 && sup <= PO_CAP) {
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
                if (tid == 0) { S.ev_ucan = INT_MAX; S.ev_bcan = INT_MAX; S.ev_umust = INT_MAX; S.ev_bmust = INT_MAX; S.f0 = INT_MAX; S.fmax = INT_MIN; S.sf_over = 0; }
                int* bin23 = S.wF;                                   // [SF_BINS] g2 | g3 << 16   (wF..wS are contiguous)
                unsigned short* bin0 = reinterpret_cast<unsigned short*>(S.ws.key);    // [SF_BINS] g0
                unsigned short* p23 = S.ws.pl;                       // [SF_BINS] inclusive prefix of g2 + g3 (pl..pr)
                unsigned short* pall = S.ws.seglo;                   // [SF_BINS] inclusive prefix of g       (seglo..seghi)
                __syncthreads();
                {
                    int fm = INT_MAX, fx = INT_MIN;
                    for (int t = tid; t < sup; t += BLOCK) { const int v = S.sF[t]; fm = min(fm, v); fx = max(fx, v); }
                    fm = -wave_max(-fm);
                    fx = wave_max(fx);
                    if (lane == 0 && fm != INT_MAX) { atomicMin(&S.f0, fm); atomicMax(&S.fmax, fx); }
                }
                __syncthreads();
                const int f0 = S.f0, c1 = S.near_end;
                // only the bins the supporters actually span (rounded to the workgroup size) are touched
                const long long span = (long long)S.fmax - f0 + 1;
                const int nb = span > SF_BINS ? SF_BINS : (int)((span + BLOCK - 1) / BLOCK) * BLOCK;
                if (span > SF_BINS && tid == 0) S.sf_over = 1;
                for (int b = tid; b < nb; b += BLOCK) { bin23[b] = 0; bin0[b] = 0; }
                __syncthreads();
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
                if (!S.sf_over) {
                    // workgroup inclusive scan: each thread owns nb/BLOCK consecutive bins
                    const int PER = nb / BLOCK;
                    const int b0 = tid * PER;
                    int s23 = 0, sall = 0;
                    for (int k = 0; k < PER; k++) {
                        const int v = bin23[b0 + k];
                        const int c23 = (v & 0xffff) + (v >> 16);
                        s23 += c23;
                        sall += c23 + bin0[b0 + k];
                    }
                    int i23 = wave_incl_scan(s23), iall = wave_incl_scan(sall);
                    if (lane == WAVE - 1) { S.wtot[0][wib] = i23; S.wtot[1][wib] = iall; }
                    __syncthreads();
                    int off23 = 0, offall = 0;
                    for (int ww = 0; ww < wib; ww++) { off23 += S.wtot[0][ww]; offall += S.wtot[1][ww]; }
                    int run23 = off23 + i23 - s23, runall = offall + iall - sall;
                    for (int k = 0; k < PER; k++) {
                        const int v = bin23[b0 + k];
                        const int c23 = (v & 0xffff) + (v >> 16);
                        run23 += c23;
                        runall += c23 + bin0[b0 + k];
                        p23[b0 + k] = (unsigned short)run23;
                        pall[b0 + k] = (unsigned short)runall;
                    }
                    __syncthreads();
                    for (int b = tid; b < nb; b += BLOCK) {
                        const int v = bin23[b];
                        const int g2 = v & 0xffff, g3 = v >> 16;
                        const int g = g2 + g3 + bin0[b];
                        if (g == 0) continue;
                        const int before = (int)p23[b] - (g2 + g3);
                        const int W = (int)pall[min(b + P.bin_len - 1, nb - 1)] - (int)pall[b];
                        const bool far = b > P.bin_len;          // f - f[0] > BIN
                        const bool ucan = far && g2 >= 1 && (c1 + before + g2 + g3 > P.unb);
                        const bool bcan = g3 >= 1 && (g + W > P.pil);
                        const bool umust = far && g2 >= 1 && (c1 + before + g2 > P.unb) && !bcan;
                        const bool bmust = g3 >= 1 && (g3 + W > P.pil) && !ucan;
                        if (ucan) atomicMin(&S.ev_ucan, b);
                        if (bcan) atomicMin(&S.ev_bcan, b);
                        if (umust) atomicMin(&S.ev_umust, b);
                        if (bmust) atomicMin(&S.ev_bmust, b);
                    }
                    __syncthreads();
                    if (S.ev_ucan == INT_MAX) action = 0;
                    else if (S.ev_bmust < S.ev_ucan) action = 0;
                    else if (S.ev_umust < S.ev_bcan) action = 3;
                }
            }
            if (tid == 0 && dbg) { atomicAdd(&dbg[action], 1u); atomicMax(&dbg[6], (unsigned)sup); atomicMax(&dbg[7], (unsigned)cnt); }
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
                if (need_order && n > PO_CAP) action = 2;
            }
            if (action == 0 || action == 3) {
                if (tid == 0) hinge_flag[off + a] = action == 3 ? 1 : 0;
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
            if (need_order && !order_ready) {
                if (tid == 0 && dbg) atomicAdd(&dbg[4], 1u);
                for (int64_t k = s + tid; k < e; k += BLOCK) {
                    const int2 av = a_span[k];
                    const int2 bs = b_span[k];
                    S.ws.key[k - s] = av.y - av.x + bs.y - bs.x;   // compare_overlap key
                }
                __syncthreads();
                block_std_sort_desc(S.ws, n, tid);
                for (int p = tid; p < n; p += BLOCK) S.ppos[p] = S.ws.pl[p];
                __syncthreads();
                order_ready = true;
            }
            if (tid == 0 && dbg && need_order) atomicAdd(&dbg[5], 1u);
            // ---- supporters in pile-up order -> wF / wS (one wave) --------------------------------
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
                            S.wF[dst] = S.sF[v - 1];
                            S.wS[dst] = S.sS[v - 1];
                        }
                        r += __popcll(bal);
                    }
                } else {
                    for (int t = lane; t < sup; t += WAVE) {
                        const int Lt = S.sL[t];
                        int rank = 0;
                        for (int u = 0; u < sup; u++) {
                            const int Lu = S.sL[u];
                            rank += (Lu > Lt) || (Lu == Lt && u < t);   // .las order breaks (harmless) ties
                        }
                        S.wF[rank] = S.sF[t];
                        S.wS[rank] = S.sS[t];
                    }
                }
            }
            __syncthreads();
            // ---- std::sort(pairAscend / pairDescend): ascending f == descending -f --------------------
            for (int t = tid; t < sup; t += BLOCK) S.ws.key[t] = -S.wF[t];
            __syncthreads();
            block_std_sort_desc(S.ws, sup, tid);
            for (int t = tid; t < sup; t += BLOCK) {
                const int p = S.ws.pl[t];
                S.sF[p] = S.wF[t];
                S.sS[p] = S.wS[t];
            }
            __syncthreads();
            if (tid == 0) {
                const int r = hinge_scan(S.sF, S.sS, nullptr, sup, m0, P.bin_len, P.theta, P.unb, P.pil);
                hinge_flag[off + a] = r == 0 ? 1 : 0;   // emit iff not bridged (support > SUP holds)
            }
        }
    }
}

}  // namespace hinge
