// HIP kernels of the overlap trim / classify step shared by `hinge maximal` and `hinge layout`:
//   ProcessAlignment            maximal/maximal.cpp:65-134  (== layout/hinging.cpp:78-147)
//   LOverlap::trim_overlap      lib/LAInterface.cpp:4552-4683
//   LOverlap::AddTypesAsymmetric lib/LAInterface.cpp:4721-4806
//   LOverlap::GetMatchingPosition lib/LAInterface.cpp:4498-4546
//
// One lane owns one overlap and walks its trace (see k_trim_classify).
// HBM-bound: 24 B of record + tlen bytes of trace per classified overlap, 40 B (or 1 B: the type) out.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "filter_kernels.h"

namespace hinge {

// MatchType numbering of src/include/LAInterface.h:30-33
enum : int { MT_FORWARD = 0, MT_BACKWARD = 1, MT_ACOVERB = 2, MT_BCOVERA = 3, MT_UNDEFINED = 4, MT_INTERNAL = 5, MT_NOT_ACTIVE = 6,
             MT_FORWARD_INTERNAL = 12, MT_BACKWARD_INTERNAL = 13 };

struct ClassifyOut {   // 40 bytes
    int eff_ab, eff_ae, eff_bb, eff_be;
    int type, active, weight, length;
    int start_idx, end_idx;
};

__device__ __forceinline__ int add_types_asymmetric(int A_left, int A_right, int B_left, int B_right, int maxo, int mino) {
    // (B_left / B_right already swapped for complemented overlaps)
    if ((max(A_left, A_right) < maxo) && (min(B_left, B_right) > mino)) return MT_BCOVERA;
    if ((max(B_left, B_right) < maxo) && (min(A_left, A_right) > mino)) return MT_ACOVERB;
    if (min(A_left, A_right) > maxo) return MT_INTERNAL;
    if (A_left <= maxo) {
        if ((B_right <= maxo) && (B_left >= maxo)) return MT_BACKWARD;
        if ((B_right >= maxo) && (B_left >= maxo)) return MT_BACKWARD_INTERNAL;
        return MT_UNDEFINED;
    }
    if (A_right <= maxo) {
        if ((B_left <= maxo) && (B_right >= maxo)) return MT_FORWARD;
        if ((B_left >= maxo) && (B_right >= maxo)) return MT_FORWARD_INTERNAL;
        return MT_UNDEFINED;
    }
    return MT_UNDEFINED;   // match_type_ keeps its initial value
}

// sel[j] = index (into the part's SoA arrays) of the j-th overlap to classify; a_of[j] = its A read.
// TB = bytes per trace element (1 for tspace <= 125, else 2).
// One LANE per overlap: the lane walks its trace front to back (B coordinate = running sum of the B advances), keeps the first
// point inside both masks and the last one.  ~12 instructions per trace point and lane; the 16-lane-row form this replaces
// (row-wide prefix sum, ballots and five cross-lane reads per 16 points) needed ~90 per point-row for 4 overlaps per wavefront
// and was bound by instruction issue at 7 % of the HBM roofline (5.0 ms for 24.7 M overlaps, 120 B each).
template <int TB>
__global__ __launch_bounds__(BLOCK) void k_trim_classify(int64_t n_sel, const int64_t* __restrict__ sel, const int* __restrict__ a_of,
                                                         const int2* __restrict__ a_span, const int2* __restrict__ b_span,
                                                         const unsigned* __restrict__ b_flag, const unsigned char* __restrict__ trace,
                                                         const int64_t* __restrict__ trace_off, const int* __restrict__ tlen,
                                                         const int2* __restrict__ eff /*[n_reads] effective_start/end*/, int aln_threshold,
                                                         int theta, int theta2, ClassifyOut* __restrict__ out,
                                                         unsigned char* __restrict__ type_out /*nullptr, or only the match type is wanted*/,
                                                         int trim /*0: PAF input, ProcessAlignment(trim = false): the match is taken as it is*/) {
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
    for (int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x; j < n_sel; j += stride) {
        const int64_t k = sel[j];
        const int2 av = a_span[k], bs = b_span[k];
        const unsigned bf = b_flag[k];
        const int comp = (int)(bf >> 31);
        const int2 ea = eff[a_of[j]], eb = eff[bf & 0x7fffffffu];
        const int tl = tlen[k];
        const unsigned char* __restrict__ tp = trace + trace_off[k];
        const int ninner = max(tl / 2 - 1, 0);
        const int np = trim ? ninner + 2 : 0;          // trace points incl. the two end points
        const int sign = 1 - 2 * comp;
        const int b_first = comp ? bs.y : bs.x;        // tp[0].second
        const int b_last = comp ? bs.x : bs.y;         // tp[np-1].second
        const int a_base = (av.x / 100) * 100;         // inner point i sits at a_base + 100*i (hard-coded 100, LAInterface.cpp:4581-4584)
        int start_idx = np, end_idx = 0;
        int s_a = 0, s_b = 0, e_a = 0, e_b = 0;        // coordinates of the first / last point inside both masks
        bool s_found = false, e_found = false;
        auto visit = [&](int i, int pa, int pb) {
            bool cs, ce;
            if (comp == 0) {
                cs = (pa >= ea.x) && (pb >= eb.x);
                ce = (pa <= ea.y) && (pb <= eb.y);
            } else {
                cs = (pa >= ea.x) && (pb <= eb.y);
                ce = (pa <= ea.y) && (pb >= eb.x);
            }
            if (cs && !s_found) { s_a = pa; s_b = pb; start_idx = i; s_found = true; }
            if (ce) { e_a = pa; e_b = pb; end_idx = i; e_found = true; }
        };
        if (np > 0) {
            visit(0, av.x, b_first);
            // the trace is read 8 bytes at a time (4 points of one byte pairs, 2 of two byte pairs; unaligned loads), two
            // loads in flight; the bytes of the last, partial group are fetched one by one so that nothing is read past the trace
            constexpr int PAIR = 2 * TB, CH = 8 / PAIR;
            const int tbytes_total = tl * TB;
            int pb_run = b_first;                      // B coordinate of the current inner point
#pragma unroll 2
            for (int base = 0; base < ninner; base += CH) {
                unsigned long long w = 0;
                if ((base + CH) * PAIR <= tbytes_total) {
                    __builtin_memcpy(&w, tp + (size_t)base * PAIR, 8);
                } else {
                    for (int q = 0; base * PAIR + q < tbytes_total && q < 8; q++) w |= (unsigned long long)tp[(size_t)base * PAIR + q] << (8 * q);
                }
#pragma unroll
                for (int q = 0; q < CH; q++) {
                    const int i = base + q + 1;
                    if (i > ninner) break;
                    const int adv = TB == 1 ? (int)((w >> (16 * q + 8)) & 0xffull) : (int)((w >> (32 * q + 16)) & 0xffffull);
                    pb_run += sign * adv;
                    visit(i, a_base + 100 * i, pb_run);
                }
            }
            visit(np - 1, av.y, b_last);
        }
        ClassifyOut o;
        o.eff_ab = av.x; o.eff_ae = av.y; o.eff_bb = bs.x; o.eff_be = bs.y;
        if (comp == 0) {
            if (s_found) { o.eff_ab = s_a; o.eff_bb = s_b; }
            if (e_found) { o.eff_ae = e_a; o.eff_be = e_b; }
        } else {
            if (s_found) { o.eff_ab = s_a; o.eff_be = s_b; }
            if (e_found) { o.eff_ae = e_a; o.eff_bb = e_b; }
        }
        bool active = trim ? !(start_idx >= end_idx) : true;   // without trimming match->active keeps its value (maximal.cpp:97-104)
        if (!trim) { start_idx = 0; end_idx = 0; }
        int type;
        if (((o.eff_be - o.eff_bb) < aln_threshold) || ((o.eff_ae - o.eff_ab) < aln_threshold) || !active) {
            active = false;
            type = MT_NOT_ACTIVE;
        } else {
            const int A_left = o.eff_ab - ea.x, A_right = ea.y - o.eff_ae;
            int B_left = o.eff_bb - eb.x, B_right = eb.y - o.eff_be;
            if (comp) { const int t = B_left; B_left = B_right; B_right = t; }
            type = add_types_asymmetric(A_left, A_right, B_left, B_right, theta, theta2);
        }
        o.type = type;
        o.active = active ? 1 : 0;
        o.weight = o.eff_ae - o.eff_ab + o.eff_be - o.eff_bb;
        o.length = av.y - av.x + bs.y - bs.x;
        o.start_idx = start_idx;
        o.end_idx = end_idx;
        if (type_out) type_out[j] = (unsigned char)o.type; else out[j] = o;
    }
}

// GetMatchingPosition for a list of (overlap, pos_A) queries: one thread each (tiny lists: hinges x matches).
template <int TB>
__global__ void k_matching_position(int64_t nq, const int64_t* __restrict__ q_ovl, const int* __restrict__ q_pos,
                                    const int2* __restrict__ a_span, const int2* __restrict__ b_span, const unsigned* __restrict__ b_flag,
                                    const unsigned char* __restrict__ trace, const int64_t* __restrict__ trace_off,
                                    const int* __restrict__ tlen, int* __restrict__ out) {
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = q_ovl[q];
        const int pos_A = q_pos[q];
        const int2 av = a_span[k], bs = b_span[k];
        const int comp = (int)(b_flag[k] >> 31);
        int res;
        if ((pos_A < av.x) || (pos_A > av.y)) { out[q] = -1; continue; }
        const int rev_sign = 1 - 2 * comp;
        int cur_A = av.x, next_A = av.x;
        int cur_B = comp ? bs.y : bs.x;
        const int tl = tlen[k];
        const int64_t toff = trace_off[k];
        bool done = false;
        res = -2;
        for (int j = 0; j < tl / 2 - 1; j++) {
            next_A = (cur_A % 100 != 0) ? (cur_A / 100 + 1) * 100 : cur_A + 100;
            if (next_A >= pos_A) { res = cur_B + pos_A - cur_A; done = true; break; }
            const int64_t p = toff + (int64_t)TB * (2 * j + 1);
            const int adv = TB == 1 ? (int)trace[p] : (int)(trace[p] | (trace[p + 1] << 8));
            cur_B = cur_B + rev_sign * adv;
            cur_A = next_A;
        }
        if (!done && cur_A < pos_A) res = cur_B + pos_A - cur_A;
        out[q] = res;
    }
}

}  // namespace hinge
