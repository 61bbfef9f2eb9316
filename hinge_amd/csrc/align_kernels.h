// HIP kernels of the overlap trim / classify step shared by `hinge maximal` and `hinge layout`:
//   ProcessAlignment            maximal/maximal.cpp:65-134  (== layout/hinging.cpp:78-147)
//   LOverlap::trim_overlap      lib/LAInterface.cpp:4552-4683
//   LOverlap::AddTypesAsymmetric lib/LAInterface.cpp:4721-4806
//   LOverlap::GetMatchingPosition lib/LAInterface.cpp:4498-4546
//
// One 16-lane DPP row owns one overlap (four overlaps per wavefront): the row streams the overlap's
// trace, 16 trace points per step, turns the B advances into B coordinates with a row-local prefix sum,
// evaluates "inside both masks" per point and finds the first / last such point with a ballot.
// HBM-bound: 24 B of record + tlen bytes of trace per classified overlap, 40 B out.
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

__device__ __forceinline__ int row_incl_scan(int v) {   // inclusive + scan inside each 16-lane DPP row
    v += dpp_or_old<0x111, 0xf>(0, v);
    v += dpp_or_old<0x112, 0xf>(0, v);
    v += dpp_or_old<0x114, 0xf>(0, v);
    v += dpp_or_old<0x118, 0xf>(0, v);
    return v;
}

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
template <int TB>
__global__ __launch_bounds__(BLOCK) void k_trim_classify(int64_t n_sel, const int64_t* __restrict__ sel, const int* __restrict__ a_of,
                                                         const int2* __restrict__ a_span, const int2* __restrict__ b_span,
                                                         const unsigned* __restrict__ b_flag, const unsigned char* __restrict__ trace,
                                                         const int64_t* __restrict__ trace_off, const int* __restrict__ tlen,
                                                         const int2* __restrict__ eff /*[n_reads] effective_start/end*/, int aln_threshold,
                                                         int theta, int theta2, ClassifyOut* __restrict__ out,
                                                         unsigned char* __restrict__ type_out /*nullptr, or only the match type is wanted*/,
                                                         int trim /*0: PAF input, ProcessAlignment(trim = false): the match is taken as it is*/) {
    const int lane = lane_id();
    const int r = lane & 15;              // lane inside the row
    const int row = lane >> 4;            // 0..3
    const int64_t rows_total = (int64_t)gridDim.x * (BLOCK / 16);
    const int64_t my_row = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 4;
    // all four rows of a wave iterate together; a row without work idles with j >= n_sel
    for (int64_t j0 = (my_row - row); j0 < n_sel; j0 += rows_total) {   // j0 = first row's item of this wave
        const int64_t j = j0 + row;
        const bool live = j < n_sel;
        int2 av = make_int2(0, 0), bs = make_int2(0, 0), ea = make_int2(0, 0), eb = make_int2(0, 0);
        int comp = 0, tl = 0;
        int64_t toff = 0;
        if (live) {
            const int64_t k = sel[j];
            av = a_span[k];
            bs = b_span[k];
            const unsigned bf = b_flag[k];
            comp = (int)(bf >> 31);
            ea = eff[a_of[j]];
            eb = eff[bf & 0x7fffffffu];
            tl = tlen[k];
            toff = trace_off[k];
        }
        const int ninner = max(tl / 2 - 1, 0);
        const int np = (live && trim) ? ninner + 2 : 0;   // trace points incl. the two end points
        const int sign = 1 - 2 * comp;
        const int b_first = comp ? bs.y : bs.x;        // tp[0].second
        const int b_last = comp ? bs.x : bs.y;         // tp[np-1].second
        const int a_base = (av.x / 100) * 100;         // inner point i sits at a_base + 100*i (hard-coded 100, LAInterface.cpp:4581-4584)
        int start_idx = np, end_idx = 0;
        int s_a = 0, s_b = 0, e_a = 0, e_b = 0;        // coordinates of the first / last point inside both masks
        bool s_found = false, e_found = false;
        int carry = 0;                                 // sum of B advances consumed so far
        int np_max = np;
        np_max = max(np_max, __shfl_xor(np_max, 16));
        np_max = max(np_max, __shfl_xor(np_max, 32));
        for (int base = 0; base < np_max; base += 16) {
            const int i = base + r;                    // trace point index
            int adv = 0;
            if (i >= 1 && i <= ninner) {
                const int64_t p = toff + (int64_t)TB * (2 * (i - 1) + 1);
                adv = TB == 1 ? (int)trace[p] : (int)(trace[p] | (trace[p + 1] << 8));
            }
            const int sc = row_incl_scan(adv);
            int pa, pb;
            if (i == 0) { pa = av.x; pb = b_first; }
            else if (i == np - 1) { pa = av.y; pb = b_last; }
            else { pa = a_base + 100 * i; pb = b_first + sign * (carry + sc); }
            carry += __shfl(sc, (row << 4) | 15);
            const bool valid = i < np;
            bool cs, ce;
            if (comp == 0) {
                cs = valid && (pa >= ea.x) && (pb >= eb.x);
                ce = valid && (pa <= ea.y) && (pb <= eb.y);
            } else {
                cs = valid && (pa >= ea.x) && (pb <= eb.y);
                ce = valid && (pa <= ea.y) && (pb >= eb.x);
            }
            const unsigned bs16 = (unsigned)((__ballot(cs) >> (row << 4)) & 0xffffull);
            const unsigned be16 = (unsigned)((__ballot(ce) >> (row << 4)) & 0xffffull);
            if (!s_found && bs16) {
                const int f = __ffs(bs16) - 1;
                s_a = __shfl(pa, (row << 4) | f);
                s_b = __shfl(pb, (row << 4) | f);
                start_idx = base + f;
                s_found = true;
            }
            if (be16) {
                const int l = 31 - __clz(be16);
                e_a = __shfl(pa, (row << 4) | l);
                e_b = __shfl(pb, (row << 4) | l);
                end_idx = base + l;
                e_found = true;
            }
        }
        if (live && r == 0) {
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
