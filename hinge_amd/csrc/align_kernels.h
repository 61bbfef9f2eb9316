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
// Eight lanes own one overlap (eight overlaps per wavefront).  Per step every lane loads 8 consecutive trace bytes - the group
// reads one 64-byte line, coalesced - and owns the 4 (TB = 1) or 2 (TB = 2) trace points in them: B coordinates = b_first +
// sign * (advances of earlier steps + exclusive prefix over the group's lanes + running sum inside the lane).  Every lane
// keeps the first point inside both masks and the last one it has seen; the group's first / last are a min / max over 8 lanes.
// History: 16 lanes per overlap with one point per lane and step (ballots and five cross-lane reads per 16 points) was bound by
// instruction issue (5.0 ms for 24.7 M overlaps of ~120 B); one lane per overlap walking its own trace is lean in instructions
// but every lane pulls its own 64-byte lines through L1 / L2 (4.0 ms).
// The trim + classify of one overlap by the eight lanes of its group (call from all 64 lanes: cross-lane reads inside).
// `o` is complete in the group's lane 0 when `live`.  PADDED: the trace buffer has 8 spare bytes behind it (the library's own copy).
template <int TB, bool PADDED>
__device__ __forceinline__ void classify_group(const bool live, const int2 av, const int2 bs, const int comp, const int2 ea, const int2 eb, const int tl,
                                               const unsigned char* __restrict__ tp, const int aln_threshold, const int theta, const int theta2,
                                               const int trim, const int lane, ClassifyOut& o) {
    constexpr int GL = 8;                     // lanes per overlap
    constexpr int PAIR = 2 * TB;              // bytes per trace point (diffs, B advance)
    constexpr int PL = 8 / PAIR;              // trace points per lane and step
    const int r = lane & (GL - 1);            // lane inside the group
    const int glast = lane | (GL - 1);        // last lane of the group
    {
        const int ninner = max(tl / 2 - 1, 0);
        const int np = (live && trim) ? ninner + 2 : 0;   // trace points incl. the two end points
        const int sign = 1 - 2 * comp;
        const int b_first = comp ? bs.y : bs.x;        // tp[0].second
        const int b_last = comp ? bs.x : bs.y;         // tp[np-1].second
        const int a_base = (av.x / 100) * 100;         // inner point i sits at a_base + 100*i (hard-coded 100, LAInterface.cpp:4581-4584)
        const int tbytes_total = tl * TB;
        // "inside both masks" on q = sign * (B coordinate), which ascends along the trace for both strands:
        //   first point:  a >= ea.x  and  q >= qlo      last point:  a <= ea.y  and  q <= qhi
        const int qlo = comp ? -eb.y : eb.x, qhi = comp ? -eb.x : eb.y;
        const int q_first = sign * b_first;
        // this lane's first point inside both masks / last one (index INT_MAX / -1: none); branch-free updates
        int f_idx = INT_MAX, f_a = 0, f_q = 0, l_idx = -1, l_a = 0, l_q = 0;
        auto visit = [&](bool valid, int i, int pa, int q) {   // called with ascending i inside a lane
            const bool cs = valid && (pa >= ea.x) && (q >= qlo) && (f_idx == INT_MAX);
            const bool ce = valid && (pa <= ea.y) && (q <= qhi);
            f_idx = cs ? i : f_idx; f_a = cs ? pa : f_a; f_q = cs ? q : f_q;
            l_idx = ce ? i : l_idx; l_a = ce ? pa : l_a; l_q = ce ? q : l_q;
        };
        visit(np > 0 && r == 0, 0, av.x, q_first);
        int steps = np > 0 ? (ninner + GL * PL - 1) / (GL * PL) : 0;
        int steps_max = steps;                         // wave-uniform trip count
        steps_max = max(steps_max, __shfl_xor(steps_max, 8));
        steps_max = max(steps_max, __shfl_xor(steps_max, 16));
        steps_max = max(steps_max, __shfl_xor(steps_max, 32));
        int carry = 0;                                 // B advances of all earlier steps of this overlap
        for (int st = 0; st < steps_max; st++) {
            const int p0 = (st * GL + r) * PL;         // 0-based index of this lane's first inner point of the step
            unsigned long long w = 0;
            if (PADDED) {   // 8 bytes behind the trace buffer are ours: the last group of an overlap reads on into the next record
                if (p0 * PAIR < tbytes_total) __builtin_memcpy(&w, tp + (size_t)p0 * PAIR, 8);
            } else if (p0 * PAIR < tbytes_total) {
                if ((p0 + PL) * PAIR <= tbytes_total) {
                    __builtin_memcpy(&w, tp + (size_t)p0 * PAIR, 8);
                } else {   // the overlap's last, partial 8 bytes: byte by byte, nothing is read past the trace
                    for (int q = 0; q < 8 && p0 * PAIR + q < tbytes_total; q++) w |= (unsigned long long)tp[(size_t)p0 * PAIR + q] << (8 * q);
                }
            }
            int adv[PL], tot = 0;
#pragma unroll
            for (int q = 0; q < PL; q++) {
                const int i = p0 + q + 1;
                const int v = TB == 1 ? (int)((w >> (16 * q + 8)) & 0xffull) : (int)((w >> (32 * q + 16)) & 0xffffull);
                adv[q] = i <= ninner ? v : 0;
                tot += adv[q];
            }
            int incl = tot;                            // inclusive prefix of the lanes' totals inside the group
#pragma unroll
            for (int d = 1; d < GL; d <<= 1) { const int t = __shfl_up(incl, d); incl += r >= d ? t : 0; }
            int run = carry + incl - tot;
#pragma unroll
            for (int q = 0; q < PL; q++) {
                const int i = p0 + q + 1;
                run += adv[q];
                visit(np > 0 && i <= ninner, i, a_base + 100 * i, q_first + run);
            }
            carry += __shfl(incl, glast);
        }
        visit(np > 0 && r == 0, np - 1, av.y, sign * b_last);
        // group minimum of (first index, lane) / maximum of (last index, lane), then the winners' coordinates
        int fk = f_idx == INT_MAX ? INT_MAX : f_idx * GL + r, lk = l_idx < 0 ? -1 : l_idx * GL + r;
#pragma unroll
        for (int d = 1; d < GL; d <<= 1) { fk = min(fk, __shfl_xor(fk, d)); lk = max(lk, __shfl_xor(lk, d)); }
        const bool s_found = fk != INT_MAX, e_found = lk >= 0;
        const int fl = (lane & ~(GL - 1)) | (s_found ? (fk & (GL - 1)) : 0), ll = (lane & ~(GL - 1)) | (e_found ? (lk & (GL - 1)) : 0);
        const int s_a = __shfl(f_a, fl), s_b = sign * __shfl(f_q, fl), e_a = __shfl(l_a, ll), e_b = sign * __shfl(l_q, ll);
        int start_idx = s_found ? fk / GL : np, end_idx = e_found ? lk / GL : 0;
        if (live && r == 0) {
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
        }
    }
}

template <int TB, bool PADDED>
__global__ __launch_bounds__(BLOCK) void k_trim_classify(int64_t n_sel, const int64_t* __restrict__ sel, const int* __restrict__ a_of,
                                                         const int2* __restrict__ a_span, const int2* __restrict__ b_span,
                                                         const unsigned* __restrict__ b_flag, const unsigned char* __restrict__ trace,
                                                         const int64_t* __restrict__ trace_off, const int* __restrict__ tlen,
                                                         const int2* __restrict__ eff /*[n_reads] effective_start/end*/, int aln_threshold,
                                                         int theta, int theta2, ClassifyOut* __restrict__ out,
                                                         unsigned char* __restrict__ type_out /*nullptr, or only the match type is wanted*/,
                                                         int trim /*0: PAF input, ProcessAlignment(trim = false): the match is taken as it is*/) {
    constexpr int GL = 8;
    const int lane = lane_id();
    const int64_t groups_total = (int64_t)gridDim.x * (BLOCK / GL);
    const int64_t my_group = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) / GL;
    // the eight groups of a wave iterate together; a group without work idles with j >= n_sel
    for (int64_t j0 = my_group - (lane >> 3); j0 < n_sel; j0 += groups_total) {
        const int64_t j = j0 + (lane >> 3);
        const bool live = j < n_sel;
        int2 av = make_int2(0, 0), bs = make_int2(0, 0), ea = make_int2(0, 0), eb = make_int2(0, 0);
        int comp = 0, tl = 0;
        const unsigned char* tp = trace;
        if (live) {
            const int64_t k = sel[j];
            av = a_span[k];
            bs = b_span[k];
            const unsigned bf = b_flag[k];
            comp = (int)(bf >> 31);
            ea = eff[a_of[j]];
            eb = eff[bf & 0x7fffffffu];
            tl = tlen[k];
            tp = trace + trace_off[k];
        }
        ClassifyOut o;
        classify_group<TB, PADDED>(live, av, bs, comp, ea, eb, tl, tp, aln_threshold, theta, theta2, trim, lane, o);
        if (live && (lane & (GL - 1)) == 0) {
            if (type_out) type_out[j] = (unsigned char)o.type; else out[j] = o;
        }
    }
}

// The same for EVERY overlap of the part, in storage order: one wavefront per A read, eight overlaps per step.  All the
// per-overlap arrays and the traces (consecutive in the .las) are read coalesced; only eff[B] is a gather (an 8-byte table
// entry per read).  `hinge maximal` classifies nearly every overlap (the best one or two per (A, B) pair), and a list of
// selected overlaps in hash-map order costs six scattered 64-byte lines per overlap: that, not the trace walk, is what held
// the list form at 4-5 ms for 24.7 M overlaps whatever the lane layout.
template <int TB, bool PADDED>
__global__ __launch_bounds__(BLOCK) void k_trim_classify_rows(int r_begin, int r_end, const int64_t* __restrict__ row_ptr,
                                                              const int2* __restrict__ a_span, const int2* __restrict__ b_span,
                                                              const unsigned* __restrict__ b_flag, const unsigned char* __restrict__ trace,
                                                              const int64_t* __restrict__ trace_off, const int* __restrict__ tlen,
                                                              const int2* __restrict__ eff, int aln_threshold, int theta, int theta2,
                                                              unsigned char* __restrict__ type_out /*[n_ovl] of the part*/, int trim) {
    constexpr int GL = 8;
    const int lane = lane_id();
    const int g = lane >> 3;
    const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * BLOCK + threadIdx.x) >> 6);
    const int nwaves = (gridDim.x * BLOCK) >> 6;
    for (int i = r_begin + wave; i <= r_end; i += nwaves) {
        const int64_t s = row_ptr[i], e = row_ptr[i + 1];
        const int2 ea = eff[i];
        for (int64_t k0 = s; k0 < e; k0 += WAVE / GL) {
            const int64_t k = k0 + g;
            const bool live = k < e;
            int2 av = make_int2(0, 0), bs = make_int2(0, 0), eb = make_int2(0, 0);
            int comp = 0, tl = 0;
            const unsigned char* tp = trace;
            if (live) {
                av = a_span[k];
                bs = b_span[k];
                const unsigned bf = b_flag[k];
                comp = (int)(bf >> 31);
                eb = eff[bf & 0x7fffffffu];
                tl = tlen[k];
                tp = trace + trace_off[k];
            }
            ClassifyOut o;
            classify_group<TB, PADDED>(live, av, bs, comp, ea, eb, tl, tp, aln_threshold, theta, theta2, trim, lane, o);
            if (live && (lane & (GL - 1)) == 0) type_out[k] = (unsigned char)o.type;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The part form again, as a STREAM: k_trim_classify_rows spends 416 vector instructions per 8 overlaps evaluating "inside both
// masks" for every trace point on eight lanes, and is bound by instruction issue at 15 % of the HBM rate.  Here ONE lane owns
// one overlap, and the raw .las bytes of 64 consecutive overlaps (records and traces as they lie in the file) are first copied
// into LDS with coalesced 16-byte loads - the only global traffic besides the SoA columns - so the per-lane walks that follow
// read LDS, not 64 scattered cache lines per instruction (what held the earlier one-lane-per-overlap kernel at 4 ms).
// A lane needs three walks, and only the first touches every point, with two instructions each:
//   1. T = sum of the B advances of the inner points (the B coordinate of the last inner point is b_first + sign * T);
//   2. forward from point 0 until the first point inside both masks (usually a handful of points);
//   3. backward from the last point until the last point inside both masks.
// "First / last index that satisfies the test" is evaluated in index order with an early exit, which is the reference's
// definition (LAInterface.cpp:4606-4640) whatever the coordinates look like - no monotonicity is assumed.
// One wavefront per workgroup (the stage buffer is its own), CAP bytes of LDS each.
// ------------------------------------------------------------------------------------------------
// STREAM_CAP = bytes of staged .las per wavefront: 64 overlaps of ~130 B are 8.3 KB.  10240 -> 16 workgroups (wavefronts) per CU,
// 8192 -> 20 (a step then often needs a second sub-step for its last few overlaps); HINGE_K4_CAP picks the instantiation.

template <int TB, typename FETCH, typename SUM>
__device__ __forceinline__ void classify_lane(const int2 av, const int2 bs, const int comp, const int2 ea, const int2 eb, const int tl,
                                              FETCH adv /*B advance of trace pair j*/, SUM sum_adv /*sum of adv(0 .. count - 1)*/, const int aln_threshold,
                                              const int theta, const int theta2, const int trim, ClassifyOut& o) {
    const int ninner = max(tl / 2 - 1, 0);
    const int np = trim ? ninner + 2 : 0;
    const int sign = 1 - 2 * comp;
    const int b_first = comp ? bs.y : bs.x, b_last = comp ? bs.x : bs.y;
    const int a_base = (av.x / 100) * 100;          // inner point i sits at a_base + 100 * i (hard-coded 100, LAInterface.cpp:4581-4584)
    const int qlo = comp ? -eb.y : eb.x, qhi = comp ? -eb.x : eb.y;
    const int q_first = sign * b_first, q_last = sign * b_last;
    bool s_found = false, e_found = false;
    int s_idx = np, s_a = 0, s_q = 0, e_idx = 0, e_a = 0, e_q = 0;
#ifndef HINGE_K4_ABLATE
#define HINGE_K4_ABLATE 0   // timing-only builds (tools/probes/k4_ablate.sh; results are WRONG by construction): 1 no forward walk, 2 no backward walk, 4 no advance sum
#endif
    if (np > 0) {
        const int T = (HINGE_K4_ABLATE & 4) ? ninner * 100 : sum_adv(ninner);
        // first point (ascending index) with a >= ea.x and q >= qlo
        if (av.x >= ea.x && q_first >= qlo) { s_found = true; s_idx = 0; s_a = av.x; s_q = q_first; }
        else {
            int q = q_first;
            for (int i = 1; i <= ((HINGE_K4_ABLATE & 1) ? 0 : ninner); i++) {
                q += adv(i - 1);
                const int a = a_base + 100 * i;
                if (a >= ea.x && q >= qlo) { s_found = true; s_idx = i; s_a = a; s_q = q; break; }
            }
            if (!s_found && av.y >= ea.x && q_last >= qlo) { s_found = true; s_idx = np - 1; s_a = av.y; s_q = q_last; }
        }
        // last point (descending index) with a <= ea.y and q <= qhi
        if (av.y <= ea.y && q_last <= qhi) { e_found = true; e_idx = np - 1; e_a = av.y; e_q = q_last; }
        else {
            int q = q_first + T;
            for (int i = ninner; i >= ((HINGE_K4_ABLATE & 2) ? ninner + 1 : 1); i--) {
                const int a = a_base + 100 * i;
                if (a <= ea.y && q <= qhi) { e_found = true; e_idx = i; e_a = a; e_q = q; break; }
                q -= adv(i - 1);
            }
            if (!e_found && av.x <= ea.y && q_first <= qhi) { e_found = true; e_idx = 0; e_a = av.x; e_q = q_first; }
        }
    }
    int start_idx = s_found ? s_idx : np, end_idx = e_found ? e_idx : 0;
    o.eff_ab = av.x; o.eff_ae = av.y; o.eff_bb = bs.x; o.eff_be = bs.y;
    if (comp == 0) {
        if (s_found) { o.eff_ab = s_a; o.eff_bb = sign * s_q; }
        if (e_found) { o.eff_ae = e_a; o.eff_be = sign * e_q; }
    } else {
        if (s_found) { o.eff_ab = s_a; o.eff_be = sign * s_q; }
        if (e_found) { o.eff_ae = e_a; o.eff_bb = sign * e_q; }
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
}

// The staging copy of the two stream kernels: `nchunk` 16-byte pieces of the image from `base` into LDS.  EVERY load of the wavefront
// is issued before the first LDS store (STREAM_CAP / 1024 loads per lane in flight): a loop of "load, wait, store" pays the memory
// latency once per 1 KiB - 8 round trips in a row per 64 overlaps, which is what held both kernels at 13 us per step (round 5).
template <int STREAM_CAP>
__device__ __forceinline__ void stage_image(uint4* __restrict__ stage4, const unsigned char* __restrict__ src, const int64_t base, const int nchunk,
                                            const int64_t readable, const int lane) {
    constexpr int NL = STREAM_CAP / (16 * WAVE);
    static_assert(NL * 16 * WAVE == STREAM_CAP, "the stage buffer is a whole number of wavefront-wide 16-byte loads");
    uint4 w[NL];
#pragma unroll
    for (int u = 0; u < NL; u++) {
        const int c = lane + u * WAVE;
        const int64_t off = base + 16ll * c;
        w[u] = make_uint4(0, 0, 0, 0);
        if (c < nchunk && off + 16 <= readable) w[u] = *reinterpret_cast<const uint4*>(src + off);
    }
#pragma unroll
    for (int u = 0; u < NL; u++) {
        const int c = lane + u * WAVE;
        if (c < nchunk) stage4[c] = w[u];
    }
    if (base + 16ll * nchunk > readable) {   // (uniform) the image's last, partial 16 bytes: byte by byte, nothing is read past the buffer
        for (int c = lane; c < nchunk; c += WAVE) {
            const int64_t off = base + 16ll * c;
            if (off + 16 <= readable) continue;
            uint4 t = make_uint4(0, 0, 0, 0);
            unsigned char* wb = reinterpret_cast<unsigned char*>(&t);
            for (int q = 0; q < 16 && off + q < readable; q++) wb[q] = src[off + q];
            stage4[c] = t;
        }
    }
}

template <int TB, int STREAM_CAP>
__global__ __launch_bounds__(WAVE) void k_trim_classify_stream(int r_begin, int r_end, const int64_t* __restrict__ row_ptr,
                                                               const int2* __restrict__ a_span, const int2* __restrict__ b_span,
                                                               const unsigned* __restrict__ b_flag, const unsigned char* __restrict__ trace,
                                                               int64_t trace_readable /*bytes that may be read from `trace`*/,
                                                               const int64_t* __restrict__ trace_off, const int* __restrict__ tlen,
                                                               const int2* __restrict__ eff, int aln_threshold, int theta, int theta2,
                                                               unsigned char* __restrict__ type_out /*[n_ovl] or nullptr*/,
                                                               ClassifyOut* __restrict__ full_out /*[n_ovl] or nullptr*/, int trim) {
    __shared__ uint4 stage4[STREAM_CAP / 16];
    const unsigned char* stage = reinterpret_cast<const unsigned char*>(stage4);
    const int lane = threadIdx.x;
    for (int i = r_begin + (int)blockIdx.x; i <= r_end; i += (int)gridDim.x) {
        const int64_t s = row_ptr[i], e = row_ptr[i + 1];
        const int2 ea = eff[i];
        for (int64_t k0 = s; k0 < e; k0 += WAVE) {
            const int64_t k = k0 + lane;
            const bool live = k < e;
            int2 av = make_int2(0, 0), bs = make_int2(0, 0), eb = make_int2(0, 0);
            int comp = 0, tl = 0;
            int64_t t0 = 0;
            if (live) {
                av = a_span[k];
                bs = b_span[k];
                const unsigned bf = b_flag[k];
                comp = (int)(bf >> 31);
                eb = eff[bf & 0x7fffffffu];
                tl = tlen[k];
                t0 = trace_off[k];
            }
            const int need = (trim && tl >= 4) ? (tl - 2) * TB : 0;   // bytes of the pairs the walks read (the last pair is never read)
            const int64_t t1 = t0 + need;
            unsigned long long pending = ballot_of(live);
            while (pending) {
                const int first = __ffsll((long long)pending) - 1;
                const int64_t base = __shfl(t0, first) & ~15ll;
                const bool mine = ((pending >> lane) & 1ull) != 0;
                // offsets ascend with the lane (storage order), so the lanes that fit are a prefix of the pending ones; whatever the
                // offsets are, only the pending lanes in front of the first one that does not fit are taken
                const unsigned long long fits = ballot_of(mine && t0 >= base && (t1 - base) <= (int64_t)STREAM_CAP);
                const unsigned long long nofit = pending & ~fits;
                const unsigned long long take = nofit ? (fits & ((1ull << (__ffsll((long long)nofit) - 1)) - 1ull)) : fits;
                ClassifyOut o;
                if (take == 0ull) {
                    // one overlap whose trace alone exceeds the stage buffer (> 5000 trace points): its lane walks global memory
                    if (lane == first) {
                        const unsigned char* tp = trace + t0;
                        auto adv = [&](int j) { return TB == 1 ? (int)tp[2 * j + 1] : (int)(tp[4 * j + 2] | (tp[4 * j + 3] << 8)); };
                        classify_lane<TB>(av, bs, comp, ea, eb, tl, adv, [&](int cnt) { int T = 0; for (int j = 0; j < cnt; j++) T += adv(j); return T; },
                                          aln_threshold, theta, theta2, trim, o);
                        if (type_out) type_out[k] = (unsigned char)o.type;
                        if (full_out) full_out[k] = o;
                    }
                    pending &= ~(1ull << first);
                    continue;
                }
                const int last = 63 - __clzll((long long)take);
                const int64_t end = __shfl(t1, last);
                const int nchunk = (int)((end - base + 15) >> 4);
                stage_image<STREAM_CAP>(stage4, trace, base, nchunk, trace_readable, lane);
                __syncthreads();
                if ((take >> lane) & 1ull) {
                    const unsigned char* tp = stage + (int)(t0 - base);
                    auto adv = [&](int j) { return TB == 1 ? (int)tp[2 * j + 1] : (int)(tp[4 * j + 2] | (tp[4 * j + 3] << 8)); };
                    // The sum over every pair is the one walk that touches the whole trace: whole aligned words when the trace starts
                    // on an even byte (it does in a .las: 12 + a sum of even record sizes) - the advances are then bytes 1 and 3 of
                    // every word (one byte per value) resp. its upper half (two bytes per value) - else value by value.
                    auto sum_words = [&](int cnt) {
                        const unsigned lo = (unsigned)(t0 - base);
                        int T = 0;
                        if (cnt <= 0) return 0;
                        if (TB == 1 && !(lo & 1u)) {
                            const unsigned* W = reinterpret_cast<const unsigned*>(stage);
                            const unsigned hi = lo + 2u * (unsigned)cnt;            // exclusive end, even
                            const unsigned w0 = lo >> 2, w1 = (hi - 1u) >> 2;
                            unsigned m0 = (lo & 2u) ? 0xff000000u : 0xff00ff00u;    // a trace that starts in the word's second half
                            const unsigned m1 = (hi & 2u) ? 0x0000ff00u : 0xff00ff00u;   // one that ends after its first half
                            if (w0 == w1) return (int)__builtin_amdgcn_sad_u8(W[w0] & m0 & m1, 0u, 0u);
                            unsigned acc = __builtin_amdgcn_sad_u8(W[w0] & m0, 0u, 0u);
                            unsigned w = w0 + 1;
                            for (; w + 4 <= w1; w += 4) {   // four independent LDS reads in flight per round trip
                                const unsigned x0 = W[w], x1 = W[w + 1], x2 = W[w + 2], x3 = W[w + 3];
                                acc = __builtin_amdgcn_sad_u8(x0 & 0xff00ff00u, 0u, acc);
                                acc = __builtin_amdgcn_sad_u8(x1 & 0xff00ff00u, 0u, acc);
                                acc = __builtin_amdgcn_sad_u8(x2 & 0xff00ff00u, 0u, acc);
                                acc = __builtin_amdgcn_sad_u8(x3 & 0xff00ff00u, 0u, acc);
                            }
                            for (; w < w1; w++) acc = __builtin_amdgcn_sad_u8(W[w] & 0xff00ff00u, 0u, acc);
                            return (int)__builtin_amdgcn_sad_u8(W[w1] & m1, 0u, acc);
                        }
                        if (TB == 2 && !(lo & 3u)) {
                            const unsigned* W = reinterpret_cast<const unsigned*>(stage) + (lo >> 2);
                            for (int j = 0; j < cnt; j++) T += (int)(W[j] >> 16);
                            return T;
                        }
                        for (int j = 0; j < cnt; j++) T += adv(j);
                        return T;
                    };
                    classify_lane<TB>(av, bs, comp, ea, eb, tl, adv, sum_words, aln_threshold, theta, theta2, trim, o);
                    if (type_out) type_out[k] = (unsigned char)o.type;
                    if (full_out) full_out[k] = o;
                }
                __syncthreads();
                pending &= ~take;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The stream form once more, reading NOTHING but the .las image (round 5).  k_trim_classify_stream stages the raw bytes of 64
// consecutive overlaps - their 40-byte records included, they lie between the traces - and then reads the very same fields a second
// time from the SoA columns (a_span, b_span, b_flag, tlen, trace_off: 32 B per overlap, 1.43 x the bytes the path needs).  Here the
// lane takes tlen, abpos, bbpos, aepos, bepos, flags, aread and bread (align.h:126-146: the Overlap record as DALIGNER writes it,
// without its trace pointer) from the staged record in LDS and does the strand flip of LAInterface.cpp:1619-1626 itself (rlen[B]
// joins the eff[A] / eff[B] gathers).  What is left of the columns is ONE 32-bit word per overlap - where its record starts, because
// a record's position is a chain through every tlen before it - and the kernel no longer knows about pile-ups at all:
//   window w = overlaps 64 w .. 64 w + 63 (storage order, whatever reads they belong to: A is in the record),
//   win_base[w] = byte offset of the window's first record in the image (win_base[n_windows] = where the last overlap ends),
//   rec_rel[k]  = offset of overlap k's record behind win_base[k / 64].
// A lane's bytes end where the next kept record starts (self-overlap records in between are staged along: nothing points at them).
//
// A wavefront is a chain of round trips per window (offsets -> image bytes -> gathers), and its 10 KiB stage buffer caps the CU at
// 16 of them, so the chain is what the kernel costs (9.8 us per window, 0.98 ms per 26.2 M overlaps with one window at a time).
// Therefore the wavefronts are persistent (window g, g + G, g + 2 G ... for wavefront g of G) and software-pipelined:
//   * the offsets of the NEXT window are loaded while this one is worked on;
//   * the image bytes of the next sub-step travel in REGISTERS (STREAM_CAP / 1024 x 16 bytes per lane) while the walks of this one
//     read LDS; they are stored to LDS when the walks are done;
//   * the gathers of this sub-step are issued BEFORE those loads (memory returns in order: waiting for a gather issued behind the
//     prefetch would wait for the prefetch) and travel during the one walk that touches the whole trace (the advance sum).
// Records start on even bytes (12 + a sum of even sizes); one that does not is read byte by byte.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lds_i32_h(const unsigned char* p) {   // a 32-bit field on a 2-byte boundary
    const unsigned short* h = reinterpret_cast<const unsigned short*>(p);
    return (int)((unsigned)h[0] | ((unsigned)h[1] << 16));
}
template <int TB> __device__ __forceinline__ int rec_field(const unsigned char* p, const bool aligned) {
    if (!aligned) return (int)((unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)p[2] << 16) | ((unsigned)p[3] << 24));   // (never in a .las)
    if (TB == 2) return *reinterpret_cast<const int*>(p);   // two-byte traces: every record size is a multiple of 4
    return lds_i32_h(p);
}

template <int TB, int STREAM_CAP>
__global__ __launch_bounds__(WAVE) void k_trim_classify_image(int64_t n_ovl, const unsigned char* __restrict__ image, int64_t image_readable,
                                                              const int64_t* __restrict__ win_base /*[n_windows + 1]*/,
                                                              const unsigned* __restrict__ rec_rel /*[n_ovl]*/, const int* __restrict__ rlen,
                                                              int n_reads, const int2* __restrict__ eff, int aln_threshold, int theta, int theta2,
                                                              unsigned char* __restrict__ type_out /*[n_ovl] or nullptr*/,
                                                              ClassifyOut* __restrict__ full_out /*[n_ovl] or nullptr*/) {
    constexpr int NL = STREAM_CAP / (16 * WAVE);
    static_assert(NL * 16 * WAVE == STREAM_CAP, "the stage buffer is a whole number of wavefront-wide 16-byte loads");
    __shared__ uint4 stage4[STREAM_CAP / 16];
    const unsigned char* stage = reinterpret_cast<const unsigned char*>(stage4);
    const int lane = threadIdx.x;
    const int64_t n_win = (n_ovl + WAVE - 1) / WAVE, G = (int64_t)gridDim.x;
    int64_t w_cur = (int64_t)blockIdx.x;
    if (w_cur >= n_win) return;

    // ---- the window whose sub-steps are being handed out, and the one behind it (already travelling) ----
    auto bcast64 = [](int64_t v, int src /*uniform*/) -> int64_t {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(unsigned long long)v, src);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v >> 32), src);
        return (int64_t)(((unsigned long long)hi << 32) | lo);
    };
    int64_t t0 = 0, t1 = 0;                      // this lane's overlap of the window: its bytes [t0, t1) of the image
    int64_t w_open = 0;                          // the window they belong to (uniform)
    unsigned long long pending = 0ull;           // lanes of the window not yet handed out
    unsigned n_rel = 0u; int64_t n_b0 = 0, n_b1 = 0, w_next = w_cur;   // the next window's offsets (loaded one window ahead)
    auto fetch_window = [&](int64_t w) {         // issue the loads of window w's offsets
        w_next = w;
        if (w < n_win) {
            const int64_t k = w * WAVE + lane;
            n_rel = k < n_ovl ? rec_rel[k] : 0u;
            n_b0 = win_base[w]; n_b1 = win_base[w + 1];
        }
    };
    auto open_window = [&]() -> bool {           // make the fetched window the current one, fetch the one behind it
        if (w_next >= n_win) return false;
        w_open = w_next;
        const int64_t kk = w_open * WAVE + lane;
        const bool live = kk < n_ovl;
        const unsigned nxt_rel = (unsigned)__shfl_down((int)n_rel, 1);
        t0 = live ? n_b0 + (int64_t)n_rel : 0;
        t1 = live ? ((lane < WAVE - 1 && kk + 1 < n_ovl) ? n_b0 + (int64_t)nxt_rel : n_b1) : 0;
        pending = ballot_of(live);
        fetch_window(w_open + G);
        return true;
    };
    // one sub-step = the longest prefix of the pending lanes whose bytes fit the stage buffer together
    struct Sub { int64_t base, win; unsigned long long take; int nchunk; int off0, len; bool valid; };   // off0, len: the lane's bytes behind `base`
    auto next_sub = [&]() -> Sub {
        Sub r; r.valid = false; r.base = 0; r.win = 0; r.take = 0ull; r.nchunk = 0; r.off0 = 0; r.len = 0;
        while (true) {
            if (!pending && !open_window()) return r;
            const int first = __ffsll((long long)pending) - 1;
            const int64_t base = bcast64(t0, first) & ~15ll;
            const bool mine = ((pending >> lane) & 1ull) != 0;
            const unsigned long long fits = ballot_of(mine && t0 >= base && t1 >= t0 + 40 && (t1 - base) <= (int64_t)STREAM_CAP);
            const unsigned long long nofit = pending & ~fits;
            const unsigned long long take = nofit ? (fits & ((1ull << (__ffsll((long long)nofit) - 1)) - 1ull)) : fits;
            if (take == 0ull) {
                // one overlap whose record + trace exceed the stage buffer (> 5000 trace points), or a table that is not an ascending
                // chain: its lane walks global memory, here and now
                if (lane == first && (t0 < 0 || t0 + 40 > image_readable)) {
                    // (a table entry that points outside the image: the caller's bug - nothing is read, the overlap is reported inactive)
                    const int64_t kk = w_open * WAVE + lane;
                    if (type_out) type_out[kk] = (unsigned char)MT_NOT_ACTIVE;
                    if (full_out) { ClassifyOut o = {0, 0, 0, 0, MT_NOT_ACTIVE, 0, 0, 0, 0, 0}; full_out[kk] = o; }
                } else if (lane == first) {
                    const unsigned char* rp = image + t0;
                    int f[9];
                    for (int q = 0; q < 9; q++) { unsigned v = 0; for (int c = 0; c < 4; c++) v |= (unsigned)rp[4 * q + c] << (8 * c); f[q] = (int)v; }
                    const int64_t room = min(t1, (int64_t)image_readable) - t0 - 40;      // trace bytes that are this overlap's AND inside the image
                    const int tl = room > 0 ? (int)min((int64_t)max(f[0], 0), room / TB) : 0;
                    const int comp = f[6] & 1, b = min(max(f[8], 0), n_reads - 1);
                    const int2 ea = eff[min(max(f[7], 0), n_reads - 1)], eb = eff[b];
                    const int bl = rlen[b];
                    const int2 av = make_int2(f[2], f[4]), bs = comp ? make_int2(bl - f[5], bl - f[3]) : make_int2(f[3], f[5]);
                    const unsigned char* tp = rp + 40;
                    auto adv = [&](int j) { return TB == 1 ? (int)tp[2 * j + 1] : (int)(tp[4 * j + 2] | (tp[4 * j + 3] << 8)); };
                    ClassifyOut o;
                    classify_lane<TB>(av, bs, comp, ea, eb, tl, adv, [&](int cnt) { int T = 0; for (int j = 0; j < cnt; j++) T += adv(j); return T; },
                                      aln_threshold, theta, theta2, 1, o);
                    const int64_t kk = w_open * WAVE + lane;
                    if (type_out) type_out[kk] = (unsigned char)o.type;
                    if (full_out) full_out[kk] = o;
                }
                pending &= ~(1ull << first);
                continue;
            }
            const int last = 63 - __clzll((long long)take);
            r.base = base; r.win = w_open; r.take = take; r.nchunk = (int)((bcast64(t1, last) - base + 15) >> 4);
            r.off0 = (int)(t0 - base); r.len = (int)(t1 - t0); r.valid = true;   // (meaningful on the lanes of `take`: both below STREAM_CAP)
            pending &= ~take;
            return r;
        }
    };
    // the sub-step's image bytes into registers: every load in flight at once.  Exactly NL loads, whatever the sub-step: pieces
    // behind its last one re-read that one, a piece that would cross the end of the buffer is read 16 bytes in front of it (land
    // repairs it), no sub-step at all reads the current one's first piece again: nothing in a branch.
    // (Twelve named registers, not an array: as an array the compiler kept it in scratch memory.)
    static_assert(NL <= 12, "twelve staging registers");
    uint4 w0, w1, w2, w3, w4, w5, w6, w7, w8, w9, w10, w11;
    const long long lim = (long long)image_readable - 16;
#define K4_LOAD(q) if (NL > q) w##q = *reinterpret_cast<const uint4*>(image + min(ib + 16ll * (long long)min(lane + q * WAVE, inc - 1), lim));
#define K4_ISSUE(base_, nchunk_) { const long long ib = (long long)(base_); const int inc = (nchunk_); \
        K4_LOAD(0) K4_LOAD(1) K4_LOAD(2) K4_LOAD(3) K4_LOAD(4) K4_LOAD(5) K4_LOAD(6) K4_LOAD(7) K4_LOAD(8) K4_LOAD(9) K4_LOAD(10) K4_LOAD(11) }
#define K4_STORE(q) if (NL > q) stage4[lane + q * WAVE] = w##q;
    auto repair = [&](const Sub& u) {            // (uniform) the image's last, partial 16 bytes: byte by byte
        if (u.base + 16ll * u.nchunk > image_readable) {
            for (int c = lane; c < u.nchunk; c += WAVE) {
                const int64_t off = u.base + 16ll * c;
                if (off + 16 <= image_readable) continue;
                uint4 t = make_uint4(0, 0, 0, 0);
                unsigned char* wb = reinterpret_cast<unsigned char*>(&t);
                for (int q = 0; q < 16 && off + q < image_readable; q++) wb[q] = image[off + q];
                stage4[c] = t;
            }
        }
    };
    // ... and from the registers into LDS (all of them, loaded or not: no wait hides in a branch)
#define K4_LAND(u_) { K4_STORE(0) K4_STORE(1) K4_STORE(2) K4_STORE(3) K4_STORE(4) K4_STORE(5) K4_STORE(6) K4_STORE(7) K4_STORE(8) K4_STORE(9) K4_STORE(10) K4_STORE(11) repair(u_); }

    fetch_window(w_cur);
    Sub cur = next_sub();
    if (cur.valid) K4_ISSUE(cur.base, cur.nchunk)
    while (cur.valid) {
        K4_LAND(cur)
        __syncthreads();
        const Sub nx = next_sub();               // (touches no LDS; its loads - the offsets of the window after next - are consumed a window later)
        const bool on = ((cur.take >> lane) & 1ull) != 0;
        const int off0 = on ? cur.off0 : 0, len = on ? cur.len : 40;          // (the other lanes read the buffer's first bytes: no branch)
        const unsigned char* rp = stage + off0;
        const bool al = ((unsigned)off0 & (TB == 2 ? 3u : 1u)) == 0u;
        int tl = rec_field<TB>(rp, al);
        const int2 av = make_int2(rec_field<TB>(rp + 8, al), rec_field<TB>(rp + 16, al));
        const int bb_raw = rec_field<TB>(rp + 12, al), be_raw = rec_field<TB>(rp + 20, al);
        const int comp = rec_field<TB>(rp + 24, al) & 1;
        const int a_read = min(max(rec_field<TB>(rp + 28, al), 0), n_reads - 1);
        const int b = min(max(rec_field<TB>(rp + 32, al), 0), n_reads - 1);
        tl = max(min(tl, (len - 40) / TB), 0);   // (a record that disagrees with the table reads nothing outside its own bytes)
        // the gathers: first used behind the advance sum
        int ea_x = eff[a_read].x, ea_y = eff[a_read].y, eb_x = eff[b].x, eb_y = eff[b].y;
        int bl = rlen[b];
        K4_ISSUE(nx.valid ? nx.base : cur.base, nx.valid ? nx.nchunk : 1)   // the next sub-step's bytes travel during everything below
        if (on) {
            const unsigned char* tp = rp + 40;
            const unsigned lo = (unsigned)off0 + 40u;
            auto adv = [&](int j) { return TB == 1 ? (int)tp[2 * j + 1] : (int)(tp[4 * j + 2] | (tp[4 * j + 3] << 8)); };
            auto sum_words = [&](int cnt) {   // as in k_trim_classify_stream
                int T = 0;
                if (cnt <= 0) return 0;
                if (!al) { for (int j = 0; j < cnt; j++) T += adv(j); return T; }
                if (TB == 1) {
                    const unsigned* W = reinterpret_cast<const unsigned*>(stage);
                    const unsigned hi = lo + 2u * (unsigned)cnt;            // exclusive end, even
                    const unsigned w0 = lo >> 2, w1 = (hi - 1u) >> 2;
                    unsigned m0 = (lo & 2u) ? 0xff000000u : 0xff00ff00u;
                    const unsigned m1 = (hi & 2u) ? 0x0000ff00u : 0xff00ff00u;
                    if (w0 == w1) return (int)__builtin_amdgcn_sad_u8(W[w0] & m0 & m1, 0u, 0u);
                    unsigned acc = __builtin_amdgcn_sad_u8(W[w0] & m0, 0u, 0u);
                    unsigned x = w0 + 1;
                    for (; x + 4 <= w1; x += 4) {
                        const unsigned x0 = W[x], x1 = W[x + 1], x2 = W[x + 2], x3 = W[x + 3];
                        acc = __builtin_amdgcn_sad_u8(x0 & 0xff00ff00u, 0u, acc);
                        acc = __builtin_amdgcn_sad_u8(x1 & 0xff00ff00u, 0u, acc);
                        acc = __builtin_amdgcn_sad_u8(x2 & 0xff00ff00u, 0u, acc);
                        acc = __builtin_amdgcn_sad_u8(x3 & 0xff00ff00u, 0u, acc);
                    }
                    for (; x < w1; x++) acc = __builtin_amdgcn_sad_u8(W[x] & 0xff00ff00u, 0u, acc);
                    return (int)__builtin_amdgcn_sad_u8(W[w1] & m1, 0u, acc);
                }
                const unsigned* W = reinterpret_cast<const unsigned*>(stage) + (lo >> 2);
                for (int j = 0; j < cnt; j++) T += (int)(W[j] >> 16);
                return T;
            };
            // the one walk over the whole trace needs none of the gathered values: they are first touched behind it
            int T = sum_words(max(tl / 2 - 1, 0));
            asm volatile("" : "+v"(T), "+v"(bl), "+v"(ea_x), "+v"(ea_y), "+v"(eb_x), "+v"(eb_y));
            const int2 ea = make_int2(ea_x, ea_y), eb = make_int2(eb_x, eb_y);
            // (strand flip, LAInterface.cpp:1619-1626: the only use of rlen[B])
            const int2 bs = comp ? make_int2(bl - be_raw, bl - bb_raw) : make_int2(bb_raw, be_raw);
            ClassifyOut o;
            classify_lane<TB>(av, bs, comp, ea, eb, tl, adv, [&](int) { return T; }, aln_threshold, theta, theta2, 1, o);
            const int64_t kk = cur.win * WAVE + lane;
            if (type_out) type_out[kk] = (unsigned char)o.type;
            if (full_out) full_out[kk] = o;
        }
        __syncthreads();
        cur = nx;
    }
#undef K4_LOAD
#undef K4_ISSUE
#undef K4_STORE
#undef K4_LAND
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
