// K5: best-overlap (greedy) edge selection of `hinge layout`, layout/hinging.cpp:1911-2148.
// The reference walks every active read's forward and backward match lists (already in compare_overlap_weight order) and
// keeps at most one edge per direction.  The walk of one (read, direction) reads only final state - which reads are active,
// the hinges of the B reads with their active flags, the read's own killed hinges - so the 2 x n_reads walks are independent:
// one thread each.  (`hinge_pos`, which the reference carries in one variable across reads (hinging.cpp:1909), is always
// assigned in the same walk that picks the edge it is printed with - 1965 / 2026 and the FORWARD_INTERNAL branch - so the
// carried value never reaches an output line; every walk starts its own.)
// The lists are a few dozen entries per read: integer compares on data that fits L2; the point of the kernel is that a sharded
// layout can select the edges of its own block where the classified matches already are.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "align_kernels.h"

namespace hinge {

struct SelMatch {   // 36 bytes, one per classified match (fields of LOverlap the selection reads)
    int b, comp, type, active, weight;
    int eff_bb, eff_be;   // eff_read_B_match_start_ / _end_ (poison test)
    int bb, be;           // read_B_match_start_ / _end_ (hinge anchor)
};
struct SelHinge { int pos, type, active; };
struct SelKilled { int pos, type; };

__global__ __launch_bounds__(256) void k_select_edges(int n_reads, const unsigned char* __restrict__ read_active,
                                                      const int64_t* __restrict__ off_fwd, const int64_t* __restrict__ off_bwd,
                                                      const SelMatch* __restrict__ m, const int64_t* __restrict__ h_off,
                                                      const SelHinge* __restrict__ h, const int64_t* __restrict__ k_off,
                                                      const SelKilled* __restrict__ kh, int hinge_tolerance, int hinge_slack,
                                                      int* __restrict__ chosen /*[2][n_reads]*/, int* __restrict__ chosen_hpos /*[2][n_reads]*/,
                                                      int* __restrict__ poison_hits /*[n_matches], zeroed by the caller*/) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * (int64_t)n_reads) return;
    const int dirn = t >= n_reads ? 1 : 0;          // 0 = forward walk (hinging.cpp:1926-2024), 1 = backward walk (:2038-2130)
    const int i = (int)(t - (int64_t)dirn * n_reads);
    int pick = -1, hpos = -1;
    if (read_active[i]) {
        const int64_t s = dirn == 0 ? off_fwd[i] : off_bwd[i], e = dirn == 0 ? off_fwd[i + 1] : off_bwd[i + 1];
        const int t_plain = dirn == 0 ? MT_FORWARD : MT_BACKWARD, t_internal = dirn == 0 ? MT_FORWARD_INTERNAL : MT_BACKWARD_INTERNAL;
        int plain = 0, internal = 0, pick_weight = 0;
        for (int64_t j = s; j < e; j++) {
            const SelMatch x = m[j];
            if (!x.active || !read_active[x.b]) continue;
            if (x.type == t_plain && plain == 0) {
                // a killed hinge of this read beyond the match's effective end on B poisons it (every hit is one line of .edges.skipped)
                int hits = 0;
                for (int64_t q = k_off[i]; q < k_off[i + 1]; q++) {
                    const SelKilled nk = kh[q];
                    bool hit;
                    if (dirn == 0) hit = ((x.comp != 1) && (nk.type == -1) && (nk.pos > x.eff_be)) || ((x.comp == 1) && (nk.type == 1) && (nk.pos < x.eff_bb));
                    else hit = ((x.comp != 1) && (nk.type == 1) && (nk.pos < x.eff_bb)) || ((x.comp == 1) && (nk.type == -1) && (nk.pos > x.eff_be));
                    hits += hit ? 1 : 0;
                }
                if (hits) poison_hits[j] = hits;
                else { pick = (int)j; pick_weight = x.weight; hpos = -1; plain = 1; }
            } else if (x.type == t_internal && internal == 0 && h_off[x.b + 1] > h_off[x.b]) {
                int anchor, want;
                if (dirn == 0) { anchor = x.comp == 1 ? x.be : x.bb; want = 1 - 2 * x.comp; }
                else { anchor = x.comp == 1 ? x.bb : x.be; want = -1 + 2 * x.comp; }
                for (int64_t q = h_off[x.b]; q < h_off[x.b + 1]; q++) {
                    const SelHinge hb = h[q];
                    if ((anchor > hb.pos - hinge_tolerance) && (anchor < hb.pos + hinge_tolerance) && hb.type == want && hb.active) {
                        if (plain == 0 || x.weight > pick_weight - 2 * hinge_slack) { pick = (int)j; pick_weight = x.weight; plain = 1; internal = 1; hpos = hb.pos; }
                        break;   // the first hinge of B inside the tolerance decides, taken or not
                    }
                }
            }
        }
    }
    chosen[t] = pick;
    chosen_hpos[t] = hpos;
}

}  // namespace hinge
