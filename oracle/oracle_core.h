// TEST INFRASTRUCTURE - NOT PRODUCT CODE (see oracle_io.h).
// CPU restatement of the LAInterface functions on the hot path, statement for statement:
//   profileCoverage        /root/reference/src/lib/LAInterface.cpp:4298-4320
//   GetMatchingPosition    /root/reference/src/lib/LAInterface.cpp:4498-4546
//   trim_overlap           /root/reference/src/lib/LAInterface.cpp:4552-4683
//   AddTypesAsymmetric     /root/reference/src/lib/LAInterface.cpp:4721-4806
//   comparators            /root/reference/src/lib/LAInterface.cpp:4875-4927
//   ProcessAlignment       /root/reference/src/maximal/maximal.cpp:65-134 (= hinging.cpp:78-147)
// Pinned against the compiled reference functions by tests/test_oracle_vs_ref.py (oracle/_ref).
#pragma once
#include <cmath>
#include <utility>
#include "oracle_io.h"

namespace oracle {

typedef std::pair<int, int> IPair;

static inline bool compare_event(IPair e1, IPair e2) { return e1.first < e2.first; }
static inline bool pairAscend(const IPair& x, const IPair& y) { return x.first < y.first; }
static inline bool pairDescend(const IPair& x, const IPair& y) { return x.first > y.first; }
static inline bool compare_overlap(Ovl* o1, Ovl* o2) {
    return (o1->ae - o1->ab + o1->be - o1->bb) > (o2->ae - o2->ab + o2->be - o2->bb);
}
static inline bool compare_overlap_weight(Ovl* o1, Ovl* o2) { return o1->weight > o2->weight; }

static inline void profile_coverage(const std::vector<Ovl*>& alns, std::vector<IPair>& coverage, int reso, int cutoff) {
    std::vector<IPair> events;
    for (size_t i = 0; i < alns.size(); i++) {
        events.push_back(IPair(alns[i]->ab + cutoff, 1));
        events.push_back(IPair(alns[i]->ae - cutoff, -1));
    }
    std::sort(events.begin(), events.end(), compare_event);
    size_t pos = 0;
    int i = 0;
    int count = 0;
    while (pos < events.size()) {
        // the reference evaluates events[pos] before the bound (UB that never changes the result)
        while ((pos < events.size()) && (events[pos].first < i * reso)) {
            count += events[pos].second;
            pos++;
        }
        coverage.push_back(IPair(i * reso, count));
        i++;
    }
}

static inline int get_matching_position(const Ovl* o, int pos_A) {
    if ((pos_A < o->ab) || (pos_A > o->ae)) return -1;
    int rev_sign = 1 - 2 * o->comp;
    int cur_A = o->ab;
    int next_A = cur_A;
    int cur_B = o->bb;
    if (o->comp == 1) cur_B = o->be;
    for (int j = 0; j < o->tlen / 2 - 1; j++) {
        if (cur_A % 100 != 0)
            next_A = int(ceil(cur_A / 100.0)) * 100;
        else
            next_A = cur_A + 100;
        if (next_A >= pos_A) return cur_B + pos_A - cur_A;
        cur_B = cur_B + rev_sign * o->trace[2 * j + 1];
        cur_A = next_A;
    }
    if (cur_A < pos_A) return cur_B + pos_A - cur_A;
    return -2;
}

static inline void trim_overlap(Ovl* o) {
    o->eff_bb = o->bb; o->eff_be = o->be; o->eff_ab = o->ab; o->eff_ae = o->ae;
    std::vector<IPair> tp;
    if (o->comp == 0) tp.push_back(IPair(o->ab, o->bb));
    else tp.push_back(IPair(o->ab, o->be));
    int rev_sign = 1 - 2 * o->comp;
    int cur_A = o->ab;
    for (int j = 0; j < o->tlen / 2 - 1; j++) {
        if (cur_A % 100 != 0) cur_A = int(ceil(cur_A / 100.0)) * 100;
        else cur_A += 100;
        tp.push_back(IPair(cur_A, tp.back().second + rev_sign * o->trace[2 * j + 1]));
    }
    if (o->comp == 0) tp.push_back(IPair(o->ae, o->be));
    else tp.push_back(IPair(o->ae, o->bb));

    o->eff_start_idx = (int)tp.size();
    o->eff_end_idx = 0;
    if (o->comp == 0) {
        for (int i = 0; i < (int)tp.size(); i++)
            if ((tp[i].first >= o->eff_a_rs) && (tp[i].second >= o->eff_b_rs)) {
                o->eff_ab = tp[i].first; o->eff_bb = tp[i].second; o->eff_start_idx = i; break;
            }
        for (int i = (int)tp.size() - 1; i >= 0; i--)
            if ((tp[i].first <= o->eff_a_re) && (tp[i].second <= o->eff_b_re)) {
                o->eff_ae = tp[i].first; o->eff_be = tp[i].second; o->eff_end_idx = i; break;
            }
    } else {
        for (int i = 0; i < (int)tp.size(); i++)
            if ((tp[i].first >= o->eff_a_rs) && (tp[i].second <= o->eff_b_re)) {
                o->eff_ab = tp[i].first; o->eff_be = tp[i].second; o->eff_start_idx = i; break;
            }
        for (int i = (int)tp.size() - 1; i >= 0; i--)
            if ((tp[i].first <= o->eff_a_re) && (tp[i].second >= o->eff_b_rs)) {
                o->eff_ae = tp[i].first; o->eff_bb = tp[i].second; o->eff_end_idx = i; break;
            }
    }
    if (o->eff_start_idx >= o->eff_end_idx) o->active = false;
}

static inline void add_types_asymmetric(Ovl* o, int max_overhang, int min_overhang) {
    int A_left = o->eff_ab - o->eff_a_rs;
    int A_right = o->eff_a_re - o->eff_ae;
    int B_left = o->eff_bb - o->eff_b_rs;
    int B_right = o->eff_b_re - o->eff_be;
    if (o->comp == 1) {
        B_left = o->eff_b_re - o->eff_be;
        B_right = o->eff_bb - o->eff_b_rs;
    }
    if ((std::max(A_left, A_right) < max_overhang) && (std::min(B_left, B_right) > min_overhang))
        o->type = BCOVERA;
    else if ((std::max(B_left, B_right) < max_overhang) && (std::min(A_left, A_right) > min_overhang))
        o->type = ACOVERB;
    else if (std::min(A_left, A_right) > max_overhang)
        o->type = INTERNAL;
    else if (A_left <= max_overhang) {
        if ((B_right <= max_overhang) && (B_left >= max_overhang)) o->type = BACKWARD;
        else if ((B_right >= max_overhang) && (B_left >= max_overhang)) o->type = BACKWARD_INTERNAL;
    } else if (A_right <= max_overhang) {
        if ((B_left <= max_overhang) && (B_right >= max_overhang)) o->type = FORWARD;
        else if ((B_left >= max_overhang) && (B_right >= max_overhang)) o->type = FORWARD_INTERNAL;
        else o->type = UNDEFINED;
    }
}

static inline bool process_alignment(Ovl* m, Read* A, Read* B, int ALN_THRESHOLD, int THETA, int THETA2, bool trim) {
    bool contained = false;
    m->eff_a_rs = A->effective_start; m->eff_a_re = A->effective_end;
    m->eff_b_rs = B->effective_start; m->eff_b_re = B->effective_end;
    if (trim) trim_overlap(m);
    else { m->eff_bb = m->bb; m->eff_be = m->be; m->eff_ab = m->ab; m->eff_ae = m->ae; }
    if (((m->eff_be - m->eff_bb) < ALN_THRESHOLD) || ((m->eff_ae - m->eff_ab) < ALN_THRESHOLD) || (!m->active)) {
        m->active = false;
        m->type = NOT_ACTIVE;
    } else {
        add_types_asymmetric(m, THETA, THETA2);
        if (m->type == BCOVERA) contained = true;
    }
    m->weight = m->eff_ae - m->eff_ab + m->eff_be - m->eff_bb;
    m->length = m->ae - m->ab + m->be - m->bb;
    return contained;
}

}  // namespace oracle
