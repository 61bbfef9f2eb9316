// TEST INFRASTRUCTURE - NOT PRODUCT CODE.
// CPU restatement ("oracle") of `hinge draft` (second half): /root/reference/src/consensus/draft.cpp -
// main() :720-1162 and draft_assembly_ctg :125-715 - with the functions under it:
//   LAInterface::getRead (lower-case bases)                    lib/LAInterface.cpp:1195-1286, lib/DB.c:1371-1412
//   LAInterface::getOverlap / getAlignment with a read range   lib/LAInterface.cpp:1404-1517, :2150-2560
//   recoverAlignment, getAlignmentTags                          oracle/cns_core.h (restated for `hinge consensus`, PINNED there)
//   LAInterface::getCoverage (LOverlap form)                   lib/LAInterface.cpp:4254-4263
//   falcon's banded O(ND) aligner `_align`                      lib/DW_banded.c:97-311
//   falcon's get_align_tags / get_cns_from_align_tags          lib/falcon.c:68-125, :246-517
// Only tests/ may use it.
//
// PARITY: draft.cpp includes spdlog and Boost.Graph (draft.cpp:14-25), neither is in this image, so the PROGRAM is unbuildable
// here and this restatement of its main() / draft_assembly_ctg is **parity unpinned** (like the three graph stages).  The
// primitives under it ARE pinned: falcon.c / DW_banded.c / kmer_lookup.c compile unmodified into oracle/_ref/libhinge_ref.so,
// and tests/test_draft_oracle.py holds falcon_align + falcon_tags + falcon_cns below against them on seeded ladders
// (oracle_falcon_ladder vs ref_falcon_ladder); recoverAlignment / getAlignmentTags are pinned through `hinge consensus`.
//
// Where the reference has undefined behaviour this restatement stops with a negative code instead of guessing (listed at
// oracle_draft below); everything else - including its quirks - is kept:
//   * `contig` is ONE variable for the whole run: a contig whose cut positions fail the size test prints the previous contig again;
//   * the last line of .edges.list is only the end-of-file marker: a file without a final newline loses its last line;
//   * prefix / suffix of a multi-read contig are cut from the FORWARD bases of the first / last A read whatever its strand, and the
//     coverage profile that picks a ladder's template is indexed in the forward frame with strand-frame positions;
//   * the first base the consensus trace-back emits is decided by a LINK index, not a base index (falcon.c: g_best_ck = best_ck);
//   * opening the output streams truncates <prefix>.garbage.txt, <prefix>.contained.txt and <out>.deadends.txt.
#include "cns_core.h"

#include <set>
#include <sstream>
#include <unordered_map>

namespace {

// ---- falcon: banded O(ND) alignment (DW_banded.c:97-311) -------------------------------------------------------------------
struct DPath { int d, k, pre_k, x1, y1, x2, y2; };
struct FAlign { std::string q, t; int dist = 0; bool aligned = false; };

static FAlign falcon_align(const std::string& query, const std::string& target, int band_tolerance) {
    const int q_len = (int)query.size(), t_len = (int)target.size();
    const int max_d = (int)(0.3 * (q_len + t_len));
    const int band_size = band_tolerance * 2;
    std::vector<int> V((size_t)max_d * 2 + 1, 0), U((size_t)max_d * 2 + 1, 0);
    const int k_offset = max_d;
    std::vector<DPath> d_path;
    FAlign out;
    int best_m = -1, min_k = 0, max_k = 0;
    for (int d = 0; d < max_d; d++) {
        if (max_k - min_k > band_size) break;
        int k, x = 0, y = 0;
        bool aligned = false;
        for (k = min_k; k <= max_k; k += 2) {
            int pre_k;
            if ((k == min_k) || ((k != max_k) && (V[(size_t)(k - 1 + k_offset)] < V[(size_t)(k + 1 + k_offset)]))) {
                pre_k = k + 1;
                x = V[(size_t)(k + 1 + k_offset)];
            } else {
                pre_k = k - 1;
                x = V[(size_t)(k - 1 + k_offset)] + 1;
            }
            y = x - k;
            DPath e;
            e.d = d; e.k = k; e.x1 = x; e.y1 = y;
            while (x < q_len && y < t_len && query[(size_t)x] == target[(size_t)y]) { x++; y++; }
            e.x2 = x; e.y2 = y; e.pre_k = pre_k;
            d_path.push_back(e);
            V[(size_t)(k + k_offset)] = x;
            U[(size_t)(k + k_offset)] = x + y;
            if (x + y > best_m) best_m = x + y;
            if (x >= q_len || y >= t_len) { aligned = true; break; }
        }
        int new_min_k = max_k, new_max_k = min_k;
        for (int k2 = min_k; k2 <= max_k; k2 += 2)
            if (U[(size_t)(k2 + k_offset)] >= best_m - band_tolerance) {
                if (k2 < new_min_k) new_min_k = k2;
                if (k2 > new_max_k) new_max_k = k2;
            }
        max_k = new_max_k + 1;
        min_k = new_min_k - 1;
        if (aligned) {
            out.aligned = true;
            out.dist = d;
            // (the entries are generated in (d, k) order: the reference's qsort + bsearch find entry (cd, ck) - so does this)
            auto find = [&](int cd, int ck) -> const DPath& {
                size_t lo = 0, hi = d_path.size();
                while (lo < hi) {
                    const size_t mid = (lo + hi) / 2;
                    const DPath& m = d_path[mid];
                    if (m.d < cd || (m.d == cd && m.k < ck)) lo = mid + 1; else hi = mid;
                }
                return d_path[lo];
            };
            std::vector<std::pair<int, int>> path;
            int cd = d, ck = k;
            while (cd >= 0 && (int)path.size() < q_len + t_len + 1) {
                const DPath& e = find(cd, ck);
                path.push_back({e.x2, e.y2});
                path.push_back({e.x1, e.y1});
                ck = e.pre_k;
                cd -= 1;
            }
            int idx = (int)path.size() - 1;
            int cx = path[(size_t)idx].first, cy = path[(size_t)idx].second;
            while (idx > 0) {
                idx--;
                const int nx = path[(size_t)idx].first, ny = path[(size_t)idx].second;
                if (cx == nx && cy == ny) continue;
                if (nx == cx && ny != cy) {
                    out.q.append((size_t)(ny - cy), '-');
                    out.t.append(target, (size_t)cy, (size_t)(ny - cy));
                } else if (nx != cx && ny == cy) {
                    out.q.append(query, (size_t)cx, (size_t)(nx - cx));
                    out.t.append((size_t)(nx - cx), '-');
                } else {
                    out.q.append(query, (size_t)cx, (size_t)(nx - cx));
                    out.t.append(target, (size_t)cy, (size_t)(ny - cy));
                }
                cx = nx; cy = ny;
            }
            break;
        }
    }
    return out;
}

// ---- falcon: alignment tags (falcon.c:68-125) --------------------------------------------------------------------------------
struct Tag { int t_pos, delta, p_t_pos, p_delta; char p_q_base, q_base; bool set; };

static std::vector<Tag> falcon_tags(const std::string& q, const std::string& t) {   // range s1 = s2 = 0, t_offset = 0
    std::vector<Tag> tags(q.size());
    int i = -1, j = -1, jj = 0, p_j = -1, p_jj = 0;
    char p_q_base = '.';
    for (size_t k = 0; k < q.size(); k++) {
        if (q[k] != '-') { i++; jj++; }
        if (t[k] != '-') { j++; jj = 0; }
        Tag& g = tags[k];
        g.set = false;
        if (j >= 0 && jj < 255 && p_jj < 255) {
            g.t_pos = j; g.delta = jj; g.p_t_pos = p_j; g.p_delta = p_jj; g.p_q_base = p_q_base; g.q_base = q[k]; g.set = true;
            p_j = j; p_jj = jj; p_q_base = q[k];
        }
    }
    (void)i;
    return tags;
}

// ---- falcon: consensus from the tags of all members (falcon.c:246-517) ---------------------------------------------------------
struct Col {
    int count = 0;
    std::vector<int> p_t_pos, p_delta, link_count;
    std::vector<char> p_q_base;
    int best_p_t_pos = 0, best_p_delta = 0, best_p_q_base = 0;
    double score = 0;
};
static int base_of(char c) {
    switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; case '-': return 4; }
    return -1;
}

// returns false where the reference indexes out of bounds (an unset tag: a run of 255+ inserted bases)
static bool falcon_cns(const std::vector<std::vector<Tag>>& seqs, int t_len, unsigned min_cov, std::string& out) {
    std::vector<unsigned> coverage((size_t)t_len, 0);
    std::vector<std::vector<std::vector<Col>>> msa((size_t)t_len);   // [t_pos][delta][base]
    int t_pos = 0;
    for (const auto& tg : seqs)
        for (const Tag& c : tg) {
            if (!c.set) return false;
            const int delta = c.delta;
            if (delta == 0) { t_pos = c.t_pos; coverage[(size_t)t_pos]++; }
            auto& pos = msa[(size_t)t_pos];
            if ((int)pos.size() < delta + 1) pos.resize((size_t)delta + 1, std::vector<Col>(5));
            const int base = base_of(c.q_base);
            if (base < 0) return false;
            Col& col = pos[(size_t)delta][(size_t)base];
            col.count++;
            size_t kk = 0;
            for (; kk < col.p_t_pos.size(); kk++)
                if (c.p_t_pos == col.p_t_pos[kk] && c.p_delta == col.p_delta[kk] && c.p_q_base == col.p_q_base[kk]) { col.link_count[kk]++; break; }
            if (kk == col.p_t_pos.size()) {
                col.p_t_pos.push_back(c.p_t_pos); col.p_delta.push_back(c.p_delta); col.p_q_base.push_back(c.p_q_base); col.link_count.push_back(1);
            }
        }
    Col* g_best = nullptr;
    unsigned g_best_ck = 0;
    int g_best_t_pos = 0;
    {
        int best_ck = -1;
        double g_best_score = -1;
        for (int i = 0; i < t_len; i++) {
            auto& pos = msa[(size_t)i];
            if (pos.empty()) pos.resize(1, std::vector<Col>(5));     // (max_delta = 0: the five columns of delta 0 exist, empty)
            for (size_t j = 0; j < pos.size(); j++)
                for (int kk = 0; kk < 5; kk++) {
                    Col& col = pos[j][(size_t)kk];
                    double best_score = -1;
                    for (size_t ck = 0; ck < col.p_t_pos.size(); ck++) {
                        const int pi = col.p_t_pos[ck], pj = col.p_delta[ck];
                        int pkk = base_of(col.p_q_base[ck]);
                        if (pkk < 0) pkk = 4;
                        double score;
                        if (pi == -1) score = (double)col.link_count[ck] - (double)coverage[(size_t)i] * 0.5;
                        else score = msa[(size_t)pi][(size_t)pj][(size_t)pkk].score + (double)col.link_count[ck] - (double)coverage[(size_t)i] * 0.5;
                        if (score > best_score) {
                            best_score = score;
                            col.best_p_t_pos = pi; col.best_p_delta = pj; col.best_p_q_base = pkk;
                            best_ck = (int)ck;
                        }
                    }
                    col.score = best_score;
                    if (best_score > g_best_score) {
                        g_best_score = best_score;
                        g_best = &col;
                        g_best_ck = (unsigned)best_ck;
                        g_best_t_pos = i;
                    }
                }
        }
        if (!g_best) return false;          // (assert(g_best_score != -1))
    }
    out.clear();
    unsigned index = 0;
    char bb = '$';
    int ck = (int)g_best_ck;
    int i = g_best_t_pos;
    while (1) {
        static const char up[5] = {'A', 'C', 'G', 'T', '-'}, lo[5] = {'a', 'c', 'g', 't', '-'};
        if (ck >= 0 && ck < 5) bb = coverage[(size_t)i] > min_cov ? up[ck] : lo[ck];
        i = g_best->best_p_t_pos;
        if (i == -1 || index >= (unsigned)t_len * 2) break;
        const int j = g_best->best_p_delta;
        ck = g_best->best_p_q_base;
        g_best = &msa[(size_t)i][(size_t)j][(size_t)ck];
        if (bb != '-') { out.push_back(bb); index++; }
    }
    std::reverse(out.begin(), out.end());
    return true;
}

static void upper_in_place(std::string& s) { for (char& c : s) c = (char)toupper((unsigned char)c); }

// one ladder as draft.cpp:597-691 runs it: member mx is the template, every member (mx included) is aligned to it
static bool falcon_ladder(const std::vector<std::string>& members, int mx, std::string& out) {
    const std::string& aseq = members[(size_t)mx];
    const int alen = (int)aseq.size();
    std::vector<std::vector<Tag>> tags;
    for (const std::string& bseq : members) {
        FAlign al = falcon_align(bseq, aseq, 150);
        std::string q = "T" + al.q, t = "T" + al.t;
        upper_in_place(q); upper_in_place(t);
        tags.push_back(falcon_tags(q, t));
    }
    return falcon_cns(tags, alen + 1, 1, out);
}

static std::string reverse_complement(const std::string& s) {   // draft.cpp:91-99
    std::string r(s.rbegin(), s.rend());
    for (char& c : r)
        switch (c) {
            case 'a': c = 't'; break; case 'c': c = 'g'; break; case 'g': c = 'c'; break; case 't': c = 'a'; break;
            case 'A': c = 'T'; break; case 'C': c = 'G'; break; case 'G': c = 'C'; break; case 'T': c = 'A'; break;
            case 'n': case 'N': case '-': break;
            default: c = '\0';            // (std::map::operator[] of a missing key: a NUL)
        }
    return r;
}

struct Ovl {     // LOverlap as getOverlap(range) fills it (B on its forward strand) + the LAlignment view of the same record
    int a, b, comp, ab, ae, bb, be, alen, blen;
    size_t full;  // index into the Aln list
};
struct Edge { int a, sa, b, sb, w; };

struct DraftRun {
    SeqDB db;
    std::vector<std::string> bases;     // lower case, as getRead delivers them
    std::vector<Aln> full;              // LAlignment list (raw coordinates, trace points)
    std::vector<Ovl> ovl;               // LOverlap list, same order
    std::unordered_map<int, std::vector<size_t>> by_a;                               // idx3 / idx_aln: record indices per A read
    std::unordered_map<int, std::unordered_map<int, std::vector<size_t>>> by_ab;      // idx
    int tspace_las = 100;
    int TSPACE = 0, EDGE_SAFE = 0, MIN_COV2 = 0;
    Waves waves;
    Out o;
};

static const std::string& read_bases(DraftRun& R, int id) { return R.bases[(size_t)id]; }

// draft_assembly_ctg (draft.cpp:125-715).  Returns the reference's return value (1, 2, 0, -1) or <= -10 where it is undefined.
static int draft_ctg(DraftRun& R, const std::vector<Edge>& edges, int cut_start, int cut_end, bool one_read, bool two_read, std::string& contig) {
    Out& o = R.o;
    o.pf("list size:%lu\n", (unsigned long)edges.size());
    if (edges.empty()) return -1;
    std::string draft;
    const Edge& e0 = edges[0];
    if (e0.a < 0 || e0.a >= (int)R.bases.size() || e0.b < 0 || e0.b >= (int)R.bases.size()) return -10;
    if (one_read) {
        draft = e0.sa == 0 ? read_bases(R, e0.a) : reverse_complement(read_bases(R, e0.a));
        o.pf("%d %d %d\n", cut_start, cut_end, R.db.rlen[(size_t)e0.a]);
        if ((size_t)cut_start <= draft.size() && (size_t)cut_end <= draft.size()) contig = draft.substr((size_t)cut_start, (size_t)(cut_end - cut_start));
        return 1;
    }
    for (const Edge& e : edges) if (e.a < 0 || e.a >= (int)R.bases.size() || e.b < 0 || e.b >= (int)R.bases.size()) return -10;
    // selected: the FIRST alignment of A with this B and this length (draft.cpp:162-178)
    std::vector<size_t> selected;
    for (const Edge& e : edges) {
        auto it = R.by_a.find(e.a);
        if (it == R.by_a.end()) continue;
        for (size_t r : it->second) {
            const Aln& al = R.full[R.ovl[r].full];
            if (al.b == e.b && al.ae - al.ab + al.be - al.bb == e.w) { selected.push_back(r); break; }
        }
    }
    o.pf("selected:%lu\n", (unsigned long)selected.size());
    if (selected.size() != edges.size()) return -11;     // the reference indexes its lists by edge number from here on
    if (two_read) {
        draft = e0.sa == 0 ? read_bases(R, e0.a) : reverse_complement(read_bases(R, e0.a));
        const Aln& s0 = R.full[R.ovl[selected[0]].full];
        const int aend = s0.ae, bstart = s0.bb;
        const std::string readB = e0.sb == 0 ? read_bases(R, e0.b) : reverse_complement(read_bases(R, e0.b));
        o.pf("alen blen aend bstart%d %d %d %d\n", R.db.rlen[(size_t)e0.a], R.db.rlen[(size_t)e0.b], aend, bstart);
        draft = draft.substr(0, (size_t)aend);
        if ((size_t)bstart > readB.size()) return -12;
        draft += readB.substr((size_t)bstart);
        o.pf("%d %d %d\n", cut_start, cut_end, R.db.rlen[(size_t)e0.a]);
        if ((size_t)cut_start <= draft.size() && (size_t)cut_end <= draft.size()) contig = draft.substr((size_t)cut_start, (size_t)(cut_end - cut_start));
        return 2;
    }
    const size_t n = edges.size();
    std::vector<std::pair<std::string, std::string>> tag_list(n), tag_true(n);
    for (size_t i = 0; i < n; i++) {
        Aln& al = R.full[R.ovl[selected[i]].full];
        if (al.trace.empty() && !al.pts.empty()) recover(R.db, R.db, al, R.tspace_las, R.waves);   // (recoverAlignment is idempotent: `recovered`)
        tags(R.db, R.db, al, tag_list[i].first, tag_list[i].second);
    }
    // coverage of every backbone read from its pile-up, forward frame (getCoverage, LAInterface.cpp:4254-4263)
    std::vector<std::vector<int>> coverages;
    for (size_t i = 0; i < n; i++) {
        auto it = R.by_a.find(edges[i].a);
        if (it == R.by_a.end() || it->second.empty()) continue;
        std::vector<int> cov((size_t)R.ovl[it->second[0]].alen, 0);
        for (size_t r : it->second)
            for (int j = R.ovl[r].ab; j < R.ovl[r].ae; j++) cov[(size_t)j]++;
        coverages.push_back(std::move(cov));
    }
    if (coverages.size() != n) return -13;
    struct BEdge { int as, ae, bs, be, alen, blen; };
    std::vector<BEdge> bedges(n);
    std::vector<std::string> breads(n);
    std::string overhang;
    int len_overhang = 0;
    for (size_t i = 0; i < n; i++) {
        const Edge& e = edges[i];
        const Ovl* cur = nullptr;
        auto ia = R.by_ab.find(e.a);
        if (ia != R.by_ab.end()) {
            auto ib = ia->second.find(e.b);
            if (ib != ia->second.end())
                for (size_t r : ib->second)
                    if (R.ovl[r].ae - R.ovl[r].ab + R.ovl[r].be - R.ovl[r].bb == e.w) cur = &R.ovl[r];      // the LAST one that fits
        }
        if (!cur) return -100;     // exit(1) in the reference
        breads[i] = e.sa == 0 ? read_bases(R, e.a) : reverse_complement(read_bases(R, e.a));
        if (e.sa == 0) tag_true[i] = tag_list[i];
        else tag_true[i] = {reverse_complement(tag_list[i].first), reverse_complement(tag_list[i].second)};
        const std::string next_seq = e.sb == 0 ? read_bases(R, e.b) : reverse_complement(read_bases(R, e.b));
        BEdge& b = bedges[i];
        b.alen = cur->alen; b.blen = cur->blen;
        if (e.sa == 0) { b.as = cur->ab; b.ae = cur->ae; } else { b.as = b.alen - cur->ae; b.ae = b.alen - cur->ab; }
        if (e.sb == 0) { b.bs = cur->bb; b.be = cur->be; } else { b.bs = b.blen - cur->be; b.be = b.blen - cur->bb; }
        overhang = next_seq;
        len_overhang = b.blen - b.be - (b.alen - b.ae);
    }
    if (len_overhang > 0 && (size_t)len_overhang < overhang.size()) overhang = overhang.substr(overhang.size() - (size_t)len_overhang);
    else overhang = "";
    // get_mapping (draft.cpp:70-87): for every A base of the alignment the number of B bases in front of its column
    std::vector<std::vector<int>> mappings(n);
    for (size_t i = 0; i < n; i++) {
        const std::string& t1 = tag_true[i].first; const std::string& t2 = tag_true[i].second;
        int count2 = 0;
        for (size_t p = 0; p < t1.size(); p++) {
            if (t1[p] != '-') mappings[i].push_back(count2);
            if (t2[p] != '-') count2++;
        }
    }
    o.pf("%lu %lu %lu %lu %lu %lu %lu %lu\n", (unsigned long)n, (unsigned long)n, (unsigned long)n, (unsigned long)n, (unsigned long)coverages.size(),
         (unsigned long)n, (unsigned long)n, (unsigned long)coverages.size());
    // ---- lanes: way points every TSPACE bases, carried from read to read through the mappings (draft.cpp:415-495) ------------
    const int ts = R.TSPACE;
    std::vector<std::vector<std::pair<int, int>>> lanes;
    std::vector<std::vector<int>> trace_pts(n);
    {
        int start_read = 0, space = 1, offset = 0, rmax = -1;
        const int nb = (int)n;
        while (start_read < nb - 1) {
            int cur = start_read;
            while (bedges[(size_t)start_read].as + space * ts + offset < bedges[(size_t)start_read].ae - R.EDGE_SAFE) {
                int way = bedges[(size_t)start_read].as + ts * space + offset;
                std::vector<std::pair<int, int>> lane;
                while (way > bedges[(size_t)cur].as && way < bedges[(size_t)cur].ae) {
                    trace_pts[(size_t)cur].push_back(way);
                    lane.push_back({cur, way});
                    if (cur > rmax) rmax = cur;
                    const int at = way - bedges[(size_t)cur].as;
                    if (at < 0 || at >= (int)mappings[(size_t)cur].size()) return -14;
                    way = mappings[(size_t)cur][(size_t)at] + bedges[(size_t)cur].bs;
                    cur++;
                    if (cur >= nb) break;
                }
                if (cur < nb && way < bedges[(size_t)cur].alen) {
                    lane.push_back({cur, way});
                    if (cur > rmax) rmax = cur;
                }
                if (cur >= rmax) lanes.push_back(lane);
                space++;
                cur = start_read;
            }
            start_read++;
            space = 1;
            offset = trace_pts[(size_t)start_read].empty() ? 0 : trace_pts[(size_t)start_read].back() - bedges[(size_t)start_read].as;
        }
    }
    for (size_t i = 0; i < n; i++) {
        o.pf("Read %d:", (int)i);
        for (int w : trace_pts[i]) o.pf("%d ", w);
        o.pf("\n");
    }
    for (size_t i = 0; i < lanes.size(); i++) {
        o.pf("Lane %d\n", (int)i);
        for (auto& p : lanes[i]) o.pf("[%d %d] ", p.first, p.second);
        o.pf("\n");
    }
    o.pf("In total %lu lanes\n", (unsigned long)lanes.size());
    if (lanes.empty() || lanes[0].empty() || lanes.back().empty()) return -15;
    const int first_start = lanes[0][0].second, last_end = lanes.back().back().second;
    const Edge& el = edges.back();
    o.pf("first %d last %d\n", first_start, last_end);
    o.pf("len %d %d\n", R.db.rlen[(size_t)e0.a], R.db.rlen[(size_t)el.b]);
    if (!(first_start <= R.db.rlen[(size_t)e0.a]) || !(last_end <= R.db.rlen[(size_t)el.a])) return -16;    // the two assert()s
    if (first_start < 0 || last_end < 0) return -16;
    const std::string prefix = read_bases(R, e0.a).substr(0, (size_t)first_start);
    const std::string suffix = read_bases(R, el.a).substr((size_t)last_end);
    o.pf("last read %d length %d, cut %d\n", el.b, R.db.rlen[(size_t)el.b], cut_end);
    cut_end = R.db.rlen[(size_t)el.b] - cut_end;
    // ---- ladders: what two consecutive lanes share (draft.cpp:540-556) -------------------------------------------------------
    std::string body;
    for (size_t i = 0; i + 1 < lanes.size(); i++) {
        const auto& l1 = lanes[i]; const auto& l2 = lanes[i + 1];
        struct Rung { int read, start, end; };
        std::vector<Rung> ladder;
        size_t pos = 0;
        for (size_t j = 0; j < l2.size(); j++) {
            while (l1[pos].first != l2[j].first && pos < l1.size() - 1) pos++;
            if (l1[pos].first == l2[j].first) ladder.push_back({l2[j].first, l1[pos].second, l2[j].second});
        }
        if (ladder.empty()) { o.pf("low coverage!\n"); continue; }
        for (const Rung& g : ladder)
            if (g.start < 0 || g.end < g.start || (size_t)g.end > breads[(size_t)g.read].size()) return -17;   // (substr / strcpy overruns in the reference)
        if (ladder.size() > 1) {
            int mx = 0, maxcoverage = 0;
            for (size_t j = 0; j < ladder.size(); j++) {
                int mincoverage = 10000;
                const std::vector<int>& cov = coverages[(size_t)ladder[j].read];
                for (int p = ladder[j].start; p < ladder[j].end; p++) {
                    if ((size_t)p >= cov.size()) return -18;       // (.at() throws)
                    if (cov[(size_t)p] < mincoverage) mincoverage = cov[(size_t)p];
                }
                if (mincoverage > maxcoverage) { maxcoverage = mincoverage; mx = (int)j; }
            }
            std::vector<std::string> members;
            for (const Rung& g : ladder) members.push_back(breads[(size_t)g.read].substr((size_t)g.start, (size_t)(g.end - g.start)));
            std::string cns;
            if (!falcon_ladder(members, mx, cns)) return -19;
            body += cns;
        } else {
            body += breads[(size_t)ladder[0].read].substr((size_t)ladder[0].start, (size_t)(ladder[0].end - ladder[0].start));
        }
    }
    o.pf("0\n%lu\n", (unsigned long)body.size());
    contig = prefix + body + suffix + overhang;
    o.pf("ctg size:%lucut_start:%dcut_end:%d\n", (unsigned long)contig.size(), cut_start, cut_end);
    if ((size_t)cut_start <= contig.size() && (size_t)cut_end <= contig.size())
        contig = contig.substr((size_t)cut_start, contig.size() - (size_t)cut_end - (size_t)cut_start);
    return 0;
}

static std::vector<std::string> split_ws(const std::string& s) {   // split(s, ' ') of draft.cpp:102-118: empty items between two blanks
    std::vector<std::string> elems;
    std::stringstream ss(s);
    std::string item;
    while (std::getline(ss, item, ' ')) elems.push_back(item);
    return elems;
}

}  // namespace

extern "C" {

// Test hook: one ladder through the falcon restatement.  seqs = n NUL-terminated member strings (lower or upper case acgt),
// mx = the template member.  Returns the consensus length (written to out, NUL-terminated, if it fits cap), -1 on undefined input.
long oracle_falcon_ladder(int n, const char** seqs, int mx, char* out, long cap) {
    std::vector<std::string> members;
    for (int i = 0; i < n; i++) members.push_back(seqs[i]);
    std::string cns;
    if (n < 1 || mx < 0 || mx >= n || !falcon_ladder(members, mx, cns)) return -1;
    if ((long)cns.size() + 1 <= cap) memcpy(out, cns.c_str(), cns.size() + 1);
    return (long)cns.size();
}

// Test hook: the two gapped rows of falcon's aligner for one pair; returns the row length or -1 when it does not align.
long oracle_falcon_align(const char* query, const char* target, int band, char* q_out, char* t_out, long cap) {
    FAlign al = falcon_align(query, target, band);
    if (!al.aligned) return -1;
    if ((long)al.q.size() + 1 <= cap) { memcpy(q_out, al.q.c_str(), al.q.size() + 1); memcpy(t_out, al.t.c_str(), al.t.size() + 1); }
    return (long)al.q.size();
}

// `draft_assembly --db D --las L [--mlas] -x PREFIX -o OUT --config INI` (draft.cpp:720-1162).
// 0 ok; 1 = "No alignments!" / unreadable config (the reference's return 1); -1 unreadable DB / .las / .edges.list; <= -10: input on
// which the reference is undefined or aborts (-10 read id out of range, -11 an edge without its alignment, -12 .. -19 see
// draft_ctg, -20 a malformed .edges.list line, -100 its exit(1)).  log_path (may be NULL): what the reference prints on stdout
// (without the logger's lines).
int oracle_draft(const char* name_db, const char* name_las, int mlas, const char* prefix, const char* out_name, const char* name_config, const char* log_path) {
    DraftRun R;
    if (log_path) R.o.log = fopen(log_path, "w");
    struct Closer { Out& o; ~Closer() { if (o.log) fclose(o.log); } } closer{R.o};
    const std::string out = prefix, outn = out_name;
    // (the three std::ofstream the reference opens and never writes)
    for (const std::string& p : {outn + ".deadends.txt", out + ".garbage.txt", out + ".contained.txt"}) { FILE* f = fopen(p.c_str(), "w"); if (f) fclose(f); }
    if (open_seq_db(name_db, R.db) != 0) return -1;
    const int n_read = (int)R.db.rlen.size();
    R.bases.resize((size_t)n_read);
    for (int i = 0; i < n_read; i++) {
        std::string& s = R.bases[(size_t)i];
        s.resize((size_t)R.db.rlen[(size_t)i]);
        for (int p = 0; p < R.db.rlen[(size_t)i]; p++) s[(size_t)p] = "acgt"[R.db.base(i, p)];
    }
    std::vector<char> active((size_t)n_read, 0);
    {
        FILE* f = fopen((out + ".max").c_str(), "r");
        if (f) {
            char line[4096];
            while (fgets(line, sizeof line, f)) {
                const int r = atoi(line);
                if (r < 0 || r >= n_read) { fclose(f); return -10; }
                active[(size_t)r] = 1;
            }
            fclose(f);
        }
    }
    std::vector<std::string> parts;
    if (mlas) parts = oracle::las_parts(name_las); else parts.push_back(oracle::las_name(name_las, false));
    int64_t n_aln = 0;
    for (const std::string& p : parts) {
        std::vector<Aln> recs;
        int ts = 0; int64_t novl = 0;
        FILE* f = fopen(p.c_str(), "rb");
        if (!f) return -1;
        if (fread(&novl, 8, 1, f) != 1 || fread(&ts, 4, 1, f) != 1) { fclose(f); return -1; }
        R.tspace_las = ts;
        n_aln += novl;
        const int tbytes = ts <= 125 ? 1 : 2;
        std::vector<unsigned char> tbuf;
        for (int64_t j = 0; j < novl; j++) {
            int32_t r[10];
            if (fread(r, 40, 1, f) != 1) break;
            const int tlen = r[0];
            tbuf.resize((size_t)tlen * tbytes);
            if (tlen > 0 && fread(tbuf.data(), (size_t)tlen * tbytes, 1, f) != 1) break;
            if (r[7] < 0 || r[7] >= n_read || r[8] < 0 || r[8] >= n_read) { fclose(f); return -10; }
            if (!active[(size_t)r[7]] || !active[(size_t)r[8]]) continue;      // A in `range` (= the active reads) and both active
            Aln al;
            al.a = r[7]; al.b = r[8]; al.alen = R.db.rlen[(size_t)al.a]; al.blen = R.db.rlen[(size_t)al.b];
            al.comp = (r[6] & 1) ? 1 : 0;
            al.ab = r[2]; al.bb = r[3]; al.ae = r[4]; al.be = r[5];
            al.pts.resize((size_t)tlen);
            for (int k = 0; k < tlen; k++) al.pts[(size_t)k] = tbytes == 1 ? tbuf[(size_t)k] : (uint16_t)(tbuf[2 * (size_t)k] | (tbuf[2 * (size_t)k + 1] << 8));
            Ovl v;
            v.a = al.a; v.b = al.b; v.comp = al.comp; v.ab = al.ab; v.ae = al.ae; v.alen = al.alen; v.blen = al.blen;
            if (al.comp) { v.bb = al.blen - al.be; v.be = al.blen - al.bb; } else { v.bb = al.bb; v.be = al.be; }
            v.full = R.full.size();
            R.full.push_back(std::move(al));
            R.ovl.push_back(v);
        }
        fclose(f);
    }
    if (n_aln == 0) return 1;
    Ini ini(name_config);
    if (ini.error < 0) return 1;
    R.MIN_COV2 = (int)ini.get_int("draft", "min_cov", -1);
    R.EDGE_SAFE = (int)ini.get_int("draft", "edge_safe", -1);
    R.TSPACE = (int)ini.get_int("draft", "tspace", -1);
    for (size_t r = 0; r < R.ovl.size(); r++) {
        R.by_a[R.ovl[r].a].push_back(r);
        R.by_ab[R.ovl[r].a][R.ovl[r].b].push_back(r);
    }
    R.o.pf("add data\nadd data\n");
    // ---- .edges.list: read twice (draft.cpp:1057-1076 only echoes it) ----------------------------------------------------------
    std::vector<std::string> lines;     // getline()'s results up to and including the one that hits end of file
    {
        FILE* f = fopen((out + ".edges.list").c_str(), "rb");
        std::string all;
        if (f) { char buf[65536]; size_t g; while ((g = fread(buf, 1, sizeof buf, f)) > 0) all.append(buf, g); fclose(f); }
        size_t at = 0;
        if (f)
            while (true) {
                const size_t nl = all.find('\n', at);
                if (nl == std::string::npos) { lines.push_back(all.substr(at)); break; }    // this getline sets eof
                lines.push_back(all.substr(at, nl - at));
                at = nl + 1;
            }
        else lines.push_back("");   // (an unopenable file: the first getline fails; eof() is not set but the loop below ends the same way)
    }
    for (const std::string& l : lines) {
        R.o.pf("%s\n", l.c_str());
        if (l.empty() || l[0] == '>') continue;
        const std::vector<std::string> tok = split_ws(l);
        if (tok.size() < 6) R.o.pf("Error! Wrong format.\n");
        if (tok.size() < 4) return -20;
    }
    FILE* fa = fopen((outn + ".fasta").c_str(), "w");
    if (!fa) return -1;
    struct FaCloser { FILE* f; ~FaCloser() { fclose(f); } } fac{fa};
    std::vector<Edge> edgelist;
    std::string current_name, contig;
    bool one_read = false, two_read = false;
    int cut_start = 0, cut_end = 0;
    for (size_t li = 0; li < lines.size(); li++) {
        const std::string& l = lines[li];
        const bool eof = li + 1 == lines.size();
        if (!l.empty() && l[0] == '>') {
            R.o.pf("%s\n", current_name.c_str());
            if (!edgelist.empty()) {
                const int rc = draft_ctg(R, edgelist, cut_start, cut_end, one_read, two_read, contig);
                if (rc <= -10) return rc;
                fprintf(fa, "%s\n%s\n", current_name.c_str(), contig.c_str());
            }
            edgelist.clear();
            current_name = l;
            one_read = two_read = false;
            cut_start = cut_end = 0;
            continue;
        }
        if (eof) {
            R.o.pf("%s\n", current_name.c_str());
            const int rc = draft_ctg(R, edgelist, cut_start, cut_end, one_read, two_read, contig);
            if (rc <= -10) return rc;
            fprintf(fa, "%s\n%s\n", current_name.c_str(), contig.c_str());
            edgelist.clear();
            continue;
        }
        const std::vector<std::string> tok = split_ws(l);
        if (tok.size() < 6) R.o.pf("Error! Wrong format.\n");
        R.o.pf("%s\n", l.c_str());
        const size_t need = tok.empty() ? 6 : (tok[0] == "O" || tok[0] == "S" || tok[0] == "E" ? 7 : tok[0] == "D" ? 8 : 6);
        if (tok.size() < need) return -20;
        Edge e;
        try {
            e.a = std::stoi(tok[1]); e.sa = std::stoi(tok[2]); e.b = std::stoi(tok[3]); e.sb = std::stoi(tok[4]);
            if (tok[0] == "O") { e.w = 0; one_read = true; }
            else if (tok[0] == "D") { e.w = std::stoi(tok[5]); two_read = true; }
            else e.w = std::stoi(tok[5]);
            edgelist.push_back(e);
            if (tok[0] == "O") { cut_start = std::stoi(tok[5]); cut_end = std::stoi(tok[6]); }
            else if (tok[0] == "S") cut_start = std::stoi(tok[6]);
            else if (tok[0] == "E") cut_end = std::stoi(tok[6]);
            else if (tok[0] == "D") { cut_start = std::stoi(tok[6]); cut_end = std::stoi(tok[7]); }
        } catch (...) { return -20; }
    }
    return 0;
}

}  // extern "C"
