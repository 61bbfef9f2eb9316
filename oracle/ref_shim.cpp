// TEST INFRASTRUCTURE - NOT PRODUCT CODE.
// C-ABI shim over the REFERENCE's own compiled library code (LAInterface.cpp, DB.c, align.c, ini.c,
// INIReader.cpp, compiled unmodified from /root/reference/src by oracle/Makefile into oracle/_ref/).
// It only marshals flat arrays into the reference's classes and calls the reference's functions, so
// that tests can pin oracle/ (the restatement) against what the reference itself computes:
//   LAInterface::openDB / getReadNumber / openAlignmentFile / resetAlignment / getOverlap / getQV
//   LAInterface::profileCoverage, LOverlap::trim_overlap, LOverlap::AddTypesAsymmetric,
//   LOverlap::GetMatchingPosition, compare_overlap, compare_overlap_weight, pairAscend, pairDescend,
//   INIReader.
// ProcessAlignment itself lives in maximal.cpp / hinging.cpp (not buildable here: spdlog/Boost are
// absent); its 20-line body is restated below around the real trim_overlap/AddTypesAsymmetric.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>
#include <unordered_map>
#include "LAInterface.h"
#include "INIReader.h"
#include <cctype>
extern "C" {
#include "common.h"
}

extern "C" {

long ref_load_las(const char* name_db, const char* las_path, int* out, long cap) {
    LAInterface la;
    la.openDB(name_db);
    int n_read = la.getReadNumber();
    la.openAlignmentFile(las_path);
    la.resetAlignment();
    std::vector<LOverlap*> aln;
    la.getOverlap(aln, 0, n_read);
    long n = (long)aln.size();
    for (long i = 0; i < n && i < cap; i++) {
        LOverlap* o = aln[i];
        int* p = out + i * 8;
        p[0] = o->read_A_id_; p[1] = o->read_B_id_; p[2] = o->read_A_match_start_; p[3] = o->read_A_match_end_;
        p[4] = o->read_B_match_start_; p[5] = o->read_B_match_end_; p[6] = o->reverse_complement_match_; p[7] = o->trace_pts_len;
    }
    for (auto o : aln) delete o;
    return n;
}

int ref_tspace(const char* las_path) {
    LAInterface la;
    la.openAlignmentFile(las_path);
    return la.tspace;
}

int ref_read_lengths(const char* name_db, int* out, int cap) {
    LAInterface la;
    la.openDB(name_db);
    int n = la.getReadNumber();
    for (int i = 0; i < n && i < cap; i++) out[i] = la.db1->reads[i].rlen;
    return n;
}

// getQV: returns -1 if no track, else total byte count; offsets[n+1], values[cap]
long ref_qv(const char* name_db, long* offsets, int* values, long cap) {
    LAInterface la;
    la.openDB(name_db);
    int n = la.getReadNumber();
    std::vector<std::vector<int>> QV;
    if (la.getQV(QV, 0, n) != 0) return -1;
    long k = 0;
    for (int i = 0; i < n; i++) {
        offsets[i] = k;
        for (size_t j = 0; j < QV[i].size(); j++) { if (k < cap) values[k] = QV[i][j]; k++; }
    }
    offsets[n] = k;
    return k;
}

int ref_profile_coverage(int n, const int* ab, const int* ae, int reso, int cutoff, int* cov_out, int cap) {
    LAInterface la;
    std::vector<LOverlap*> v;
    for (int i = 0; i < n; i++) {
        LOverlap* o = new LOverlap();
        o->trace_pts = NULL;
        o->read_A_match_start_ = ab[i];
        o->read_A_match_end_ = ae[i];
        v.push_back(o);
    }
    std::vector<std::pair<int, int>> c;
    la.profileCoverage(v, c, reso, cutoff);
    for (int i = 0; i < (int)c.size() && i < cap; i++) cov_out[i] = c[i].second;
    for (auto o : v) delete o;
    return (int)c.size();
}

static LOverlap* make_ovl(int ab, int ae, int bb, int be, int comp, const uint16_t* trace, int tlen) {
    LOverlap* o = new LOverlap();
    o->read_A_match_start_ = ab; o->read_A_match_end_ = ae;
    o->read_B_match_start_ = bb; o->read_B_match_end_ = be;
    o->reverse_complement_match_ = comp;
    o->trace_pts_len = tlen;
    o->trace_pts = (uint16*)malloc(sizeof(uint16) * (tlen > 0 ? tlen : 1));
    memcpy(o->trace_pts, trace, sizeof(uint16) * tlen);
    return o;
}

void ref_process_alignment(const int* in, const uint16_t* trace, int tlen, int aln_threshold, int theta, int theta2, int* out) {
    LOverlap* m = make_ovl(in[0], in[1], in[2], in[3], in[4], trace, tlen);
    // body of ProcessAlignment(match, read_A, read_B, ALN_THRESHOLD, THETA, THETA2, trim=true),
    // maximal.cpp:65-134, around the reference's own trim_overlap / AddTypesAsymmetric
    m->eff_read_A_read_start_ = in[5]; m->eff_read_A_read_end_ = in[6];
    m->eff_read_B_read_start_ = in[7]; m->eff_read_B_read_end_ = in[8];
    m->trim_overlap();
    if (((m->eff_read_B_match_end_ - m->eff_read_B_match_start_) < aln_threshold) ||
        ((m->eff_read_A_match_end_ - m->eff_read_A_match_start_) < aln_threshold) || (!m->active)) {
        m->active = false;
        m->match_type_ = NOT_ACTIVE;
    } else {
        m->AddTypesAsymmetric(theta, theta2);
    }
    m->weight = m->eff_read_A_match_end_ - m->eff_read_A_match_start_ + m->eff_read_B_match_end_ - m->eff_read_B_match_start_;
    m->length = m->read_A_match_end_ - m->read_A_match_start_ + m->read_B_match_end_ - m->read_B_match_start_;
    out[0] = m->eff_read_A_match_start_; out[1] = m->eff_read_A_match_end_;
    out[2] = m->eff_read_B_match_start_; out[3] = m->eff_read_B_match_end_;
    out[4] = (int)m->match_type_; out[5] = m->active ? 1 : 0; out[6] = m->weight; out[7] = m->length;
    out[8] = m->eff_start_trace_point_index_; out[9] = m->eff_end_trace_point_index_;
    delete m;
}

// The whole .las through the reference's OWN reader and its own trim / classify: LAInterface::getOverlap parses the file (records,
// strand flip, trace points - LAInterface.cpp:1519-1634), then the body of ProcessAlignment (maximal.cpp:65-134) as above with the
// effective read bounds eff[n_reads][2] (the .mas file, maximal.cpp:524-531).  out[n][12] = A, B, then the ten fields of
// ref_process_alignment, one row per record in file order (self-overlaps included).  What tests/test_ref_direct_gpu.py holds
// k_trim_classify_image against: the same FILE on both sides, no array of ours in between.
long ref_process_las(const char* name_db, const char* las_path, const int* eff, int aln_threshold, int theta, int theta2, int* out, long cap) {
    LAInterface la;
    la.openDB(name_db);
    int n_read = la.getReadNumber();
    la.openAlignmentFile(las_path);
    la.resetAlignment();
    std::vector<LOverlap*> aln;
    la.getOverlap(aln, 0, n_read);
    long n = (long)aln.size();
    for (long i = 0; i < n && i < cap; i++) {
        LOverlap* m = aln[i];
        const int a = m->read_A_id_, b = m->read_B_id_;
        m->eff_read_A_read_start_ = eff[2 * a]; m->eff_read_A_read_end_ = eff[2 * a + 1];
        m->eff_read_B_read_start_ = eff[2 * b]; m->eff_read_B_read_end_ = eff[2 * b + 1];
        m->trim_overlap();
        if (((m->eff_read_B_match_end_ - m->eff_read_B_match_start_) < aln_threshold) ||
            ((m->eff_read_A_match_end_ - m->eff_read_A_match_start_) < aln_threshold) || (!m->active)) {
            m->active = false;
            m->match_type_ = NOT_ACTIVE;
        } else {
            m->AddTypesAsymmetric(theta, theta2);
        }
        m->weight = m->eff_read_A_match_end_ - m->eff_read_A_match_start_ + m->eff_read_B_match_end_ - m->eff_read_B_match_start_;
        m->length = m->read_A_match_end_ - m->read_A_match_start_ + m->read_B_match_end_ - m->read_B_match_start_;
        int* o = out + 12 * i;
        o[0] = a; o[1] = b;
        o[2] = m->eff_read_A_match_start_; o[3] = m->eff_read_A_match_end_;
        o[4] = m->eff_read_B_match_start_; o[5] = m->eff_read_B_match_end_;
        o[6] = (int)m->match_type_; o[7] = m->active ? 1 : 0; o[8] = m->weight; o[9] = m->length;
        o[10] = m->eff_start_trace_point_index_; o[11] = m->eff_end_trace_point_index_;
    }
    for (auto o : aln) delete o;
    return n;
}

// .coverage.txt of a .las as `hinge filter` prints it (filter.cpp:599-602), from the reference's own reader and its own
// profileCoverage: getOverlap over the file, the pile-ups of filter.cpp:529-548 (self-overlaps dropped), then for every read from the
// first to the last A read of the file "read i pos,cov pos,cov ... \n" of profileCoverage(pile-up, reso, cutoff 0).  What
// tests/test_ref_direct_gpu.py holds the file written by the GPU executable against, byte for byte.
int ref_coverage_txt_las(const char* name_db, const char* las_path, int reso, const char* out_path) {
    LAInterface la;
    la.openDB(name_db);
    int n_read = la.getReadNumber();
    la.openAlignmentFile(las_path);
    la.resetAlignment();
    std::vector<LOverlap*> aln;
    la.getOverlap(aln, 0, n_read);
    if (aln.empty()) return 1;
    const int r_begin = aln.front()->read_A_id_, r_end = aln.back()->read_A_id_;
    std::vector<std::vector<LOverlap*>> pile((size_t)n_read);
    for (auto o : aln)
        if (o->read_A_id_ != o->read_B_id_) pile[(size_t)o->read_A_id_].push_back(o);
    FILE* f = fopen(out_path, "w");
    if (!f) return 2;
    for (int i = r_begin; i <= r_end; i++) {
        std::vector<std::pair<int, int>> coverage;
        la.profileCoverage(pile[(size_t)i], coverage, reso, 0);
        fprintf(f, "read %d ", i);
        for (size_t j = 0; j < coverage.size(); j++) fprintf(f, "%d,%d ", coverage[j].first, coverage[j].second);
        fprintf(f, "\n");
    }
    fclose(f);
    for (auto o : aln) delete o;
    return 0;
}

// A TIMED slice of `hinge filter` through the reference's OWN code only (bench.py's cpu_baseline.reference_slice): what
// filter.cpp's main() does with library calls and its comparator, in its order, on one thread:
//   secs[0]  LAInterface::openDB + openAlignmentFile + getOverlap(aln, 0, n_read)            filter.cpp:474-512, LAInterface.cpp:1519-1634
//   secs[1]  the pile-up index (self-overlaps dropped) + std::sort(compare_overlap) per read   filter.cpp:527-548, 565-567
//   secs[2]  profileCoverage(pile-up, CUT_OFF) + profileCoverage(pile-up, 0) per read          filter.cpp:588-598, LAInterface.cpp:4298-4320
// OMITTED (inline code of main(), unbuildable here: spdlog): the idx_ab maps and the dedup pile-up (:569-583), the .coverage.txt text
// and the gradient (:599-610), median / MIN_COV (:642-678), the masks (:696-789), repeat annotation + merge (:796-865), hinge calling
// (:867-1068), the writers.  So secs[0..2] is a LOWER bound on the reference's filter time for this file.  counts = {records,
// reads with a pile-up, checksum of the coverage values (keeps the calls alive)}.
int ref_filter_slice(const char* name_db, const char* las_path, int reso, int cut_off, double* secs, long long* counts) {
    struct timespec t0, t1, t2, t3;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    LAInterface la;
    la.openDB(name_db);
    int n_read = la.getReadNumber();
    la.openAlignmentFile(las_path);
    la.resetAlignment();
    std::vector<LOverlap*> aln;
    la.getOverlap(aln, 0, n_read);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (aln.empty()) return 1;
    const int r_begin = aln.front()->read_A_id_, r_end = aln.back()->read_A_id_;
    std::vector<std::vector<LOverlap*>> idx_pileup((size_t)n_read);
    for (size_t i = 0; i < aln.size(); i++) {
        if (aln[i]->read_A_id_ == aln[i]->read_B_id_) aln[i]->active = false;
        if (aln[i]->active) idx_pileup[(size_t)aln[i]->read_A_id_].push_back(aln[i]);
    }
    for (int i = 0; i < n_read; i++) std::sort(idx_pileup[i].begin(), idx_pileup[i].end(), compare_overlap);
    clock_gettime(CLOCK_MONOTONIC, &t2);
    long long sum = 0, with_pile = 0;
    for (int i = r_begin; i <= r_end; i++) {
        std::vector<std::pair<int, int>> coverage, cutoff_coverage;
        la.profileCoverage(idx_pileup[(size_t)i], cutoff_coverage, reso, cut_off);
        la.profileCoverage(idx_pileup[(size_t)i], coverage, reso, 0);
        for (size_t j = 0; j < coverage.size(); j++) sum += coverage[j].second;
        for (size_t j = 0; j < cutoff_coverage.size(); j++) sum += 3 * cutoff_coverage[j].second;
        with_pile += idx_pileup[(size_t)i].empty() ? 0 : 1;
    }
    clock_gettime(CLOCK_MONOTONIC, &t3);
    auto dt = [](const timespec& a, const timespec& b) { return (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec); };
    secs[0] = dt(t0, t1); secs[1] = dt(t1, t2); secs[2] = dt(t2, t3);
    counts[0] = (long long)aln.size(); counts[1] = with_pile; counts[2] = sum;
    for (auto o : aln) delete o;
    return 0;
}

int ref_matching_position(int ab, int ae, int bb, int be, int comp, const uint16_t* trace, int tlen, int pos_A) {
    LOverlap* m = make_ovl(ab, ae, bb, be, comp, trace, tlen);
    int r = m->GetMatchingPosition(pos_A);
    delete m;
    return r;
}

// Batch forms of the three calls above (the -m gpu tests hold the HIP kernels against 10^5+ cases per primitive; one ctypes
// call per case would be the test's time).  Same reference functions, a loop around them.
// hdr[n][9] as for ref_process_alignment; trace = all traces back to back, toff[n + 1] in VALUES; out[n][10].
void ref_process_alignment_batch(long n, const int* hdr, const uint16_t* trace, const long* toff, int aln_threshold, int theta, int theta2, int* out) {
    for (long i = 0; i < n; i++)
        ref_process_alignment(hdr + 9 * i, trace + toff[i], (int)(toff[i + 1] - toff[i]), aln_threshold, theta, theta2, out + 10 * i);
}

// q queries: overlap qi[k] (rows of hdr[n][9], of which the first five are used) at position qpos[k]
void ref_matching_position_batch(long nq, const long* qi, const int* qpos, const int* hdr, const uint16_t* trace, const long* toff, int* out) {
    for (long k = 0; k < nq; k++) {
        const int* h = hdr + 9 * qi[k];
        out[k] = ref_matching_position(h[0], h[1], h[2], h[3], h[4], trace + toff[qi[k]], (int)(toff[qi[k] + 1] - toff[qi[k]]), qpos[k]);
    }
}

// n pile-ups: overlaps row_ptr[i] .. row_ptr[i + 1] of (ab, ae); nbins[i] = K, bins of pile-up i from cov_out[sum of K before]
// (cov_out == NULL: only nbins).  Returns the total number of bins.
long ref_profile_coverage_batch(long n, const long* row_ptr, const int* ab, const int* ae, int reso, int cutoff, int* nbins, int* cov_out, long cap) {
    LAInterface la;
    long tot = 0;
    for (long i = 0; i < n; i++) {
        std::vector<LOverlap*> v;
        for (long j = row_ptr[i]; j < row_ptr[i + 1]; j++) {
            LOverlap* o = new LOverlap();
            o->trace_pts = NULL;
            o->read_A_match_start_ = ab[j];
            o->read_A_match_end_ = ae[j];
            v.push_back(o);
        }
        std::vector<std::pair<int, int>> c;
        la.profileCoverage(v, c, reso, cutoff);
        nbins[i] = (int)c.size();
        if (cov_out)
            for (size_t k = 0; k < c.size(); k++)
                if (tot + (long)k < cap) cov_out[tot + (long)k] = c[k].second;
        tot += (long)c.size();
        for (auto o : v) delete o;
    }
    return tot;
}

// One ladder of `hinge draft` through the reference's OWN falcon code (lib/DW_banded.c _align, lib/falcon.c get_align_tags /
// get_cns_from_align_tags): the marshalling of draft.cpp:597-691 around them - member mx is the template, every member is aligned
// to it with band tolerance 150, both rows get a leading 'T' and go to upper case, consensus over alen + 1 positions with
// min_cov 1.  Returns the consensus length (NUL-terminated into out if it fits).
long ref_falcon_ladder(int n, const char** seqs, int mx, char* out, long cap) {
    const int alen = (int)strlen(seqs[mx]);
    align_tags_t** tags_list = (align_tags_t**)calloc((size_t)n, sizeof(align_tags_t*));
    for (int j = 0; j < n; j++) {
        const int blen = (int)strlen(seqs[j]);
        char* aseq = (char*)malloc((size_t)alen + 20);
        char* bseq = (char*)malloc((size_t)blen + 20);
        strcpy(aseq, seqs[mx]);
        strcpy(bseq, seqs[j]);
        aln_range* arange = (aln_range*)calloc(1, sizeof(aln_range));
        arange->s1 = 0; arange->e1 = (int)strlen(bseq); arange->s2 = 0; arange->e2 = (int)strlen(aseq); arange->score = 5;
        alignment* alng = _align(bseq, blen, aseq, alen, 150, 1);
        char* q = (char*)malloc(5 + strlen(alng->q_aln_str));
        char* t = (char*)malloc(5 + strlen(alng->t_aln_str));
        strcpy(q + 1, alng->q_aln_str);
        strcpy(t + 1, alng->t_aln_str);
        q[0] = 'T'; t[0] = 'T';
        for (size_t p = 0; p < strlen(q); p++) q[p] = (char)toupper(q[p]);
        for (size_t p = 0; p < strlen(t); p++) t[p] = (char)toupper(t[p]);
        tags_list[j] = get_align_tags(q, t, (seq_coor_t)strlen(alng->q_aln_str) + 1, arange, (unsigned)j, 0);
        free(q); free(t); free(aseq); free(bseq); free(arange);
        free_alignment(alng);
    }
    consensus_data* c = get_cns_from_align_tags(tags_list, (unsigned)n, (unsigned)alen + 1, 1);
    const long len = (long)strlen(c->sequence);
    if (len + 1 <= cap) memcpy(out, c->sequence, (size_t)len + 1);
    free_consensus_data(c);
    for (int j = 0; j < n; j++) free_align_tags(tags_list[j]);
    free(tags_list);
    return len;
}

// falcon's aligner alone: the two gapped rows; returns their length (0 when it does not align)
long ref_falcon_align(const char* query, const char* target, int band, char* q_out, char* t_out, long cap) {
    alignment* a = _align((char*)query, (seq_coor_t)strlen(query), (char*)target, (seq_coor_t)strlen(target), band, 1);
    const long len = (long)strlen(a->q_aln_str);
    if (len + 1 <= cap) { memcpy(q_out, a->q_aln_str, (size_t)len + 1); memcpy(t_out, a->t_aln_str, (size_t)len + 1); }
    free_alignment(a);
    return len;
}

// LAInterface::getCoverage (LOverlap form), LAInterface.cpp:4254-4263: cov[alen] from n (abpos, aepos)
void ref_get_coverage(int n, const int* ab, const int* ae, int alen, int* cov) {
    LAInterface la;
    std::vector<LOverlap*> v;
    for (int i = 0; i < n; i++) {
        LOverlap* o = new LOverlap();
        o->trace_pts = NULL;
        o->read_A_match_start_ = ab[i]; o->read_A_match_end_ = ae[i]; o->alen = alen;
        v.push_back(o);
    }
    std::vector<int>* r = la.getCoverage(v);
    for (int i = 0; i < alen; i++) cov[i] = (*r)[i];
    delete r;
    for (auto o : v) delete o;
}

// mode 0: compare_overlap on LOverlap* whose length sum is key[i]; mode 1: pairAscend; mode 2: pairDescend;
// mode 3: compare_overlap_weight
void ref_sort_perm(int n, const int* key, int mode, int* perm) {
    if (mode == 1 || mode == 2) {
        std::vector<std::pair<int, int>> v(n);
        for (int i = 0; i < n; i++) v[i] = std::pair<int, int>(key[i], i);
        if (mode == 1) std::sort(v.begin(), v.end(), pairAscend);
        else std::sort(v.begin(), v.end(), pairDescend);
        for (int i = 0; i < n; i++) perm[i] = v[i].second;
        return;
    }
    std::vector<LOverlap*> v(n);
    for (int i = 0; i < n; i++) {
        LOverlap* o = new LOverlap();
        o->trace_pts = NULL;
        o->read_A_match_start_ = 0; o->read_A_match_end_ = key[i];
        o->read_B_match_start_ = 0; o->read_B_match_end_ = 0;
        o->weight = key[i];
        o->tps = i;                       // carries the original index
        v[i] = o;
    }
    if (mode == 0) std::sort(v.begin(), v.end(), compare_overlap);
    else std::sort(v.begin(), v.end(), compare_overlap_weight);
    for (int i = 0; i < n; i++) { perm[i] = v[i]->tps; delete v[i]; }
}

// LAInterface::loadPAF over the reference's own lib/paf.c: n x 8 ints (a, b, ab, ae, bb, be, comp, 0)
long ref_load_paf(const char* paf_path, int* out, long cap) {
    LAInterface la;
    std::vector<LOverlap*> aln;
    long n = la.loadPAF(std::string(paf_path), aln);
    for (long i = 0; i < n && i < cap; i++) {
        LOverlap* o = aln[i];
        int* p = out + i * 8;
        p[0] = o->read_A_id_; p[1] = o->read_B_id_; p[2] = o->read_A_match_start_; p[3] = o->read_A_match_end_;
        p[4] = o->read_B_match_start_; p[5] = o->read_B_match_end_; p[6] = o->reverse_complement_match_; p[7] = 0;
    }
    for (auto o : aln) { o->trace_pts = NULL; delete o; }   // loadPAF leaves trace_pts uninitialised and ~LOverlap frees it
    return n;
}

// LAInterface::loadFASTA: read lengths in file order
int ref_fasta_lengths(const char* fasta_path, int* out, int cap) {
    LAInterface la;
    std::vector<Read*> reads;
    int n = la.loadFASTA(std::string(fasta_path), reads);
    for (int i = 0; i < n && i < cap; i++) out[i] = reads[i]->len;
    for (auto r : reads) delete r;
    return n;
}

long ref_ini_int(const char* file, const char* section, const char* name, long def) { INIReader r(file); return r.GetInteger(section, name, def); }
int ref_ini_bool(const char* file, const char* section, const char* name, int def) { INIReader r(file); return (int)r.GetBoolean(section, name, def != 0); }
double ref_ini_real(const char* file, const char* section, const char* name, double def) { INIReader r(file); return r.GetReal(section, name, def); }
int ref_ini_error(const char* file) { INIReader r(file); return r.ParseError(); }

}  // extern "C"
