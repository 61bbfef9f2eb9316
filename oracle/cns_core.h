// TEST INFRASTRUCTURE - NOT PRODUCT CODE.
// Shared by oracle/consensus_oracle.cpp and oracle/draft_oracle.cpp: the sequence DB reader, the LAlignment record, iter_np /
// recoverAlignment / getAlignmentTags restated (reference lines at each function; header comment of consensus_oracle.cpp).
#ifndef HINGE_ORACLE_CNS_CORE_H
#define HINGE_ORACLE_CNS_CORE_H
#include "oracle_io.h"

#include <climits>
#include <cstdarg>

namespace {

using oracle::Ini;

struct SeqDB {
    int ureads = 0, cutoff = 0, all = 1;
    std::vector<int> rlen;            // trimmed
    std::vector<int64_t> boff;        // trimmed
    std::vector<unsigned char> bps;
    int nfiles = 0;
    // numeric base (0..3) of trimmed read r at position p (Load_Subread: 4 bases per byte, first base in the top bits)
    int base(int r, int p) const {
        const unsigned char b = bps[(size_t)boff[r] + (size_t)(p >> 2)];
        return (b >> (6 - 2 * (p & 3))) & 3;
    }
};

static int open_seq_db(const std::string& name, SeqDB& db) {
    const std::string dir = oracle::path_dir(name), root = oracle::path_root(name, ".db");
    FILE* s = fopen((dir + "/" + root + ".db").c_str(), "r");
    if (!s) return -1;
    FILE* f = fopen((dir + "/." + root + ".idx").c_str(), "rb");
    if (!f) { fclose(s); return -1; }
    unsigned char hdr[112];
    if (fread(hdr, 112, 1, f) != 1) { fclose(f); fclose(s); return -1; }
    memcpy(&db.ureads, hdr + 0, 4);
    int nblocks = 0;
    if (fscanf(s, "files = %9d\n", &db.nfiles) != 1) { fclose(f); fclose(s); return -1; }
    for (int p = 0; p < db.nfiles; p++) {
        int last; char fname[10000], prolog[10000];
        if (fscanf(s, "  %9d %s %s\n", &last, fname, prolog) != 3) { fclose(f); fclose(s); return -1; }
    }
    if (fscanf(s, "blocks = %9d\n", &nblocks) == 1) {
        long long size;
        if (fscanf(s, "size = %9lld cutoff = %9d all = %1d\n", &size, &db.cutoff, &db.all) != 3) { fclose(f); fclose(s); return -1; }
    }
    fclose(s);
    std::vector<unsigned char> rec((size_t)db.ureads * 40);
    if (db.ureads && fread(rec.data(), 40, db.ureads, f) != (size_t)db.ureads) { fclose(f); return -1; }
    fclose(f);
    const bool trim = !(db.cutoff <= 0 && db.all);
    const int allflag = db.all ? 0 : 0x800;
    for (int i = 0; i < db.ureads; i++) {
        int rlen, flags; int64_t boff;
        memcpy(&rlen, &rec[(size_t)i * 40 + 4], 4);
        memcpy(&boff, &rec[(size_t)i * 40 + 16], 8);
        memcpy(&flags, &rec[(size_t)i * 40 + 32], 4);
        if (trim && !((flags & 0x800) >= allflag && rlen >= db.cutoff)) continue;
        db.rlen.push_back(rlen);
        db.boff.push_back(boff);
    }
    FILE* b = fopen((dir + "/." + root + ".bps").c_str(), "rb");
    if (!b) return -1;
    fseek(b, 0, SEEK_END);
    const long sz = ftell(b);
    fseek(b, 0, SEEK_SET);
    db.bps.resize((size_t)std::max(sz, 0l));
    if (sz > 0 && fread(db.bps.data(), (size_t)sz, 1, b) != 1) { fclose(b); return -1; }
    fclose(b);
    return 0;
}

struct Aln {   // LAlignment, the fields the path touches (LAInterface.h:44-74): raw .las coordinates (B in the complement frame when flags == 1)
    int a = 0, b = 0, alen = 0, blen = 0, comp = 0;
    int ab = 0, ae = 0, bb = 0, be = 0;
    std::vector<uint16_t> pts;       // trace points, 16-bit (Decompress_TraceTo16)
    std::vector<int> trace;          // the recovered indel list (recoverAlignment)
};

// ---- iter_np (LAInterface.cpp:3152-3404) ------------------------------------------------------------------------------
// Wave D holds, per diagonal k = (A index) - (B index), the furthest B index F[D][k] reachable with "D" units, and H[D][k],
// the move that got there (0: from wave D-1 on k; -1 / 2: from wave D-2 on k -+ 1; 4 / 1: from THIS wave's neighbour k +- 1).
struct Waves {
    int kmin = 0, width = 0;
    std::vector<std::vector<int>> V, H;   // row D at index D + 2
    void reset(int M, int N) {
        kmin = -(N + 4); width = M + N + 9;
        V.clear(); H.clear();
    }
    void need(int D) {
        while ((int)V.size() < D + 3) { V.emplace_back(width, INT_MIN / 2); H.emplace_back(width, 0); }
    }
    int& v(int D, int k) { return V[(size_t)(D + 2)][(size_t)(k - kmin)]; }
    int& h(int D, int k) { return H[(size_t)(D + 2)][(size_t)(k - kmin)]; }
};

static long g_neg_slides = 0;   // probe: slides that start at a negative B index (they would read in front of the segment)

// A, B: numeric bases of the segment (A[0..M), B[0..N)); a_abs / b_abs: 0-based absolute start of the segment in aseq / bseq
static void iter_np(const signed char* A, int M, const signed char* B, int N, Waves& w, std::vector<int>& stop, int a_abs, int b_abs) {
    const int del = M - N;
    int low = del >= 0 ? 0 : del, hgh = del >= 0 ? del : 0;
    w.reset(M, N);
    w.need(-1);
    for (int k = low - 1; k <= hgh + 1; k++) w.v(-2, k) = w.v(-1, k) = -2;
    w.v(-1, 0) = -1;
    low += 1; hgh -= 1;
    int D;
    for (D = 0;; D++) {
        w.need(D);
        if ((D & 1) == 0) { low -= 1; hgh += 1; }   // (posl / posh only bind when A and B are the same sequence)
        w.v(D, hgh + 1) = w.v(D, low - 1) = -2;
        auto move = [&](int k, int am, int ap, int mdir, int pdir) {
            const int ac = w.v(D - 1, k) + 1;
            int j;
            if (ac < am) { if (ap < am) { w.h(D, k) = mdir; j = am; } else { w.h(D, k) = pdir; j = ap; } }
            else { if (ap < ac) { w.h(D, k) = 0; j = ac; } else { w.h(D, k) = pdir; j = ap; } }
            const int i = M - k;               // A[j + k] exists while j < M - k
            const int lim = N < i ? N : i;
            if (j < 0 && j < lim) g_neg_slides++;
            while (j < lim && j >= 0 && B[j] == A[j + k]) j++;
            w.v(D, k) = j;
            return j;
        };
        int j = -2;
        for (int k = hgh; k > del; k--) j = move(k, w.v(D - 2, k - 1), j + 1, -1, 4);
        j = -2;
        for (int k = low; k < del; k++) j = move(k, j, w.v(D - 2, k + 1) + 1, 2, 1);
        move(del, j, w.v(D, del + 1) + 1, 2, 4);
        if (w.v(D, del) >= N) break;
    }
    // trace-back: reverse the move chain from (D, del) to (0, 0), re-sliding each horizontal / vertical step as far back along
    // its snake as the neighbouring waves allow (LAInterface.cpp:3285-3352)
    {
        w.h(0, 0) = 3;
        int c = N, k = del;
        int e = w.h(D, k);
        w.h(D, k) = 3;
        while (e != 3) {
            int h = k + e;
            if (e > 1) h -= 3;
            else if (e == 0) D -= 1;
            else D -= 2;
            if (h < k) {   // e = -1 or 2
                int m = k < 0 ? -k : 0;
                if (w.v(D, h) <= c) c = w.v(D, h) - 1;
                while (c >= m && A[c + k] == B[c]) c -= 1;
                if (e < 1) {   // the edge is 2, the others are 1 and 0
                    if (c <= w.v(D + 2, k + 1)) { e = 4; h = k + 1; D = D + 2; }
                    else if (c == w.v(D + 1, k)) { e = 0; h = k; D = D + 1; }
                    else w.v(D, h) = c + 1;
                } else {       // the edge is 0, the others are 1 and 2 (k != del) or 0
                    m = (k == del) ? D : D - 2;
                    if (c <= w.v(m, k + 1)) { e = (k == del) ? 4 : 1; h = k + 1; D = m; }
                    else if (c == w.v(D - 1, k)) { e = 0; h = k; D = D - 1; }
                    else w.v(D, h) = c + 1;
                }
            }
            const int m2 = w.h(D, h);
            w.h(D, h) = e;
            e = m2;
            k = h;
        }
    }
    // forward again along the reversed chain: one entry per indel (LAInterface.cpp:3354-3371); positions are 1-based and
    // absolute: +B position for a gap in B, -(A position) for a gap in A
    {
        const int ap = -a_abs - 1, bp = b_abs + 1;
        int k = 0, DD = 0;
        int e = w.h(DD, k);
        while (e != 3) {
            int h = k - e;
            const int c = w.v(DD, k);
            if (e > 1) h += 3;
            else if (e == 0) DD += 1;
            else DD += 2;
            if (h > k) stop.push_back(bp + c);
            else if (h < k) stop.push_back(ap - (c + k));
            k = h;
            e = w.h(DD, h);
        }
    }
}

struct Out {
    FILE* log = nullptr;
    void pf(const char* fmt, ...) {
        if (!log) return;
        va_list ap; va_start(ap, fmt); vfprintf(log, fmt, ap); va_end(ap);
    }
};

// recoverAlignment (LAInterface.cpp:4125-4244) + computeTracePTS (:3410-3506)
static void recover(const SeqDB& d1, const SeqDB& d2, Aln& al, int tspace, Waves& w) {
    std::vector<signed char> aseg, bseg;
    auto a_at = [&](int x) { return (signed char)d1.base(al.a, x); };
    auto b_at = [&](int x) { return (signed char)(al.comp ? 3 - d2.base(al.b, al.blen - 1 - x) : d2.base(al.b, x)); };
    al.trace.clear();
    auto segment = [&](int ab, int ae, int bb, int be) {
        aseg.resize((size_t)std::max(ae - ab, 0)); bseg.resize((size_t)std::max(be - bb, 0));
        for (int x = ab; x < ae; x++) aseg[(size_t)(x - ab)] = a_at(x);
        for (int x = bb; x < be; x++) bseg[(size_t)(x - bb)] = b_at(x);
        iter_np(aseg.data(), ae - ab, bseg.data(), be - bb, w, al.trace, ab, bb);
    };
    int ab = al.ab, ae = (ab / tspace) * tspace, bb = al.bb;
    const int tl = (int)al.pts.size() - 2;
    for (int i = 1; i < tl; i += 2) {
        ae += tspace;
        const int be = bb + al.pts[(size_t)i];
        segment(ab, ae, bb, be);
        ab = ae; bb = be;
    }
    segment(ab, al.ae, bb, al.be);
}

// getAlignmentTags (LAInterface.cpp:3709-3905): the two gapped rows as 0..3 / 4 = '-'
static void tags(const SeqDB& d1, const SeqDB& d2, const Aln& al, std::string& ra, std::string& rb) {
    static const char U[5] = {'A', 'C', 'G', 'T', '-'};
    auto a_at = [&](int i1) { return d1.base(al.a, i1 - 1); };                                      // 1-based
    auto b_at = [&](int j1) { return al.comp ? 3 - d2.base(al.b, al.blen - 1 - (j1 - 1)) : d2.base(al.b, j1 - 1); };
    ra.clear(); rb.clear();
    int i = al.ab + 1, j = al.bb + 1;   // (what the prefix loops of :3783-3813 leave)
    for (int p : al.trace) {
        if (p < 0) {
            p = -p;
            while (i != p) { ra.push_back(U[a_at(i)]); rb.push_back(U[b_at(j)]); i++; j++; }
            ra.push_back('-'); rb.push_back(U[b_at(j)]); j++;
        } else {
            while (j != p) { ra.push_back(U[a_at(i)]); rb.push_back(U[b_at(j)]); i++; j++; }
            ra.push_back(U[a_at(i)]); rb.push_back('-'); i++;
        }
    }
    while (i <= al.ae) { ra.push_back(U[a_at(i)]); rb.push_back(U[b_at(j)]); i++; j++; }
}

// chop_end (consensus.cpp:27-45)
static int chop_end(std::string& ra, std::string& rb, int chop) {
    const int len = (int)ra.size();
    if (len < chop * 2 + 10) return 0;
    int start = chop;
    while (ra[(size_t)start] == '-') start++;
    int offset = 0;
    for (int i = 0; i < start; i++) if (ra[(size_t)i] != '-') offset++;
    ra = ra.substr((size_t)start, (size_t)(len - start - chop));
    rb = rb.substr((size_t)start, (size_t)(len - start - chop));
    return offset;
}

static int load_alignments(const std::string& path, const SeqDB& d1, const SeqDB& d2, std::vector<Aln>& out, int& tspace, int64_t& novl) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return -1;
    if (fread(&novl, 8, 1, f) != 1 || fread(&tspace, 4, 1, f) != 1) { fclose(f); return -1; }
    const int tbytes = tspace <= 125 ? 1 : 2;
    std::vector<unsigned char> tbuf;
    for (int64_t j = 0; j < novl; j++) {
        int32_t r[10];
        if (fread(r, 40, 1, f) != 1) break;
        const int tlen = r[0];
        tbuf.resize((size_t)tlen * tbytes);
        if (tlen > 0 && fread(tbuf.data(), (size_t)tlen * tbytes, 1, f) != 1) break;
        const int ar = r[7] + 1;
        // getAlignment(res, 0, n_alns): A reads 1 .. n_alns (1-based) pass the range filter (LAInterface.cpp:1800-1890)
        if (!(ar >= 1 && ar <= novl)) continue;
        if (r[7] >= (int)d1.rlen.size() || r[8] < 0 || r[8] >= (int)d2.rlen.size()) { fclose(f); return -3; }
        Aln al;
        al.a = r[7]; al.b = r[8]; al.alen = d1.rlen[(size_t)al.a]; al.blen = d2.rlen[(size_t)al.b];
        al.comp = (r[6] & 1) ? 1 : 0;
        al.ab = r[2]; al.bb = r[3]; al.ae = r[4]; al.be = r[5];
        al.pts.resize((size_t)tlen);
        for (int k = 0; k < tlen; k++) al.pts[(size_t)k] = tbytes == 1 ? tbuf[(size_t)k] : (uint16_t)(tbuf[2 * (size_t)k] | (tbuf[2 * (size_t)k + 1] << 8));
        out.push_back(std::move(al));
    }
    fclose(f);
    return 0;
}

static bool by_aligned_length(const Aln* x, const Aln* y) {   // compare_overlap_aln (LAInterface.cpp:4925-4927)
    return (x->ae - x->ab + x->be - x->bb) > (y->ae - y->ab + y->be - y->bb);
}

}  // namespace
#endif
