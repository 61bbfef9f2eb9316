// TEST INFRASTRUCTURE - NOT PRODUCT CODE.
// CPU restatement ("oracle") of the HINGE filter / maximal / layout path: input side.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under
// oracle/.  The product (hinge_amd/) never includes, links or calls this.
//
// Restates, with plain fread()s, what the reference reads through LAInterface + DAZZ_DB:
//   * DB stub + index + trimming      /root/reference/src/lib/DB.c:395-578, 585-683
//   * qual track                      /root/reference/src/lib/DB.c:1080-1300,
//                                     /root/reference/src/lib/LAInterface.cpp:4369-4494
//   * .las header / records / trace   /root/reference/src/lib/LAInterface.cpp:595-621,1519-1634,
//                                     /root/reference/src/lib/align.c:3042-3081
//   * inih + INIReader                /root/reference/src/lib/ini.c:66-165,
//                                     /root/reference/src/lib/INIReader.cpp:23-80
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cctype>
#include <zlib.h>
#include <map>
#include <string>
#include <vector>
#include <algorithm>

namespace oracle {

enum MatchType {  // same numbering as /root/reference/src/include/LAInterface.h:30-33
    FORWARD, BACKWARD, ACOVERB, BCOVERA, UNDEFINED, INTERNAL, NOT_ACTIVE, COVERING,
    COVERED, MIDDLE, MISMATCH_LEFT, MISMATCH_RIGHT, FORWARD_INTERNAL, BACKWARD_INTERNAL
};

struct Read {
    int id = 0;
    int len = 0;
    int effective_start = 0, effective_end = 0;   // reference leaves these uninitialised (LAInterface.h:21)
    bool active = true;
};

struct Ovl {  // the fields of LOverlap the path touches (LAInterface.h:76-110)
    int a = 0, b = 0;
    int alen = 0, blen = 0;
    int tlen = 0;
    int ab = 0, ae = 0, bb = 0, be = 0;            // B already flipped to the forward strand
    int comp = 0;
    int eff_ab = 0, eff_ae = 0, eff_bb = 0, eff_be = 0;
    int eff_a_rs = 0, eff_a_re = 0, eff_b_rs = 0, eff_b_re = 0;
    int eff_start_idx = 0, eff_end_idx = 0;
    MatchType type = UNDEFINED;
    bool active = true;
    int weight = 0, length = 0;
    std::vector<uint16_t> trace;                   // decompressed to 16 bit like Decompress_TraceTo16
};

struct DB {
    bool fasta = false;                             // reads came from --fasta (LAInterface::loadFASTA), no DAZZ_DB
    int ureads = 0, treads = 0, cutoff = 0, all = 1;
    std::vector<int> rlen;                          // trimmed
    std::vector<char> keep;                         // per untrimmed read
    std::string dir, root;
};

static inline std::string path_dir(const std::string& p) {
    size_t s = p.rfind('/');
    return s == std::string::npos ? std::string(".") : p.substr(0, s);
}
static inline std::string path_root(const std::string& p, const char* suffix) {
    size_t s = p.rfind('/');
    std::string b = s == std::string::npos ? p : p.substr(s + 1);
    size_t n = strlen(suffix);
    if (b.size() >= n && b.compare(b.size() - n, n, suffix) == 0) b = b.substr(0, b.size() - n);
    return b;
}

// Open_DB + Trim_DB (DB.c:395-683): returns 0 ok, -1 on failure (reference exits 1).
// ---- FASTA / PAF input (filter.cpp:289-291,499-503; LAInterface.cpp:4808-4870; lib/paf.c:59-92) -----------
// A name prefixed "fasta:" / "paf:" selects this input in open_db() / load_overlaps(), so the three stage
// restatements below are shared between the two input modes exactly like the reference's mains are.
static inline bool has_prefix(const std::string& s, const char* p) { return s.compare(0, strlen(p), p) == 0; }

// whole file through zlib (gzopen reads plain files too), like kseq / kstream over gzread
static inline int slurp_gz(const std::string& path, std::string& out) {
    gzFile f = gzopen(path.c_str(), "r");
    if (!f) return -1;
    char buf[1 << 16];
    int n;
    while ((n = gzread(f, buf, sizeof(buf))) > 0) out.append(buf, (size_t)n);
    gzclose(f);
    return n < 0 ? -1 : 0;
}

// loadFASTA (LAInterface.cpp:4849-4870) over this tree's kseq_read (include/kseq.h:193-232): a record starts at the
// next '>' or '@'; name = up to the first whitespace, rest of the line = comment; the sequence is every following
// non-empty line, WHOLE (blanks included), until a line that starts with '>', '+' or '@'; for '+' the rest of that
// line is skipped and whole quality lines are read until they hold at least as many characters, which must then be
// exactly as many (otherwise kseq_read fails and loadFASTA stops).  Only the length reaches the path; read k is the
// k-th record.
static inline int load_fasta_lengths(const std::string& path, std::vector<int>& rlen) {
    typedef int RLEN_T;
    std::string t;
    if (slurp_gz(path, t) != 0) return -1;
    size_t i = 0;
    const size_t n = t.size();
    int last_char = 0;
    for (;;) {
        if (last_char == 0) {                                      // jump to the next header character, wherever it is
            while (i < n && t[i] != '>' && t[i] != '@') i++;
            if (i >= n) break;
            last_char = t[i++];
        }
        if (i >= n) break;                                         // ks_getuntil(name) at end of stream: -1
        while (i < n && !isspace((unsigned char)t[i])) i++;        // name
        const int dc = i < n ? t[i] : 0;
        if (i < n) i++;
        if (dc != '\n') { while (i < n && t[i] != '\n') i++; if (i < n) i++; }   // comment
        size_t len = 0;
        int c = -1;
        bool last_cr = false;                                      // is the last accumulated sequence character a '\r'?
        while (i < n) {
            c = (unsigned char)t[i++];
            if (c == '>' || c == '+' || c == '@') break;
            if (c == '\n') { c = -1; continue; }
            len++; last_cr = (c == '\r');
            while (i < n && t[i] != '\n') { len++; last_cr = (t[i] == '\r'); i++; }
            if (i < n) i++;
            if (len > 1 && last_cr) { len--; last_cr = false; }    // KS_SEP_LINE drops one trailing CR of the accumulated string
            c = -1;
        }
        if (c == '>' || c == '@') last_char = c;
        if (c != '+') { rlen.push_back((RLEN_T)len); if (c == -1) { if (i >= n) break; } continue; }
        while (i < n && t[i] != '\n') i++;                         // rest of the '+' line
        if (i >= n) break;                                         // no quality string: kseq_read returns -2, the record is dropped
        i++;
        size_t ql = 0;
        bool q_cr = false;
        while (i < n && ql < len) {                                // whole lines until at least len quality characters
            while (i < n && t[i] != '\n') { ql++; q_cr = (t[i] == '\r'); i++; }
            if (i < n) i++;
            if (ql > 1 && q_cr) { ql--; q_cr = false; }
        }
        last_char = 0;
        if (ql != len) break;                                      // -2: loadFASTA's loop ends, the record is dropped
        rlen.push_back((RLEN_T)len);
    }
    return 0;
}

// get_id_from_string (LAInterface.cpp:4808-4819): the number between the first and the second '/'.
// The reference dereferences NULL when a name has fewer than two '/': reported as -1 here (undefined).
static inline int id_from_name(const char* name, bool& ok) {
    const char* s0 = strchr(name, '/');
    if (!s0) { ok = false; return 0; }
    const char* s1 = s0 + 1;
    const char* s2 = strchr(s1, '/');
    if (!s2) { ok = false; return 0; }
    char sub[32];
    size_t l = (size_t)(s2 - s1);
    if (l >= 15) { ok = false; return 0; }                       // the reference's char substr[15] would overflow
    memcpy(sub, s1, l); sub[l] = 0;
    return atoi(sub);
}

static inline int open_db(const std::string& name, DB& db) {
    if (has_prefix(name, "fasta:")) {
        db.fasta = true;
        if (load_fasta_lengths(name.substr(6), db.rlen) != 0) return -1;
        db.treads = db.ureads = (int)db.rlen.size();
        return 0;
    }
    db.dir = path_dir(name);
    db.root = path_root(name, ".db");
    std::string stub = db.dir + "/" + db.root + ".db";
    std::string idx = db.dir + "/." + db.root + ".idx";
    FILE* s = fopen(stub.c_str(), "r");
    if (!s) return -1;
    FILE* f = fopen(idx.c_str(), "rb");
    if (!f) { fclose(s); return -1; }
    unsigned char hdr[112];
    if (fread(hdr, 112, 1, f) != 1) { fclose(f); fclose(s); return -1; }
    memcpy(&db.ureads, hdr + 0, 4);
    memcpy(&db.treads, hdr + 4, 4);
    int nfiles = 0, nblocks = 0;
    db.cutoff = 0; db.all = 1;
    if (fscanf(s, "files = %9d\n", &nfiles) != 1) { fclose(f); fclose(s); return -1; }
    for (int p = 0; p < nfiles; p++) {
        int last; char fname[10000], prolog[10000];
        if (fscanf(s, "  %9d %s %s\n", &last, fname, prolog) != 3) { fclose(f); fclose(s); return -1; }
    }
    if (fscanf(s, "blocks = %9d\n", &nblocks) == 1) {
        long long size;
        if (fscanf(s, "size = %9lld cutoff = %9d all = %1d\n", &size, &db.cutoff, &db.all) != 3) {
            fclose(f); fclose(s); return -1;
        }
    }
    fclose(s);
    std::vector<unsigned char> rec((size_t)db.ureads * 40);
    if (db.ureads && fread(rec.data(), 40, db.ureads, f) != (size_t)db.ureads) { fclose(f); return -1; }
    fclose(f);
    db.keep.assign(db.ureads, 1);
    bool trim = !(db.cutoff <= 0 && db.all);
    int allflag = db.all ? 0 : 0x800;
    for (int i = 0; i < db.ureads; i++) {
        int rlen, flags;
        memcpy(&rlen, &rec[(size_t)i * 40 + 4], 4);
        memcpy(&flags, &rec[(size_t)i * 40 + 32], 4);
        if (trim && !((flags & 0x800) >= allflag && rlen >= db.cutoff)) { db.keep[i] = 0; continue; }
        db.rlen.push_back(rlen);
    }
    return 0;
}

// getQV (LAInterface.cpp:4369-4494): 0 ok, 1 = no (usable) qual track.
static inline int load_qv(const DB& db, std::vector<std::vector<int>>& QV) {
    if (db.fasta) return -1;   // has_qv = false, filter.cpp:291
    std::string pre = db.dir + "/." + db.root + ".qual";
    FILE* a = fopen((pre + ".anno").c_str(), "rb");
    if (!a) return 1;
    int tracklen = 0, size = 0;
    if (fread(&tracklen, 4, 1, a) != 1 || fread(&size, 4, 1, a) != 1) { fclose(a); return 1; }
    int n = (int)db.rlen.size();
    if (tracklen != db.ureads && tracklen != db.treads) { fclose(a); return 1; }
    std::vector<int64_t> off((size_t)tracklen + 1);
    if (fread(off.data(), 8, (size_t)tracklen + 1, a) != (size_t)tracklen + 1) { fclose(a); return 1; }
    fclose(a);
    FILE* d = fopen((pre + ".data").c_str(), "rb");
    if (!d) return 1;
    std::vector<unsigned char> data((size_t)off[tracklen]);
    if (!data.empty() && fread(data.data(), 1, data.size(), d) != data.size()) { fclose(d); return 1; }
    fclose(d);
    QV.clear();
    bool untrimmed_track = (tracklen == db.ureads) && (db.ureads != n);
    for (int i = 0, j = 0; i < tracklen; i++) {
        if (untrimmed_track && !db.keep[i]) continue;
        std::vector<int> q;
        for (int64_t k = off[i]; k < off[i + 1]; k++) q.push_back(data[(size_t)k]);
        QV.push_back(q);
        j++;
    }
    return 0;
}

struct LasHeader { int64_t novl = 0; int tspace = 0; int tbytes = 1; };

static inline int las_header(const std::string& path, LasHeader& h) {
    if (has_prefix(path, "paf:")) { h.novl = 0; h.tspace = 100; h.tbytes = 1; return 0; }
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return -1;
    int ok = fread(&h.novl, 8, 1, f) == 1 && fread(&h.tspace, 4, 1, f) == 1;
    fclose(f);
    if (!ok) return -1;
    h.tbytes = h.tspace <= 125 ? 1 : 2;
    return 0;
}

// getOverlap(vec, 0, n_read) (LAInterface.cpp:1519-1634)
// loadPAF (LAInterface.cpp:4822-4845) over paf_read / paf_parse (lib/paf.c:59-92): tab-separated lines, lines with
// fewer than 10 fields are skipped, numbers by strtol base 10 into uint32, strand = (first char == '-').
// B coordinates are NOT flipped for reverse matches here (minimap's target coordinates are forward-strand already),
// there are no trace points (tlen 0).  Returns -3 when a read name has no "/id/" part (the reference crashes).
static inline int load_paf(const std::string& path, std::vector<Ovl*>& out, int64_t& n_rec) {
    std::string t;
    if (slurp_gz(path, t) != 0) return -1;
    n_rec = 0;
    size_t i = 0, n = t.size();
    while (i < n) {
        size_t e = t.find('\n', i);
        if (e == std::string::npos) e = n;
        std::string line = t.substr(i, e - i);
        i = e + 1;
        if (!line.empty() && line[line.size() - 1] == '\r') { /* ks_getuntil(KS_SEP_LINE) strips a trailing CR */ line.erase(line.size() - 1); }
        std::vector<std::string> f;
        size_t p0 = 0;
        for (size_t k = 0; k <= line.size(); k++)
            if (k == line.size() || line[k] == '\t') { f.push_back(line.substr(p0, k - p0)); p0 = k + 1; }
        if (f.size() < 10) continue;
        bool ok = true;
        Ovl* o = new Ovl();
        o->ab = (int)(uint32_t)strtol(f[2].c_str(), NULL, 10);
        o->ae = (int)(uint32_t)strtol(f[3].c_str(), NULL, 10);
        o->bb = (int)(uint32_t)strtol(f[7].c_str(), NULL, 10);
        o->be = (int)(uint32_t)strtol(f[8].c_str(), NULL, 10);
        o->alen = (int)(uint32_t)strtol(f[1].c_str(), NULL, 10);
        o->blen = (int)(uint32_t)strtol(f[6].c_str(), NULL, 10);
        o->comp = (!f[4].empty() && f[4][0] == '-') ? 1 : 0;
        o->a = id_from_name(f[0].c_str(), ok) - 1;
        o->b = id_from_name(f[5].c_str(), ok) - 1;
        o->tlen = 0;
        if (!ok) { delete o; return -3; }
        out.push_back(o);
        n_rec++;
    }
    return 0;
}

static inline int load_overlaps(const std::string& path, const DB& db, std::vector<Ovl*>& out, LasHeader& h) {
    if (has_prefix(path, "paf:")) {
        h.tspace = 100; h.tbytes = 1;
        int rc = load_paf(path.substr(4), out, h.novl);
        if (rc != 0) return rc;
        int n_read = (int)db.rlen.size();
        for (size_t k = 0; k < out.size(); k++)   // the reference indexes reads[] / idx_pileup[] with these ids unchecked
            if (out[k]->a < 0 || out[k]->a >= n_read || out[k]->b < 0 || out[k]->b >= n_read) return -3;
        return 0;
    }
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return -1;
    if (fread(&h.novl, 8, 1, f) != 1 || fread(&h.tspace, 4, 1, f) != 1) { fclose(f); return -1; }
    h.tbytes = h.tspace <= 125 ? 1 : 2;
    int n_read = (int)db.rlen.size();
    std::vector<unsigned char> tbuf;
    for (int64_t j = 0; j < h.novl; j++) {
        int32_t r[10];
        if (fread(r, 40, 1, f) != 1) break;
        int tlen = r[0];
        tbuf.resize((size_t)tlen * h.tbytes);
        if (tlen > 0 && fread(tbuf.data(), (size_t)tlen * h.tbytes, 1, f) != 1) break;
        int aread = r[7], bread = r[8];
        int ar = aread + 1;
        if (!(ar >= 1 && ar <= n_read)) continue;
        Ovl* o = new Ovl();
        o->a = aread; o->b = bread;
        o->alen = db.rlen[aread]; o->blen = db.rlen[bread];
        o->comp = (r[6] & 1) ? 1 : 0;
        o->ab = r[2]; o->ae = r[4];
        if (!o->comp) { o->bb = r[3]; o->be = r[5]; }
        else { o->bb = o->blen - r[5]; o->be = o->blen - r[3]; }
        o->tlen = tlen;
        o->trace.resize(tlen);
        for (int k = 0; k < tlen; k++)
            o->trace[k] = h.tbytes == 1 ? tbuf[k] : (uint16_t)(tbuf[2 * k] | (tbuf[2 * k + 1] << 8));
        out.push_back(o);
    }
    fclose(f);
    return 0;
}

// ---- inih + INIReader -----------------------------------------------------------------------
struct Ini {
    std::map<std::string, std::string> values;
    int error = 0;

    static char* rstrip(char* s) {
        char* p = s + strlen(s);
        while (p > s && isspace((unsigned char)(*--p))) *p = '\0';
        return s;
    }
    static char* lskip(char* s) {
        while (*s && isspace((unsigned char)(*s))) s++;
        return s;
    }
    static char* find_char_or_comment(char* s, char c) {
        int was_ws = 0;
        while (*s && *s != c && !(was_ws && *s == ';')) { was_ws = isspace((unsigned char)(*s)); s++; }
        return s;
    }
    static std::string key(const std::string& section, const std::string& name) {
        std::string k = section + "=" + name;
        std::transform(k.begin(), k.end(), k.begin(), ::tolower);
        return k;
    }
    void handle(const char* section, const char* name, const char* value) {
        std::string k = key(section, name);
        if (values[k].size() > 0) values[k] += "\n";
        values[k] += value;
    }
    explicit Ini(const std::string& filename) {
        FILE* file = fopen(filename.c_str(), "r");
        if (!file) { error = -1; return; }
        char line[200];
        char section[50] = "", prev_name[50] = "";
        int lineno = 0;
        while (fgets(line, 200, file) != NULL) {
            lineno++;
            char* start = line;
            if (lineno == 1 && (unsigned char)start[0] == 0xEF && (unsigned char)start[1] == 0xBB &&
                (unsigned char)start[2] == 0xBF)
                start += 3;
            start = lskip(rstrip(start));
            if (*start == ';' || *start == '#') {
            } else if (*prev_name && *start && start > line) {
                handle(section, prev_name, start);
            } else if (*start == '[') {
                char* end = find_char_or_comment(start + 1, ']');
                if (*end == ']') {
                    *end = '\0';
                    strncpy(section, start + 1, sizeof(section)); section[sizeof(section) - 1] = '\0';
                    *prev_name = '\0';
                } else if (!error) error = lineno;
            } else if (*start && *start != ';') {
                char* end = find_char_or_comment(start, '=');
                if (*end != '=') end = find_char_or_comment(start, ':');
                if (*end == '=' || *end == ':') {
                    *end = '\0';
                    char* name = rstrip(start);
                    char* value = lskip(end + 1);
                    end = find_char_or_comment(value, '\0');
                    if (*end == ';') *end = '\0';
                    rstrip(value);
                    strncpy(prev_name, name, sizeof(prev_name)); prev_name[sizeof(prev_name) - 1] = '\0';
                    handle(section, name, value);
                } else if (!error) error = lineno;
            }
        }
        fclose(file);
    }
    std::string get(const std::string& s, const std::string& n, const std::string& def) const {
        auto it = values.find(key(s, n));
        return it != values.end() ? it->second : def;
    }
    long get_int(const std::string& s, const std::string& n, long def) const {
        std::string v = get(s, n, "");
        const char* c = v.c_str();
        char* end;
        long r = strtol(c, &end, 0);
        return end > c ? r : def;
    }
    double get_real(const std::string& s, const std::string& n, double def) const {
        std::string v = get(s, n, "");
        const char* c = v.c_str();
        char* end;
        double r = strtod(c, &end);
        return end > c ? r : def;
    }
    bool get_bool(const std::string& s, const std::string& n, bool def) const {
        std::string v = get(s, n, "");
        std::transform(v.begin(), v.end(), v.begin(), ::tolower);
        if (v == "true" || v == "yes" || v == "on" || v == "1") return true;
        if (v == "false" || v == "no" || v == "off" || v == "0") return false;
        return def;
    }
};

// name.1.las, name.2.las, ... (filter.cpp:35-63)
static inline std::vector<std::string> las_parts(const std::string& base) {
    std::vector<std::string> r;
    for (int i = 1;; i++) {
        std::string p = base + "." + std::to_string(i) + ".las";
        FILE* f = fopen(p.c_str(), "rb");
        if (!f) break;
        fclose(f);
        r.push_back(p);
    }
    return r;
}

static inline std::string las_name(const std::string& base, bool mlas) {
    if (has_prefix(base, "paf:")) return base;
    if (mlas) return base;
    if (base.size() >= 4 && base.substr(base.size() - 4) == ".las") return base;
    return base + ".las";
}

}  // namespace oracle
