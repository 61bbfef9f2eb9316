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
static inline int open_db(const std::string& name, DB& db) {
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
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return -1;
    int ok = fread(&h.novl, 8, 1, f) == 1 && fread(&h.tspace, 4, 1, f) == 1;
    fclose(f);
    if (!ok) return -1;
    h.tbytes = h.tspace <= 125 ? 1 : 2;
    return 0;
}

// getOverlap(vec, 0, n_read) (LAInterface.cpp:1519-1634)
static inline int load_overlaps(const std::string& path, const DB& db, std::vector<Ovl*>& out, LasHeader& h) {
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
    if (mlas) return base;
    if (base.size() >= 4 && base.substr(base.size() - 4) == ".las") return base;
    return base + ".las";
}

}  // namespace oracle
