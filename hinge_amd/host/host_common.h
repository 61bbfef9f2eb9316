// Host side of the three stage programs (Reads_filter, get_maximal_reads, hinging): argv and nominal.ini
// surface, DAZZ_DB / .las ingest into the SoA layout of include/hinge_hip.h, text writers.
// All arithmetic on overlaps happens behind the C ABI (libhinge_hip.so); nothing here computes coverage,
// masks, hinges or overlap types.
//
// Reference behaviour mirrored (file:line under /root/reference/src):
//   cmdline flags            filter/filter.cpp:172-183, maximal/maximal.cpp:242-253, layout/hinging.cpp:621-640
//   INI reader + quirks      lib/ini.c:43-165, lib/INIReader.cpp:23-80
//   DB stub/index/trim       lib/DB.c:395-683;  qual track lib/DB.c:1080-1300, lib/LAInterface.cpp:4369-4494
//   .las header/records      lib/LAInterface.cpp:595-621,1519-1634, lib/align.c:3042-3081
#pragma once
#include <algorithm>
#include <cctype>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <map>
#include <memory>
#include <chrono>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <atomic>
#include <thread>
#include <unistd.h>
#include <vector>
#include <zlib.h>

#include "../../include/hinge_hip.h"

// Error exits while the HIP runtime may still be initialising on the helper thread (CtxInit): exit() would run the static
// destructors under that thread's feet (an occasional SIGSEGV instead of the exit code); _exit() after a flush does not.
[[noreturn]] inline void quit(int code) {
    fflush(nullptr);
    _exit(code);
}


namespace hh {

// ---------------------------------------------------------------------------------------------------
// logging: stdout + <log dir>/log.txt (the reference logs through spdlog; the text is not part of parity)
// ---------------------------------------------------------------------------------------------------
struct Log {
    FILE* file = nullptr;
    void open(const std::string& dir) {
        mkdir(dir.c_str(), S_IRWXU | S_IRWXG | S_IROTH | S_IXOTH);
        file = fopen((dir + "/log.txt").c_str(), "a");
    }
    void line(const char* level, const char* fmt, va_list ap) {
        char buf[4096];
        vsnprintf(buf, sizeof(buf), fmt, ap);
        printf("[log] [%s] %s\n", level, buf);
        if (file) { fprintf(file, "[log] [%s] %s\n", level, buf); fflush(file); }
    }
    void info(const char* fmt, ...) { va_list ap; va_start(ap, fmt); line("info", fmt, ap); va_end(ap); }
    void warn(const char* fmt, ...) { va_list ap; va_start(ap, fmt); line("warning", fmt, ap); va_end(ap); }
    void error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); line("error", fmt, ap); va_end(ap); }
};

// wall-clock marks of the host phases on stderr when HINGE_HOST_TIMING is set (tools/e2e_bench.py reads them)
struct PhaseTimer {
    bool on;
    const char* prog;
    std::chrono::steady_clock::time_point t0, last;
    explicit PhaseTimer(const char* p) : on(getenv("HINGE_HOST_TIMING") != nullptr), prog(p), t0(std::chrono::steady_clock::now()), last(t0) {}
    void mark(const char* what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[timing] %s %-28s %8.1f ms\n", prog, what, std::chrono::duration<double, std::milli>(now - last).count());
        last = now;
    }
    ~PhaseTimer() {
        if (on) fprintf(stderr, "[timing] %s %-28s %8.1f ms\n", prog, "TOTAL", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
};

// ---------------------------------------------------------------------------------------------------
// argv: the subset of tanakh cmdline.h the three programs use
// ---------------------------------------------------------------------------------------------------
class CmdLine {
public:
    void add_string(const std::string& name, char shortn, const std::string& desc, bool need, const std::string& def) {
        opts_.push_back({name, shortn, desc, need, true, def, false});
    }
    void add_flag(const std::string& name, char shortn, const std::string& desc) { opts_.push_back({name, shortn, desc, false, false, "", false}); }
    std::string get(const std::string& name) const { for (auto& o : opts_) if (o.name == name) return o.value; return ""; }
    bool exist(const std::string& name) const { for (auto& o : opts_) if (o.name == name) return o.set; return false; }

    // parse_check: usage + exit(1) on any error, usage + exit(0) on --help
    void parse_check(int argc, char** argv) {
        prog_ = argc > 0 ? argv[0] : "prog";
        std::vector<std::string> errors;
        bool help = false;
        for (int i = 1; i < argc; i++) {
            std::string a = argv[i];
            Opt* o = nullptr;
            std::string val;
            bool has_val = false;
            if (a.rfind("--", 0) == 0) {
                std::string name = a.substr(2);
                size_t eq = name.find('=');
                if (eq != std::string::npos) { val = name.substr(eq + 1); name = name.substr(0, eq); has_val = true; }
                if (name == "help") { help = true; continue; }
                o = find_long(name);
                if (!o) { errors.push_back("undefined option: --" + name); continue; }
            } else if (a.size() >= 2 && a[0] == '-') {
                if (a[1] == '?') { help = true; continue; }
                o = find_short(a[a.size() - 1]);
                if (!o) { errors.push_back(std::string("undefined short option: -") + a[a.size() - 1]); continue; }
            } else {
                rest_.push_back(a);
                continue;
            }
            if (o->has_value) {
                if (!has_val) {
                    if (i + 1 >= argc) { errors.push_back("option needs value: --" + o->name); continue; }
                    val = argv[++i];
                }
                o->value = val;
            } else if (has_val) {
                errors.push_back("option does not take a value: --" + o->name);
            }
            o->set = true;
        }
        if (help) { usage(stdout); quit(0); }
        for (auto& o : opts_) if (o.need && !o.set) errors.push_back("need option: --" + o.name);
        if (!errors.empty()) {
            for (auto& e : errors) fprintf(stderr, "%s\n", e.c_str());
            usage(stderr);
            quit(1);
        }
    }

private:
    struct Opt { std::string name; char shortn; std::string desc; bool need; bool has_value; std::string value; bool set; };
    std::vector<Opt> opts_;
    std::vector<std::string> rest_;
    std::string prog_;
    Opt* find_long(const std::string& n) { for (auto& o : opts_) if (o.name == n) return &o; return nullptr; }
    Opt* find_short(char c) { for (auto& o : opts_) if (o.shortn && o.shortn == c) return &o; return nullptr; }
    void usage(FILE* f) const {
        fprintf(f, "usage: %s", prog_.c_str());
        for (auto& o : opts_) if (o.need) fprintf(f, " --%s=string", o.name.c_str());
        fprintf(f, " [options] ...\noptions:\n");
        for (auto& o : opts_) {
            if (o.shortn) fprintf(f, "  -%c, ", o.shortn); else fprintf(f, "      ");
            fprintf(f, "--%-16s %s%s\n", o.name.c_str(), o.desc.c_str(), o.has_value ? " (string)" : "");
        }
        fprintf(f, "  -?, --help             print this message\n");
    }
};

// ---------------------------------------------------------------------------------------------------
// nominal.ini
// ---------------------------------------------------------------------------------------------------
class Config {
public:
    int error = 0;   // 0 ok, -1 cannot open, >0 first bad line (inih); ParseError() < 0 is what the programs test
    explicit Config(const std::string& path) {
        FILE* f = fopen(path.c_str(), "r");
        if (!f) { error = -1; return; }
        char raw[200];             // INI_MAX_LINE: longer lines are split, as fgets does in the reference
        std::string section, prev;
        int lineno = 0;
        while (fgets(raw, sizeof(raw), f)) {
            lineno++;
            std::string s(raw);
            size_t p0 = 0;
            if (lineno == 1 && s.size() >= 3 && (unsigned char)s[0] == 0xEF && (unsigned char)s[1] == 0xBB && (unsigned char)s[2] == 0xBF) p0 = 3;
            size_t end = s.size();
            while (end > p0 && isspace((unsigned char)s[end - 1])) end--;
            size_t beg = p0;
            while (beg < end && isspace((unsigned char)s[beg])) beg++;
            const bool indented = beg > 0;     // "start > line" (also true after a BOM, like the original)
            std::string body = s.substr(beg, end - beg);
            if (body.empty()) continue;
            if (body[0] == ';' || body[0] == '#') continue;
            if (!prev.empty() && indented) { put(section, prev, body); continue; }
            if (body[0] == '[') {
                size_t c = cut(body, 1, ']');
                if (c < body.size() && body[c] == ']') { section = body.substr(1, c - 1).substr(0, 49); prev.clear(); }
                else if (!error) error = lineno;
                continue;
            }
            size_t c = cut(body, 0, '=');
            if (!(c < body.size() && body[c] == '=')) c = cut(body, 0, ':');
            if (c < body.size() && (body[c] == '=' || body[c] == ':')) {
                std::string name = rtrim(body.substr(0, c));
                std::string value = ltrim(body.substr(c + 1));
                size_t cc = cut(value, 0, '\0');
                if (cc < value.size() && value[cc] == ';') value = value.substr(0, cc);
                value = rtrim(value);
                prev = name.substr(0, 49);
                put(section, name, value);
            } else if (!error) error = lineno;
        }
        fclose(f);
    }
    std::string get(const std::string& s, const std::string& n, const std::string& d) const {
        auto it = kv_.find(key(s, n));
        return it == kv_.end() ? d : it->second;
    }
    long get_int(const std::string& s, const std::string& n, long d) const {
        std::string v = get(s, n, "");
        const char* c = v.c_str();
        char* e;
        long r = strtol(c, &e, 0);      // prefix parse: "1000;" -> 1000, "0x10" -> 16
        return e > c ? r : d;
    }
    bool get_bool(const std::string& s, const std::string& n, bool d) const {
        std::string v = get(s, n, "");
        std::transform(v.begin(), v.end(), v.begin(), ::tolower);
        if (v == "true" || v == "yes" || v == "on" || v == "1") return true;
        if (v == "false" || v == "no" || v == "off" || v == "0") return false;
        return d;                        // "true;" lands here
    }

private:
    std::map<std::string, std::string> kv_;
    static std::string key(const std::string& s, const std::string& n) {
        std::string k = s + "=" + n;
        std::transform(k.begin(), k.end(), k.begin(), ::tolower);
        return k;
    }
    void put(const std::string& s, const std::string& n, const std::string& v) {
        std::string& slot = kv_[key(s, n)];
        if (!slot.empty()) slot += "\n";
        slot += v;
    }
    static size_t cut(const std::string& s, size_t from, char c) {   // first `c` or whitespace-preceded ';'
        bool ws = false;
        size_t i = from;
        while (i < s.size() && s[i] != c && !(ws && s[i] == ';')) { ws = isspace((unsigned char)s[i]); i++; }
        return i;
    }
    static std::string rtrim(std::string s) { while (!s.empty() && isspace((unsigned char)s.back())) s.pop_back(); return s; }
    static std::string ltrim(const std::string& s) { size_t i = 0; while (i < s.size() && isspace((unsigned char)s[i])) i++; return s.substr(i); }
};

inline hinge_filter_params filter_params_from(const Config& c, bool has_qv) {   // filter.cpp:377-409
    hinge_filter_params p;
    p.reso = 40;
    p.cut_off = (int)c.get_int("filter", "cut_off", -1);
    p.min_cov = (int)c.get_int("filter", "min_cov", -1);
    p.est_cov = (int)c.get_int("filter", "ec", 0);
    p.theta = (int)c.get_int("filter", "theta", -1);
    p.coverage_fraction = (int)c.get_int("filter", "coverage_frac_repeat_annotation", 3);
    p.min_repeat_annotation = (int)c.get_int("filter", "min_repeat_annotation_threshold", 10);
    p.max_repeat_annotation = (int)c.get_int("filter", "max_repeat_annotation_threshold", 20);
    p.repeat_annotation_gap = (int)c.get_int("filter", "repeat_annotation_gap_threshold", 300);
    p.no_hinge_region = (int)c.get_int("filter", "no_hinge_region", 500);
    p.hinge_min_support = (int)c.get_int("filter", "hinge_min_support", 7);
    p.hinge_bin_pileup = (int)c.get_int("filter", "hinge_min_pileup", 7);
    p.hinge_unbridged = (int)c.get_int("filter", "hinge_unbridged", 6);
    p.hinge_tolerance = (int)c.get_int("filter", "hinge_tolerance_length", 100);
    p.use_qv_mask = (c.get_bool("filter", "use_qv", true) && has_qv) ? 1 : 0;
    p.use_coverage_mask = c.get_bool("filter", "coverage", true) ? 1 : 0;
    p.delete_telomere = ((int)c.get_int("layout", "del_telomere", 0)) != 0 ? 1 : 0;   // (sic) filter reads del_telomere
    return p;
}

// ---------------------------------------------------------------------------------------------------
// read-only memory map
// ---------------------------------------------------------------------------------------------------
// Host threads for the ingest / grouping / formatting loops.  HINGE_THREADS overrides; the loops are written so
// that the result does not depend on the number of threads (outputs are per item or per chunk, stitched in order).
inline int host_threads() {
    static int n = [] {
        if (const char* e = getenv("HINGE_THREADS")) { const int v = atoi(e); if (v > 0) return std::min(v, 256); }
        const unsigned hw = std::thread::hardware_concurrency();
        return (int)std::min<unsigned>(std::max<unsigned>(hw, 1u), 64u);
    }();
    return n;
}
// fn(chunk, begin, end) over [0, n) cut into `chunks` contiguous pieces, one thread each
template <typename F>
inline void parallel_chunks(int64_t n, int chunks, F fn) {
    chunks = (int)std::max<int64_t>(1, std::min<int64_t>(chunks, n));
    if (chunks == 1) { fn(0, (int64_t)0, n); return; }
    std::vector<std::thread> th;
    th.reserve((size_t)chunks);
    for (int c = 0; c < chunks; c++) {
        const int64_t b = n * c / chunks, e = n * (c + 1) / chunks;
        th.emplace_back([=, &fn] { fn(c, b, e); });
    }
    for (auto& t : th) t.join();
}

// fn(begin, end) over [0, n) in pieces of `grain` items handed out dynamically to host_threads() threads
template <typename F>
inline void parallel_dynamic(int64_t n, int64_t grain, F fn) {
    grain = std::max<int64_t>(1, grain);
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), (n + grain - 1) / grain));
    if (T == 1) { if (n > 0) fn((int64_t)0, n); return; }
    std::atomic<int64_t> next(0);
    std::vector<std::thread> th;
    th.reserve((size_t)T);
    for (int t = 0; t < T; t++)
        th.emplace_back([&] {
            for (;;) {
                const int64_t b = next.fetch_add(grain);
                if (b >= n) break;
                fn(b, std::min(n, b + grain));
            }
        });
    for (auto& t : th) t.join();
}

// uninitialised array for the big SoA columns: std::vector would zero (and page-fault) hundreds of MB on one thread
// before the parallel fill writes every element anyway
template <typename T>
struct UVec {
    T* p = nullptr;
    size_t n = 0;
    UVec() = default;
    UVec(const UVec&) = delete;
    UVec& operator=(const UVec&) = delete;
    ~UVec() { free(p); }
    // big columns on transparent huge pages (2 MiB): a 200 MB column is 100 page faults and 100 pages to free at exit instead
    // of 50 000 each (the fill threads fault the pages in; the kernel honours the hint where THP is 'madvise' or 'always')
    void resize(size_t m) {
        free(p);
        p = nullptr;
        n = 0;
        if (!m) return;
        const size_t bytes = m * sizeof(T), huge = (size_t)2 << 20;
        if (bytes >= 4 * huge) {
            void* q = nullptr;
            if (posix_memalign(&q, huge, (bytes + huge - 1) / huge * huge) == 0) {
#ifdef MADV_HUGEPAGE
                (void)madvise(q, (bytes + huge - 1) / huge * huge, MADV_HUGEPAGE);
#endif
                p = (T*)q;
            }
        }
        if (!p) p = (T*)malloc(bytes);
        n = p ? m : 0;
    }
    T* data() { return p; }
    const T* data() const { return p; }
    size_t size() const { return n; }
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
};

struct Mapped {
    const uint8_t* p = nullptr;
    size_t n = 0;
    bool open(const std::string& path) {
        int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) { ::close(fd); return false; }
        n = (size_t)st.st_size;
        if (n == 0) { ::close(fd); p = nullptr; return true; }
        void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        ::close(fd);
        if (m == MAP_FAILED) { n = 0; return false; }
        p = (const uint8_t*)m;
        return true;
    }
    // fault the pages in from host_threads() threads (one thread walking a 5 GB mapping pays ~0.5 us per 4 KiB page)
    void prefault() const {
        if (!p || n < (64u << 20)) return;
        const size_t page = 4096, pages = (n + page - 1) / page;
        parallel_dynamic((int64_t)pages, 4096, [&](int64_t b, int64_t e) {
            const size_t o = (size_t)b * page, len = std::min(n, (size_t)e * page) - o;
#ifdef MADV_POPULATE_READ
            if (madvise((void*)(p + o), len, MADV_POPULATE_READ) == 0) return;
#endif
            volatile uint8_t sink = 0;
            for (size_t q = o; q < o + len; q += page) sink += p[q];
            (void)sink;
        });
    }
    ~Mapped() { if (p) munmap((void*)p, n); }
};

template <typename T> inline T rd(const uint8_t* p) { T v; memcpy(&v, p, sizeof(T)); return v; }

// ---------------------------------------------------------------------------------------------------
// DAZZ_DB: trimmed read lengths + qual track
// ---------------------------------------------------------------------------------------------------
struct ReadDB {
    int ureads = 0, treads = 0, cutoff = 0, all = 1;
    std::vector<int32_t> rlen;        // trimmed order = the ids the .las uses
    std::vector<int64_t> boff;        // trimmed order: first byte of the read's bases in .<root>.bps (`hinge consensus`)
    std::vector<uint8_t> keep;        // per untrimmed read
    std::string dir, root;
    int nfiles = 0;

    static std::string dirname_of(const std::string& p) { size_t s = p.rfind('/'); return s == std::string::npos ? "." : p.substr(0, s); }
    static std::string rootname_of(const std::string& p) {
        size_t s = p.rfind('/');
        std::string b = s == std::string::npos ? p : p.substr(s + 1);
        if (b.size() > 3 && b.compare(b.size() - 3, 3, ".db") == 0) b.resize(b.size() - 3);
        return b;
    }
    // Open_DB + Trim_DB: 0 ok, -1 = the reference would exit(1)
    int open(const std::string& name) {
        dir = dirname_of(name);
        root = rootname_of(name);
        FILE* stub = fopen((dir + "/" + root + ".db").c_str(), "r");
        if (!stub) return -1;
        nfiles = 0;
        bool ok = fscanf(stub, "files = %9d\n", &nfiles) == 1;
        for (int i = 0; ok && i < nfiles; i++) {
            int last;
            char a[10000], b[10000];
            ok = fscanf(stub, "  %9d %s %s\n", &last, a, b) == 3;
        }
        int nblocks = 0;
        cutoff = 0; all = 1;
        if (ok && fscanf(stub, "blocks = %9d\n", &nblocks) == 1) {
            long long size;
            ok = fscanf(stub, "size = %9lld cutoff = %9d all = %1d\n", &size, &cutoff, &all) == 3;
        }
        fclose(stub);
        if (!ok) return -1;
        Mapped idx;
        if (!idx.open(dir + "/." + root + ".idx") || idx.n < 112) return -1;
        ureads = rd<int32_t>(idx.p);
        treads = rd<int32_t>(idx.p + 4);
        if (idx.n < 112 + (size_t)ureads * 40) return -1;
        const bool trim = !(cutoff <= 0 && all);
        const int allflag = all ? 0 : 0x800;                 // DB_BEST
        keep.assign(ureads, 1);
        rlen.clear();
        boff.clear();
        for (int i = 0; i < ureads; i++) {
            const uint8_t* r = idx.p + 112 + (size_t)i * 40;
            const int len = rd<int32_t>(r + 4), flags = rd<int32_t>(r + 32);
            if (trim && !((flags & 0x800) >= allflag && len >= cutoff)) { keep[i] = 0; continue; }
            rlen.push_back(len);
            boff.push_back(rd<int64_t>(r + 16));
        }
        return 0;
    }
    // getQV: false = no usable qual track (has_qv = false in the programs)
    bool load_qual(std::vector<std::vector<uint8_t>>& qv) const {
        Mapped anno, data;
        const std::string pre = dir + "/." + root + ".qual";
        if (!anno.open(pre + ".anno") || anno.n < 8) return false;
        const int tracklen = rd<int32_t>(anno.p);
        const int n = (int)rlen.size();
        if (tracklen != ureads && tracklen != treads) return false;
        if (anno.n < 8 + 8 * ((size_t)tracklen + 1)) return false;
        if (!data.open(pre + ".data")) return false;
        const bool untrimmed = tracklen == ureads && ureads != n;
        qv.clear();
        for (int i = 0; i < tracklen; i++) {
            if (untrimmed && !keep[i]) continue;
            const int64_t a = rd<int64_t>(anno.p + 8 + 8 * (size_t)i), b = rd<int64_t>(anno.p + 8 + 8 * ((size_t)i + 1));
            if (a < 0 || b < a || (size_t)b > data.n) return false;
            qv.emplace_back(data.p + a, data.p + b);
        }
        return (int)qv.size() == n;
    }
};

// QV mask, filter.cpp:309-312,340-369: longest run of good (< 40) segments; the last segment always breaks a run
inline void qv_masks(const std::vector<std::vector<uint8_t>>& qv, int tspace, std::vector<int32_t>& out) {
    out.assign(2 * qv.size(), 0);
    for (size_t i = 0; i < qv.size(); i++) {
        int s = 0, e = 0, mx = 0, maxs = 0, maxe = 0;
        const size_t n = qv[i].size();
        for (size_t j = 0; j < n; j++) {
            if (qv[i][j] < 40 && j < n - 1) e++;
            else {
                if (e - s > mx) { maxe = e; maxs = s; mx = e - s; }
                s = (int)j + 1;
                e = (int)j + 1;
            }
        }
        out[2 * i] = maxs * tspace;
        out[2 * i + 1] = maxe * tspace;
    }
}

// ---------------------------------------------------------------------------------------------------
// --fasta / --paf input (filter.cpp:289-291,499-503): reads = the records of a FASTA/FASTQ file in file order
// (only their lengths matter here), alignments = PAF lines whose read names carry the 1-based id between the
// first two '/' (LAInterface.cpp:4808-4845).  Both files go through zlib like the reference's kseq / paf.c.
// ---------------------------------------------------------------------------------------------------
inline bool slurp_gz(const std::string& path, std::string& out) {
    gzFile f = gzopen(path.c_str(), "r");
    if (!f) return false;
    gzbuffer(f, 1 << 20);
    std::vector<char> buf(1 << 22);
    int n;
    while ((n = gzread(f, buf.data(), (unsigned)buf.size())) > 0) out.append(buf.data(), (size_t)n);
    gzclose(f);
    return n >= 0;
}

// kseq_read (include/kseq.h:193-232 of the reference): a record starts at the next '>' or '@'; the sequence is every
// following non-empty line, whole, until a line that starts with '>', '+' or '@'; after '+' whole quality lines are read
// until they hold as many characters (a different count ends the file's parsing).
inline int read_fasta_lengths(const std::string& path, std::vector<int32_t>& rlen) {
    typedef int32_t RLEN_T;
    std::string t;
    if (!slurp_gz(path, t)) return -1;
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

// ---------------------------------------------------------------------------------------------------
struct LasPart {
    Mapped file;
    int64_t novl = 0;
    int tspace = 0, tbytes = 1;
    int r_begin = 0, r_end = -1;               // A read of the first / last record (filter.cpp:516-517)
    bool is_paf = false;                       // loaded by load_paf(): no traces, ProcessAlignment(trim = false)
    // kept (non-self) overlaps in .las order
    std::vector<int64_t> row_ptr;              // n_reads + 1
    UVec<int32_t> a_span, b_span;              // 2 per overlap
    UVec<uint32_t> b_flag;
    // what hinge_set_pileups_packed wants besides the columns (the fill pass touches every record anyway)
    bool want_span16 = false;                  // `hinge filter` only: the 16|16 copy of a_span, the stream of the two coverage passes
    UVec<uint32_t> span16;                     // abpos | aepos << 16, + HINGE_SPAN16_PAD spare elements; empty if unusable
    uint32_t max_pile = 0;                     // largest pile-up of the part
    bool spans_in_range = true;                // every (abpos, aepos) inside [0, rlen[A]]
    const uint32_t* span16_ptr() const { return span16.size() ? span16.data() : nullptr; }
    // ... and the bins of every read's plain coverage profile at reso 40 (hinge_set_pile_bins): 0 for an empty pile-up, else
    // max(abpos, aepos) / 40 + 2 over its overlaps, -1 for a coordinate outside [0, rlen] or 65 536+ overlaps
    std::vector<int32_t> nbins40;              // reads r_begin .. r_end
    void finish_facts(int n_reads, const std::vector<int32_t>* rlen = nullptr) {
        int64_t mp = 0;
        for (int q = 0; q < n_reads; q++) mp = std::max(mp, row_ptr[(size_t)q + 1] - row_ptr[(size_t)q]);
        max_pile = (uint32_t)std::min<int64_t>(mp, 0x7fffffff);
        nbins40.clear();
        if (!rlen || r_end < r_begin || r_end >= n_reads) return;
        nbins40.assign((size_t)(r_end - r_begin + 1), 0);
        parallel_dynamic((int64_t)nbins40.size(), 512, [&](int64_t k0, int64_t k1) {
            for (int64_t k = k0; k < k1; k++) {
                const int q = r_begin + (int)k;
                const int64_t s = row_ptr[(size_t)q], e = row_ptr[(size_t)q + 1];
                if (e == s) continue;
                const uint32_t rl = (uint32_t)std::max((*rlen)[(size_t)q], 0);
                int32_t mx = 0;
                bool ok = e - s < 65536;
                for (int64_t t = s; t < e; t++) {
                    const int32_t ab = a_span[(size_t)(2 * t)], ae = a_span[(size_t)(2 * t + 1)];
                    ok = ok && (uint32_t)ab <= rl && (uint32_t)ae <= rl;
                    mx = std::max(mx, std::max(ab, ae));
                }
                nbins40[(size_t)k] = ok ? mx / 40 + 2 : -1;
            }
        });
    }
    UVec<int64_t> trace_off;                   // byte offset of the trace inside the mapped file
    UVec<int32_t> tlen;
    // every record (self-overlaps included), for the (A, B) grouping of maximal / layout
    std::vector<int64_t> rec_row_ptr;          // n_reads + 1 over records
    UVec<int32_t> rec_b;
    UVec<int64_t> rec_kept;                    // index into the kept arrays, -1 for a self-overlap
    // self-overlaps (filter.cpp:538-544)
    std::vector<int32_t> self_a, self_span;    // self_span: abpos, aepos, bbpos', bepos'

    int64_t n_kept() const { return (int64_t)b_flag.size(); }

    // What hinge_set_las_image wants besides the mapped file: the kept overlaps in windows of 64 - where every window's first
    // record lies in the image (last entry: where the last kept overlap ends) and every record's offset behind that, in 32 bits.
    // false: a window of 4 GiB+ - the caller keeps hinge_set_traces.
    std::vector<int64_t> img_win_base;
    UVec<uint32_t> img_rec_rel;
    bool build_image_table() {
        const int64_t kept = n_kept();
        if (is_paf || kept == 0 || trace_off.size() != (size_t)kept) return false;
        const int64_t nw = (kept + 63) / 64;
        img_win_base.assign((size_t)nw + 1, 0);
        img_win_base[(size_t)nw] = trace_off[(size_t)kept - 1] + (int64_t)tlen[(size_t)kept - 1] * tbytes;
        img_rec_rel.resize((size_t)kept);
        std::atomic<int> bad(0);
        parallel_dynamic(nw, 1024, [&](int64_t w0, int64_t w1) {
            int64_t acc = 0;
            for (int64_t w = w0; w < w1; w++) {
                const int64_t wb = trace_off[(size_t)(64 * w)] - 40;
                img_win_base[(size_t)w] = wb;
                const int64_t e = std::min(kept, 64 * w + 64);
                for (int64_t k = 64 * w; k < e; k++) {
                    const int64_t d = trace_off[(size_t)k] - 40 - wb;
                    acc |= d >> 32;                       // (negative: the sign bits)
                    img_rec_rel[(size_t)k] = (uint32_t)d;
                }
            }
            if (acc) bad = 1;
        });
        return bad == 0;
    }

    static int header(const std::string& path, int64_t& novl, int& tspace) {
        FILE* f = fopen(path.c_str(), "rb");
        if (!f) return -1;
        int ok = fread(&novl, 8, 1, f) == 1 && fread(&tspace, 4, 1, f) == 1;
        fclose(f);
        return ok ? 0 : -1;
    }

    // A record that could start at byte p: every field in range for this DB.  Only used to GUESS where a thread may
    // start walking; guesses are verified against the true chain below, so a wrong guess costs time, never correctness.
    bool plausible(size_t p, const std::vector<int32_t>& rlen) const {
        if (p + 40 > file.n) return false;
        const uint8_t* r = file.p + p;
        const int tl = rd<int32_t>(r), abpos = rd<int32_t>(r + 8), bbpos = rd<int32_t>(r + 12), aepos = rd<int32_t>(r + 16),
                  bepos = rd<int32_t>(r + 20), a = rd<int32_t>(r + 28), b = rd<int32_t>(r + 32);
        const uint32_t flags = rd<uint32_t>(r + 24);
        const int n_reads = (int)rlen.size();
        if (tl < 0 || (tl & 1) || p + 40 + (size_t)tl * tbytes > file.n) return false;
        if (a < 0 || a >= n_reads || b < 0 || b >= n_reads || flags >= 64u) return false;
        if (abpos < 0 || abpos >= aepos || aepos > rlen[(size_t)a] || bbpos < 0 || bbpos >= bepos || bepos > rlen[(size_t)b]) return false;
        if (tl / 2 != (aepos + tspace - 1) / tspace - abpos / tspace) return false;   // one (diffs, b-advance) pair per tspace panel of A
        return true;
    }

    // off[j] = byte offset of record j.  Every record's position depends on the previous tlen, so one thread walking a
    // 5 GB file is ~0.5 s.  Instead the file is cut into byte ranges; each thread guesses the first record start in its
    // range (first position from which 16 consecutive records look plausible), walks to the end of its range, and the
    // pieces are accepted only if every walk ends exactly where the next one started.  Anything else (an unusual
    // writer, corrupt data, a false guess) falls back to the sequential walk, which also produces the error codes.
    mutable bool indexed_in_pieces = false;   // diagnostics: the verified multi-thread walk was used
    bool index_records(UVec<int64_t>& off, const std::vector<int32_t>& rlen) const {
        const size_t fn = file.n;
        const int T = host_threads();
        const size_t body = fn - 12;
        if (T > 1 && novl >= (int64_t)T * 64 && body / (size_t)T > (64u << 10)) {
            const int S = T;
            std::vector<size_t> start((size_t)S + 1, 0), stop((size_t)S, 0);
            std::vector<std::vector<int64_t>> pieces((size_t)S);
            std::vector<char> ok((size_t)S, 0);
            start[0] = 12;
            start[(size_t)S] = fn;
            parallel_chunks(S, S, [&](int c, int64_t, int64_t) {
                size_t p = 12 + body * (size_t)c / (size_t)S;
                if (c > 0) {   // guess: first byte position from which 16 records in a row are plausible
                    const size_t limit = std::min(fn, p + (1u << 20));
                    bool found = false;
                    for (; p < limit; p++) {
                        size_t q = p;
                        int good = 0;
                        while (good < 16 && q < fn && plausible(q, rlen)) { q += 40 + (size_t)rd<int32_t>(file.p + q) * tbytes; good++; }
                        if (good == 16 || (good > 0 && q == fn)) { found = true; break; }
                    }
                    if (!found) return;
                    start[(size_t)c] = p;
                }
                const size_t end = c + 1 < S ? 12 + body * (size_t)(c + 1) / (size_t)S : fn;
                std::vector<int64_t>& v = pieces[(size_t)c];
                v.reserve((size_t)(novl / S + novl / (8 * S) + 1024));
                while (p < end) {
                    if (p + 40 > fn) return;
                    const int tl = rd<int32_t>(file.p + p);
                    if (tl < 0) return;
                    v.push_back((int64_t)p);
                    p += 40 + (size_t)tl * tbytes;
                }
                stop[(size_t)c] = p;
                ok[(size_t)c] = 1;
            });
            bool all = true;
            int64_t total = 0;
            for (int c = 0; c < S && all; c++) {
                all = ok[(size_t)c] && stop[(size_t)c] == start[(size_t)c + 1];
                total += (int64_t)pieces[(size_t)c].size();
            }
            if (all && total == novl) {
                std::vector<int64_t> base((size_t)S + 1, 0);
                for (int c = 0; c < S; c++) base[(size_t)c + 1] = base[(size_t)c] + (int64_t)pieces[(size_t)c].size();
                parallel_chunks(S, S, [&](int c, int64_t, int64_t) {
                    if (!pieces[(size_t)c].empty()) memcpy(off.data() + base[(size_t)c], pieces[(size_t)c].data(), pieces[(size_t)c].size() * sizeof(int64_t));
                });
                indexed_in_pieces = true;
                return true;
            }
        }
        size_t pos = 12;
        for (int64_t j = 0; j < novl; j++) {
            if (pos + 40 > fn) return false;
            off[(size_t)j] = (int64_t)pos;
            const int tl = rd<int32_t>(file.p + pos);
            if (tl < 0) return false;
            pos += 40 + (size_t)tl * tbytes;
        }
        return pos <= fn;
    }

    // PAF input: same columns as load(), no traces (tlen 0).  Records may come in any order: the reference files every
    // alignment under its A read in file order (filter.cpp:529-548), which is a stable sort by A; r_begin / r_end are
    // the A reads of the first / last LINE (filter.cpp:515-516).
    // 0 ok; -1 cannot open; -3 a read name without "/id/" or an id outside the FASTA (the reference crashes)
    int load_paf(const std::string& path, const std::vector<int32_t>& rlen) {
        PhaseTimer lt("  paf.load");
        std::string t;
        if (!slurp_gz(path, t)) return -1;
        tspace = 100; tbytes = 1;
        const int n_reads = (int)rlen.size();
        struct Rec { int a, b, ab, ae, bb, be, comp; };
        std::vector<Rec> recs;
        auto id_of = [](const char* q, const char* qe, bool& ok) {   // get_id_from_string - 1
            const char* s0 = (const char*)memchr(q, '/', (size_t)(qe - q));
            if (!s0) { ok = false; return 0; }
            const char* s1 = s0 + 1;
            const char* s2 = (const char*)memchr(s1, '/', (size_t)(qe - s1));
            if (!s2 || s2 - s1 >= 15) { ok = false; return 0; }
            char sub[16];
            memcpy(sub, s1, (size_t)(s2 - s1)); sub[s2 - s1] = 0;
            return atoi(sub) - 1;
        };
        size_t i = 0;
        const size_t n = t.size();
        while (i < n) {
            size_t e = t.find('\n', i);
            if (e == std::string::npos) e = n;
            size_t le = e;
            if (le > i && t[le - 1] == '\r') le--;
            const char* f[12];
            const char* fe[12];
            int nf = 0;
            size_t p0 = i;
            for (size_t k = i; k <= le && nf < 12; k++)
                if (k == le || t[k] == '\t') { f[nf] = t.data() + p0; fe[nf] = t.data() + k; nf++; p0 = k + 1; }
            // (fields beyond the 12th are never looked at; a line needs at least 10)
            int total = nf;
            if (nf == 12) for (size_t k = (size_t)(fe[11] - t.data()); k < le; k++) if (t[k] == '\t') total++;
            i = e + 1;
            if (total < 10) continue;
            auto num = [&](int k) { char tmp[24]; size_t l = std::min<size_t>((size_t)(fe[k] - f[k]), 23); memcpy(tmp, f[k], l); tmp[l] = 0; return (int)(uint32_t)strtol(tmp, nullptr, 10); };
            bool ok = true;
            Rec r;
            r.ab = num(2); r.ae = num(3); r.bb = num(7); r.be = num(8);
            r.comp = (fe[4] > f[4] && *f[4] == '-') ? 1 : 0;
            r.a = id_of(f[0], fe[0], ok);
            r.b = id_of(f[5], fe[5], ok);
            if (!ok || r.a < 0 || r.a >= n_reads || r.b < 0 || r.b >= n_reads) return -3;
            recs.push_back(r);
        }
        lt.mark("parse");
        novl = (int64_t)recs.size();
        if (novl > 0) { r_begin = recs.front().a; r_end = recs.back().a; }
        // stable counting sort by A
        rec_row_ptr.assign((size_t)n_reads + 1, 0);
        row_ptr.assign((size_t)n_reads + 1, 0);
        for (auto& r : recs) { rec_row_ptr[(size_t)r.a + 1]++; if (r.a != r.b) row_ptr[(size_t)r.a + 1]++; }
        for (int q = 0; q < n_reads; q++) { rec_row_ptr[(size_t)q + 1] += rec_row_ptr[(size_t)q]; row_ptr[(size_t)q + 1] += row_ptr[(size_t)q]; }
        const int64_t kept = row_ptr[(size_t)n_reads];
        a_span.resize((size_t)kept * 2); b_span.resize((size_t)kept * 2); b_flag.resize((size_t)kept);
        trace_off.resize((size_t)kept); tlen.resize((size_t)kept);
        rec_b.resize((size_t)novl); rec_kept.resize((size_t)novl);
        std::vector<int64_t> next_rec(rec_row_ptr.begin(), rec_row_ptr.end() - 1), next_kept(row_ptr.begin(), row_ptr.end() - 1);
        std::vector<std::pair<int64_t, Rec>> selfs;   // (position among the sorted records, record): self_a order = pile-up order
        for (auto& r : recs) {
            const int64_t j = next_rec[(size_t)r.a]++;
            rec_b[(size_t)j] = r.b;
            if (r.a == r.b) { rec_kept[(size_t)j] = -1; selfs.emplace_back(j, r); continue; }
            const int64_t k = next_kept[(size_t)r.a]++;
            rec_kept[(size_t)j] = k;
            a_span[(size_t)k * 2] = r.ab; a_span[(size_t)k * 2 + 1] = r.ae;
            b_span[(size_t)k * 2] = r.bb; b_span[(size_t)k * 2 + 1] = r.be;   // no strand flip: PAF target coordinates are forward-strand
            b_flag[(size_t)k] = (uint32_t)r.b | ((uint32_t)r.comp << 31);
            trace_off[(size_t)k] = 0; tlen[(size_t)k] = 0;
            const uint32_t rl = (uint32_t)std::max(rlen[(size_t)r.a], 0);
            if ((uint32_t)r.ab > rl || (uint32_t)r.ae > rl) spans_in_range = false;
        }
        finish_facts(n_reads, want_span16 ? &rlen : nullptr);
        if (want_span16 && kept > 0 && spans_in_range && *std::max_element(rlen.begin(), rlen.end()) < 65536) {
            span16.resize((size_t)kept + (size_t)HINGE_SPAN16_PAD);
            for (int64_t k = 0; k < kept; k++) span16[(size_t)k] = (uint32_t)a_span[(size_t)k * 2] | ((uint32_t)a_span[(size_t)k * 2 + 1] << 16);
            for (int t = 0; t < HINGE_SPAN16_PAD; t++) span16[(size_t)kept + (size_t)t] = 0;
        }
        // self overlaps in FILE order (filter.cpp:537-544 walks aln[] in file order)
        self_a.resize(selfs.size()); self_span.resize(selfs.size() * 4);
        {
            size_t si = 0;
            for (auto& r : recs)
                if (r.a == r.b) { self_a[si] = r.a; self_span[si * 4] = r.ab; self_span[si * 4 + 1] = r.ae; self_span[si * 4 + 2] = r.bb; self_span[si * 4 + 3] = r.be; si++; }
        }
        is_paf = true;
        return 0;
    }

    // 0 ok; -1 cannot open / truncated (reference: exit(1)); -2 records not grouped by ascending A read
    // The record chain (every record's position depends on the previous tlen) is walked once; everything else -
    // validation, the strand flip of LAInterface.cpp:1619-1626, the SoA fill, both CSR row tables - runs on
    // host_threads() threads over contiguous record ranges.
    // pairs = false: skip the columns only maximal / layout read (per-record B and kept index, trace offsets, tlen)
    int load(const std::string& path, const std::vector<int32_t>& rlen, bool pairs = true) {
        PhaseTimer lt("  las.load");
        if (!file.open(path) || file.n < 12) return -1;
        file.prefault();
        lt.mark("prefault");
        novl = rd<int64_t>(file.p);
        tspace = rd<int32_t>(file.p + 8);
        tbytes = tspace <= 125 ? 1 : 2;            // TRACE_XOVR
        if (novl < 0 || (uint64_t)novl > file.n / 40) return -1;
        const int n_reads = (int)rlen.size();
        UVec<int64_t> off;
        off.resize((size_t)novl);
        if (novl > 0 && !off.data()) return -1;
        if (!index_records(off, rlen)) return -1;
        lt.mark(indexed_in_pieces ? "record chain (pieces)" : "record chain (1 thread)");
        row_ptr.assign((size_t)n_reads + 1, 0);
        rec_row_ptr.assign((size_t)n_reads + 1, 0);
        if (pairs) { rec_b.resize((size_t)novl); rec_kept.resize((size_t)novl); }
        const int T = host_threads();
        std::vector<int64_t> self_cnt((size_t)T + 1, 0);
        std::vector<int> err((size_t)T, 0);
        auto a_of = [&](int64_t j) { return rd<int32_t>(file.p + off[(size_t)j] + 28); };
        // phase A: validate, count self-overlaps per chunk
        int chunks = (int)std::max<int64_t>(1, std::min<int64_t>(T, novl));
        parallel_chunks(novl, chunks, [&](int c, int64_t j0, int64_t j1) {
            int prev_a = j0 > 0 ? a_of(j0 - 1) : -1;
            int64_t ns = 0;
            for (int64_t j = j0; j < j1; j++) {
                const uint8_t* r = file.p + off[(size_t)j];
                const int a = rd<int32_t>(r + 28), b = rd<int32_t>(r + 32);
                if (a < 0 || a >= n_reads || b < 0 || b >= n_reads) { err[(size_t)c] = -1; return; }
                if (a < prev_a) { if (err[(size_t)c] == 0) err[(size_t)c] = -2; }
                prev_a = a;
                ns += a == b;
            }
            self_cnt[(size_t)c + 1] = ns;
        });
        for (int c = 0; c < chunks; c++) if (err[(size_t)c] == -1) return -1;
        for (int c = 0; c < chunks; c++) if (err[(size_t)c] == -2) return -2;
        for (int c = 0; c < chunks; c++) self_cnt[(size_t)c + 1] += self_cnt[(size_t)c];
        lt.mark("validate");
        const int64_t n_self = self_cnt[(size_t)chunks], kept = novl - n_self;
        a_span.resize((size_t)kept * 2);
        b_span.resize((size_t)kept * 2);
        b_flag.resize((size_t)kept);
        if (pairs) { trace_off.resize((size_t)kept); tlen.resize((size_t)kept); }
        self_a.resize((size_t)n_self);
        self_span.resize((size_t)n_self * 4);
        const bool fill16 = want_span16 && kept > 0 && !rlen.empty() && *std::max_element(rlen.begin(), rlen.end()) < 65536;
        if (fill16) span16.resize((size_t)kept + (size_t)HINGE_SPAN16_PAD);
        std::vector<char> out_of_range((size_t)chunks, 0);
        lt.mark("allocate");
        // phase B: fill.  Records are sorted by A, so row tables are written at the A boundaries:
        // row_ptr[r] = kept records with aread < r, rec_row_ptr[r] = records with aread < r
        parallel_chunks(novl, chunks, [&](int c, int64_t j0, int64_t j1) {
            int64_t si = self_cnt[(size_t)c];
            int64_t k = j0 - si;
            int prev_a = j0 > 0 ? a_of(j0 - 1) : -1;
            for (int64_t j = j0; j < j1; j++) {
                const uint8_t* r = file.p + off[(size_t)j];
                const int tl = rd<int32_t>(r), abpos = rd<int32_t>(r + 8), bbpos = rd<int32_t>(r + 12), aepos = rd<int32_t>(r + 16),
                          bepos = rd<int32_t>(r + 20), a = rd<int32_t>(r + 28), b = rd<int32_t>(r + 32);
                const uint32_t flags = rd<uint32_t>(r + 24);
                if (a != prev_a) {
                    for (int q = prev_a + 1; q <= a; q++) { rec_row_ptr[(size_t)q] = j; row_ptr[(size_t)q] = k; }
                    prev_a = a;
                }
                const int comp = (int)(flags & 1u);                       // COMP()
                int bb = bbpos, be = bepos;
                if (comp) { bb = rlen[(size_t)b] - bepos; be = rlen[(size_t)b] - bbpos; }   // LAInterface.cpp:1619-1626
                if (pairs) rec_b[(size_t)j] = b;
                if (a == b) {
                    if (pairs) rec_kept[(size_t)j] = -1;
                    self_a[(size_t)si] = a;
                    self_span[(size_t)si * 4] = abpos; self_span[(size_t)si * 4 + 1] = aepos; self_span[(size_t)si * 4 + 2] = bb; self_span[(size_t)si * 4 + 3] = be;
                    si++;
                    continue;
                }
                if (pairs) rec_kept[(size_t)j] = k;
                {
                    const uint32_t rl = (uint32_t)std::max(rlen[(size_t)a], 0);
                    if ((uint32_t)abpos > rl || (uint32_t)aepos > rl) out_of_range[(size_t)c] = 1;   // unsigned: negative = too large
                }
                if (fill16) span16[(size_t)k] = ((uint32_t)abpos & 0xffffu) | ((uint32_t)aepos << 16);
                a_span[(size_t)k * 2] = abpos; a_span[(size_t)k * 2 + 1] = aepos;
                b_span[(size_t)k * 2] = bb; b_span[(size_t)k * 2 + 1] = be;
                b_flag[(size_t)k] = (uint32_t)b | ((uint32_t)comp << 31);
                if (pairs) { trace_off[(size_t)k] = off[(size_t)j] + 40; tlen[(size_t)k] = tl; }
                k++;
            }
        });
        lt.mark("fill");
        {
            const int last_a = novl > 0 ? a_of(novl - 1) : -1;
            for (int q = last_a + 1; q <= n_reads; q++) { rec_row_ptr[(size_t)q] = novl; row_ptr[(size_t)q] = kept; }
        }
        if (novl > 0) {
            r_begin = a_of(0);
            r_end = a_of(novl - 1);
        }
        for (int c = 0; c < chunks; c++) if (out_of_range[(size_t)c]) spans_in_range = false;
        finish_facts(n_reads, want_span16 ? &rlen : nullptr);
        if (fill16) {
            if (!spans_in_range) span16.resize(0);
            else for (int t = 0; t < HINGE_SPAN16_PAD; t++) span16[(size_t)kept + (size_t)t] = 0;
        }
        return 0;
    }
};

// decimal text of v appended at p (no terminator); returns the new end
inline char* put_int(char* p, long long v) {
    unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
    if (v < 0) *p++ = '-';
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    while (n) *p++ = tmp[--n];
    return p;
}

// `.coverage.txt`: "read i p,c p,c ...\n" per read (filter.cpp:599-602, maximal.cpp:659-685).  nb[k] bins of read
// r_begin + k start at cov[sum(nb[<k])].  Formatted by host_threads() threads in windows, written in read order.
// at != nullptr: the bins of read k start at cov[(*at)[k]] (K2's strided layout) instead of back to back.
inline void write_coverage_txt(FILE* f, int r_begin, const std::vector<int32_t>& nb, const UVec<int32_t>& cov, int reso, const std::vector<int64_t>* at = nullptr) {
    const int64_t nr = (int64_t)nb.size();
    std::vector<int64_t> first((size_t)nr + 1, 0);   // cumulative bins: sizes the text buffers
    for (int64_t k = 0; k < nr; k++) first[(size_t)k + 1] = first[(size_t)k] + nb[(size_t)k];
    const std::vector<int64_t>& where = at ? *at : first;
    const int64_t chunk = 256, window = (int64_t)host_threads() * 8;
    const int64_t n_chunks = (nr + chunk - 1) / chunk;
    // uninitialised chunk buffers (a value-initialised vector<char> of the upper bound would zero and fault in three
    // times the bytes that are written)
    struct Raw { char* p = nullptr; size_t cap = 0, len = 0; ~Raw() { free(p); } };
    std::vector<Raw> buf((size_t)std::min(window, std::max<int64_t>(n_chunks, 1)));
    for (int64_t w0 = 0; w0 < n_chunks; w0 += window) {
        const int64_t w1 = std::min(n_chunks, w0 + window);
        parallel_dynamic(w1 - w0, 1, [&](int64_t c0, int64_t c1) {
            for (int64_t c = c0; c < c1; c++) {
                const int64_t k0 = (w0 + c) * chunk, k1 = std::min(nr, k0 + chunk);
                Raw& b = buf[(size_t)c];
                const size_t need = (size_t)((k1 - k0) * 32 + (first[(size_t)k1] - first[(size_t)k0]) * 24);   // upper bound
                if (need > b.cap) { free(b.p); b.p = (char*)malloc(need); b.cap = b.p ? need : 0; }
                if (!b.p) { fprintf(stderr, "out of memory\n"); abort(); }
                char* p = b.p;
                for (int64_t k = k0; k < k1; k++) {
                    memcpy(p, "read ", 5); p += 5;
                    p = put_int(p, r_begin + k);
                    *p++ = ' ';
                    const int32_t* c32 = cov.data() + where[(size_t)k];
                    for (int j = 0; j < nb[(size_t)k]; j++) {
                        p = put_int(p, (long long)reso * j);
                        *p++ = ',';
                        p = put_int(p, c32[j]);
                        *p++ = ' ';
                    }
                    *p++ = '\n';
                }
                b.len = (size_t)(p - b.p);
            }
        });
        // the chunks of a window go to the file at their final offsets from all threads (one thread copying ~150 MB into the
        // page cache takes as long as formatting them did); a stream that cannot seek gets them one after the other
        fflush(f);
        const off_t base = ftello(f);
        const int fd = fileno(f);
        std::vector<int64_t> at((size_t)(w1 - w0) + 1, 0);
        for (int64_t c = 0; c < w1 - w0; c++) at[(size_t)c + 1] = at[(size_t)c] + (int64_t)buf[(size_t)c].len;
        bool positioned = base >= 0 && fd >= 0;
        if (positioned) {
            std::atomic<int> failed{0};
            parallel_dynamic(w1 - w0, 1, [&](int64_t c0, int64_t c1) {
                for (int64_t c = c0; c < c1; c++) {
                    const char* q = buf[(size_t)c].p;
                    size_t left = buf[(size_t)c].len;
                    off_t o = base + (off_t)at[(size_t)c];
                    while (left > 0) {
                        const ssize_t w = pwrite(fd, q, left, o);
                        if (w <= 0) { failed = 1; break; }
                        q += w; left -= (size_t)w; o += w;
                    }
                }
            });
            if (failed || fseeko(f, base + (off_t)at[(size_t)(w1 - w0)], SEEK_SET) != 0) { fprintf(stderr, "write error on the coverage file\n"); quit(1); }
        } else {
            for (int64_t c = 0; c < w1 - w0; c++) fwrite(buf[(size_t)c].p, 1, buf[(size_t)c].len, f);
        }
    }
}

// `hinge pipeline` (round 4): filter -> maximal -> layout in ONE process (host/pipeline_main.cpp includes the three stage
// programs as functions).  What the stages then share: the HIP runtime's start-up (one hipInit instead of three: 0.15-0.3 s
// each right after another GPU process has exited), ONE ingest of a single .las (the part is kept and handed to the later
// stages), one process teardown.  Everything else - argv, files, exit codes - is the separate programs'.
struct PipelineState {
    bool on = false;
    struct LasPart* part0 = nullptr;      // the single .las of the run, loaded once with everything any stage needs
    std::string part0_path;
};
inline PipelineState& pipeline() { static PipelineState s; return s; }

// HIP runtime initialisation (80-250 ms) and the first .las part's ingest (CPU only) run side by side: the context is
// created on a helper thread as soon as the arguments are parsed and joined right before its first use.
struct CtxInit {
    hinge_ctx* ctx = nullptr;
    int rc = HINGE_OK;
    std::thread t;
    void start() { t = std::thread([this] { rc = hinge_ctx_create(0, &ctx); }); }
    int join() { if (t.joinable()) t.join(); return rc; }
    ~CtxInit() { if (t.joinable()) t.join(); }
};
// Ranks (host thread + context) of a run: one per visible GPU for a --mlas set, HINGE_RANKS overrides.  A single part - the
// common case - never asks: hipGetDeviceCount() waits for the runtime start-up that CtxInit runs in the background, and the
// ingest of the first part is meant to run UNDER that start-up, not behind it.
static inline int rank_count(size_t n_parts, bool sequential_only) {
    if (n_parts <= 1 || sequential_only) return 1;
    const char* e = getenv("HINGE_RANKS");
    const int n = e ? atoi(e) : hinge_device_count();
    return std::max(1, std::min(n, (int)n_parts));
}
// What the ranks of a --mlas wave hand to the one sequential pass that follows them (containment candidates, classified matches):
// with a communicator over the ranks' GPUs the rows are all-gathered over RCCL (hinge_comm_allgather_rows) and the pass reads the
// gathered copy; refused (ranks share a device, no librccl, HINGE_HOST_EXCHANGE=1) they are concatenated on the host - the same
// bytes either way (tools/mlas_rccl_check.py runs both and compares the stage's files).  HINGE_COMM_ONE_RANK=1 sends a one-rank
// run's rows through a one-rank communicator too (what a 1-GPU box can exercise of this path).
struct RowGather {
    bool rccl = false;
    std::vector<hinge_ctx*>* ctxs = nullptr;
    long long waves = 0, rows_moved = 0;
    Log* log = nullptr;
    void init(std::vector<hinge_ctx*>& c, const char* what, Log& console) {
        ctxs = &c;
        log = &console;
        const bool one = getenv("HINGE_COMM_ONE_RANK") && atoi(getenv("HINGE_COMM_ONE_RANK")) != 0;
        rccl = (c.size() > 1 || one) && hinge_comm_create(c.data(), (int32_t)c.size()) == HINGE_OK;
        if (c.size() > 1 || one) console.info("%zu ranks, %s %s", c.size(), what, rccl ? "over RCCL (one all-gather per wave)" : "through the host");
    }
    // parts[k] = rank k's rows (row_bytes each); returns all rows in rank order and, in offs[k], where rank k's begin (in rows)
    bool gather(const std::vector<const void*>& parts, const std::vector<int64_t>& counts, int row_bytes, std::vector<char>& out, std::vector<int64_t>& offs) {
        const size_t nw = parts.size();
        offs.assign(nw + 1, 0);
        for (size_t k = 0; k < nw; k++) offs[k + 1] = offs[k] + counts[k];
        out.assign((size_t)std::max<int64_t>(offs[nw], 1) * (size_t)row_bytes, 0);
        if (rccl && nw == ctxs->size()) {
            std::vector<int64_t> got(nw, 0);
            if (hinge_comm_allgather_rows(ctxs->data(), (int32_t)nw, parts.data(), counts.data(), row_bytes, out.data(), offs[nw], got.data()) == HINGE_OK) {
                waves++; rows_moved += offs[nw];
                return true;
            }
            log->warn("row all-gather failed (%s): this wave's rows go through the host", hinge_last_error((*ctxs)[0]));
        }
        for (size_t k = 0; k < nw; k++)
            if (counts[k]) memcpy(out.data() + (size_t)offs[k] * (size_t)row_bytes, parts[k], (size_t)counts[k] * (size_t)row_bytes);
        return false;
    }
    void report(const char* what) const {
        if (waves) log->info("%s: %lld rows of %lld wave(s) exchanged over RCCL", what, rows_moved, waves);
    }
};

// the first part is loaded before the context is joined; later parts when their turn comes
struct PartLoader {
    std::unique_ptr<LasPart> first;
    int first_rc = 0;
    bool pairs = true;   // false in `hinge filter`: see LasPart::load
    bool paf = false;    // the (single) part is a PAF file
    bool span16 = false; // `hinge filter`: also produce the 16|16 span copy (LasPart::want_span16)
    bool shared = false; // `hinge pipeline`, single .las: the part belongs to the process, the caller must not delete it
    void preload(const std::string& path, const std::vector<int32_t>& rlen) {
        PipelineState& pl = pipeline();
        if (pl.on && !paf && single) {          // one ingest for all stages: the first one loads the superset (records' B column + the 16|16 span copy)
            if (!pl.part0 || pl.part0_path != path) {
                LasPart* p = new LasPart();
                p->want_span16 = true;
                first_rc = p->load(path, rlen, true);
                if (first_rc != 0) { first.reset(p); return; }     // (a failed load is reported by the stage as usual; nothing is kept)
                pl.part0 = p; pl.part0_path = path;
            } else first_rc = 0;
            first.reset(pl.part0);
            shared = true;
            return;
        }
        first.reset(new LasPart()); first->want_span16 = span16; first_rc = paf ? first->load_paf(path, rlen) : first->load(path, rlen, pairs);
    }
    bool single = false; // the run has exactly one part (set by the stage before preload)
    ~PartLoader() { if (shared) (void)first.release(); }
    // returns the part (ownership passes to the caller - unless `shared`) and its load() code
    LasPart* take(size_t part, const std::string& path, const std::vector<int32_t>& rlen, int& rc) {
        if (part == 0 && first) { rc = first_rc; return first.release(); }
        LasPart* p = new LasPart();
        p->want_span16 = span16;
        rc = paf ? p->load_paf(path, rlen) : p->load(path, rlen, pairs);
        return p;
    }
};

// Every output file is closed by now.  Leaving through _exit() skips unmapping the .las (GBs of page tables), freeing
// the SoA columns and the HIP runtime's own teardown: ~0.1 s of a sub-second run.  HINGE_SLOW_EXIT=1 keeps the
// orderly path (leak checkers).
inline int finish(hinge_ctx* ctx, PhaseTimer& tm, int code = 0) {
    // the next stage follows in this process: the context (its device buffers: a few GB of 288) is left to the process's exit -
    // giving them back costs the NEXT stage 25-30 ms of hipFree (round 5, tools/probes/pipeline_timing.sh); HINGE_SLOW_EXIT=1 frees
    // (the stages' buffers then add up - on a device that is running short (less than HINGE_PIPELINE_KEEP_FRACTION, default a
    // quarter, of its memory free) the context IS destroyed: the next stage pays the hipFree, and fits)
    if (pipeline().on) {
        tm.mark("(stage end)");
        bool give_back = getenv("HINGE_SLOW_EXIT") != nullptr;
        int64_t fr = 0, tot = 0;
        if (!give_back && hinge_ctx_device_memory(ctx, &fr, &tot) == HINGE_OK && tot > 0) {
            const double keep = getenv("HINGE_PIPELINE_KEEP_FRACTION") ? atof(getenv("HINGE_PIPELINE_KEEP_FRACTION")) : 0.25;
            give_back = (double)fr < keep * (double)tot;
        }
        if (give_back) hinge_ctx_destroy(ctx);
        return code;
    }
    if (getenv("HINGE_SLOW_EXIT")) { hinge_ctx_destroy(ctx); return code; }
    tm.mark("(exit)");
    tm.~PhaseTimer();
    fflush(nullptr);
    _exit(code);
}

inline std::string las_name(const std::string& base, bool mlas) {   // filter.cpp:228-241
    if (mlas) return base;
    if (base.size() >= 4 && base.compare(base.size() - 4, 4, ".las") == 0) return base;
    return base + ".las";
}
inline std::vector<std::string> las_parts(const std::string& base) {   // glob(), filter.cpp:35-63
    std::vector<std::string> out;
    for (int i = 1;; i++) {
        std::string p = base + "." + std::to_string(i) + ".las";
        if (access(p.c_str(), F_OK) != 0) break;
        out.push_back(p);
    }
    return out;
}

// .mas -> effective_start / effective_end (maximal.cpp:524-531, hinging.cpp:867-874)
inline bool read_mas(const std::string& path, int n_reads, std::vector<int32_t>& eff, std::vector<uint8_t>& seen) {
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return false;
    eff.assign((size_t)n_reads * 2, 0);
    seen.assign((size_t)n_reads, 0);
    int r, s, e;
    while (fscanf(f, "%d %d %d", &r, &s, &e) != EOF) {
        if (r < 0 || r >= n_reads) continue;
        eff[(size_t)r * 2] = s; eff[(size_t)r * 2 + 1] = e; seen[(size_t)r] = 1;
    }
    fclose(f);
    return true;
}

#define HH_CHECK(ctx, call)                                                                   \
    do {                                                                                      \
        int _rc = (call);                                                                     \
        if (_rc != HINGE_OK) {                                                                \
            console.error("%s failed (%d): %s", #call, _rc, hinge_last_error(ctx));           \
            return _rc == HINGE_E_UNDEFINED ? 1 : 2;                                          \
        }                                                                                     \
    } while (0)

}  // namespace hh
