// draft_assembly  ==  `hinge draft --db D --las L [--mlas] -x PREFIX -o OUT --config nominal.ini`
// Same flags, nominal.ini keys ([draft] min_cov / trim / edge_safe / tspace / step), input files (<PREFIX>.max, <PREFIX>.edges.list
// from `hinge draft-path`), output (<OUT>.fasta: one `>UnitigN` record per path of the file) and side effects (three files
// truncated) as src/consensus/draft.cpp:720-1162.  Host side - what is a walk over a few thousand reads in the reference too:
// which alignment belongs to which edge, the way points carried from read to read, lanes, ladders, the template choice by
// coverage, prefix / suffix / overhang / cuts, the text.  Base-level work runs behind the C ABI on the GPU, for ALL contigs of the
// file at once: hinge_draft_mappings (realignment between trace points -> the A-to-B maps of every edge) and hinge_draft_ladders
// (falcon's aligner + consensus for every ladder with more than one member).
#include "host_common.h"

#include <unordered_map>

using namespace hh;

namespace {

struct Rec {       // one .las record between two maximal reads: the LAlignment view (raw) and the LOverlap view (B flipped forward)
    int a, b, comp, ab, ae, bb, be, bbf, bef, alen, blen, tlen;
    const uint8_t* tr;
    int tbytes;
};
struct Edge { int a, sa, b, sb, w; };
struct Contig {
    std::string name;
    std::vector<Edge> edges;
    int cut_start = 0, cut_end = 0;
    bool one = false, two = false;
    // results
    int rc = -1;                     // draft_assembly_ctg's return value
    bool set = false;                // `contig` was assigned
    std::string text;
    std::string echo;                // its lines of the path file as main() echoes them while reading (printed in front of its name)
    std::string log;                 // what draft_assembly_ctg prints
    // multi-read path: per edge
    std::vector<int> selected, cur;  // record indices
    std::vector<int> map_id;         // index into the batch of hinge_draft_mappings
    struct Rung { int read, start, end; };
    struct Ladder { std::vector<Rung> rungs; int mx = 0; int64_t gpu = -1; };
    std::vector<Ladder> ladders;
    std::string prefix, suffix, overhang;
    int cut_end_eff = 0;
};

struct Bases {
    const ReadDB* db; const Mapped* bps;
    char at(int r, int p) const {
        const uint8_t b = bps->p[(size_t)db->boff[(size_t)r] + (size_t)(p >> 2)];
        return "acgt"[(b >> (6 - 2 * (p & 3))) & 3];
    }
    // bases [start, start + len) of the read in its strand frame, lower case (getRead: Load_Read(.., 1); reverse_complement of draft.cpp:91-99)
    std::string str(int r, int strand, int start, int len) const {
        const int rl = db->rlen[(size_t)r];
        std::string s((size_t)std::max(len, 0), 'a');
        for (int i = 0; i < len; i++) {
            const int p = start + i;
            if (!strand) s[(size_t)i] = at(r, p);
            else { const char c = at(r, rl - 1 - p); s[(size_t)i] = c == 'a' ? 't' : c == 'c' ? 'g' : c == 'g' ? 'c' : 'a'; }
        }
        return s;
    }
};

void logf(std::string& out, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    out += buf;
}

std::vector<std::string> split_blank(const std::string& s) {   // split(s, ' ') of draft.cpp:102-118
    std::vector<std::string> out;
    size_t at = 0;
    if (s.empty()) return out;
    while (true) {
        const size_t sp = s.find(' ', at);
        if (sp == std::string::npos) { out.push_back(s.substr(at)); break; }
        out.push_back(s.substr(at, sp - at));
        at = sp + 1;
        if (at == s.size()) break;       // (std::getline does not deliver an empty last item)
    }
    return out;
}

[[noreturn]] void die(const char* what, int code = 2) { fprintf(stderr, "draft_assembly: %s\n", what); fflush(nullptr); quit(code); }

}  // namespace

int main(int argc, char* argv[]) {
    CmdLine cmdp;
    cmdp.add_string("db", 'b', "db file name", false, "");
    cmdp.add_string("las", 'l', "las file name", false, "");
    cmdp.add_string("paf", 'p', "paf file name", false, "");
    cmdp.add_string("config", 'c', "configuration file name", false, "");
    cmdp.add_string("fasta", 'f', "fasta file name", false, "");
    cmdp.add_string("prefix", 'x', "(intermediate output) input file prefix", true, "");
    cmdp.add_string("out", 'o', "final output file name", true, "");
    cmdp.add_string("log", 'g', "log folder name", false, "log");
    cmdp.add_string("path", 0, "path file name", false, "path");
    cmdp.add_flag("debug", '\0', "debug mode");
    cmdp.add_flag("mlas", '\0', "multiple las files");
    cmdp.parse_check(argc, argv);
    PhaseTimer tm("draft");
    CtxInit gpu;
    gpu.start();
    const std::string name_db = cmdp.get("db"), name_las = cmdp.get("las"), name_config = cmdp.get("config");
    const std::string out = cmdp.get("prefix"), out_name = cmdp.get("out");
    // (the three std::ofstream draft.cpp:763-765 opens and never writes)
    for (const std::string& p : {out_name + ".deadends.txt", out + ".garbage.txt", out + ".contained.txt"}) { FILE* f = fopen(p.c_str(), "w"); if (f) fclose(f); }
    Log console;
    console.open(cmdp.get("log"));
    console.info("draft consensus");
    console.info("name of db: %s, name of .las file %s", name_db.c_str(), name_las.c_str());
    console.info("filter files prefix: %s", out.c_str());
    console.info("output prefix: %s", out_name.c_str());
    if (!cmdp.get("paf").empty() || !cmdp.get("fasta").empty()) die("FASTA + PAF input carries no trace points: `hinge draft` needs a DAZZ_DB and a .las (draft.cpp:213 realigns between trace points)", 1);

    ReadDB db;
    Mapped bps;
    if (name_db.empty() || db.open(name_db) != 0) { gpu.join(); die("Could not open database", 1); }
    if (!bps.open(db.dir + "/." + db.root + ".bps") && !db.rlen.empty()) { gpu.join(); die("cannot read the .bps file of the database", 1); }
    const int n_read = (int)db.rlen.size();
    console.info("# Reads: %d", n_read);
    Bases B{&db, &bps};
    std::vector<uint8_t> active((size_t)n_read, 0);
    int n_active = 0;
    {
        FILE* f = fopen((out + ".max").c_str(), "r");
        if (f) {
            char line[4096];
            while (fgets(line, sizeof line, f)) {
                const int r = atoi(line);
                if (r < 0 || r >= n_read) die("a read of the .max file lies outside the database");
                active[(size_t)r] = 1; n_active++;
            }
            fclose(f);
        }
    }
    console.info("Total number of active reads: %d/%d", n_active, n_read);

    // ---- the .las parts: records between two maximal reads (getOverlap / getAlignment with range = the active reads) ----------
    const bool mlas = cmdp.exist("mlas");
    std::vector<std::string> parts = mlas ? las_parts(name_las) : std::vector<std::string>{las_name(name_las, false)};
    std::vector<std::unique_ptr<Mapped>> maps;
    std::vector<Rec> recs;
    int64_t n_aln = 0;
    int tspace = 100;
    for (size_t part = 0; part < parts.size(); part++) {
        console.info("part:%d", (int)part);
        console.info("name of las %s", parts[part].c_str());
        maps.emplace_back(new Mapped());
        Mapped& las = *maps.back();
        if (!las.open(parts[part]) || las.n < 12) { gpu.join(); die("cannot read the .las file", 1); }
        const int64_t novl = rd<int64_t>(las.p);
        {   //  hinge_draft_mappings takes ONE trace spacing: parts that disagree would be realigned with the wrong segments
            const int ts = rd<int32_t>(las.p + 8);
            if (part > 0 && ts != tspace) { gpu.join(); die("the .las parts have different trace spacings", 1); }
            tspace = ts;
        }
        const int tbytes = tspace <= 125 ? 1 : 2;
        n_aln += novl;
        console.info("# Alignments: %lld", (long long)novl);
        size_t p = 12;
        for (int64_t j = 0; j < novl; j++) {
            if (p + 40 > las.n) break;
            const uint8_t* r = las.p + p;
            const int tlen = rd<int32_t>(r);
            const size_t tb = (size_t)std::max(tlen, 0) * (size_t)tbytes;
            if (tlen < 0 || p + 40 + tb > las.n) break;
            p += 40 + tb;
            const int a = rd<int32_t>(r + 28), b = rd<int32_t>(r + 32);
            if (a < 0 || a >= n_read || b < 0 || b >= n_read) die("an alignment names a read outside the database");
            if (!active[(size_t)a] || !active[(size_t)b]) continue;
            Rec q;
            q.a = a; q.b = b; q.comp = (int)(rd<uint32_t>(r + 24) & 1u);
            q.ab = rd<int32_t>(r + 8); q.bb = rd<int32_t>(r + 12); q.ae = rd<int32_t>(r + 16); q.be = rd<int32_t>(r + 20);
            q.alen = db.rlen[(size_t)a]; q.blen = db.rlen[(size_t)b];
            //  the coordinates index per-base tables below (coverage_of, substr): a corrupt or foreign .las stops here
            if (q.ab < 0 || q.ab > q.ae || q.ae > q.alen || q.bb < 0 || q.bb > q.be || q.be > q.blen) die("an alignment's coordinates lie outside its reads");
            if (q.comp) { q.bbf = q.blen - q.be; q.bef = q.blen - q.bb; } else { q.bbf = q.bb; q.bef = q.be; }
            q.tlen = tlen; q.tr = r + 40; q.tbytes = tbytes;
            recs.push_back(q);
        }
    }
    if (n_aln == 0) { console.error("No alignments!"); gpu.join(); return 1; }
    console.info("Input data finished");
    Config ini(name_config);
    if (ini.error < 0) { console.warn("Can't load %s", name_config.c_str()); gpu.join(); return 1; }
    const int EDGE_SAFE = (int)ini.get_int("draft", "edge_safe", -1);
    const int TSPACE = (int)ini.get_int("draft", "tspace", -1);
    tm.mark("ingest");

    std::vector<std::vector<int>> by_a((size_t)n_read);                     // idx3 / idx_aln: the pile-up of every read, file order
    for (size_t r = 0; r < recs.size(); r++) by_a[(size_t)recs[r].a].push_back((int)r);
    printf("add data\nadd data\n");

    // ---- <PREFIX>.edges.list: the lines std::getline delivers, the last one being the end-of-file marker -----------------------
    std::vector<std::string> lines;
    {
        Mapped el;
        if (el.open(out + ".edges.list")) {
            const std::string all((const char*)el.p, el.n);
            size_t at = 0;
            while (true) {
                const size_t nl = all.find('\n', at);
                if (nl == std::string::npos) { lines.push_back(all.substr(at)); break; }
                lines.push_back(all.substr(at, nl - at));
                at = nl + 1;
            }
        } else lines.push_back("");
    }
    for (const std::string& l : lines) {                                    // the first pass only echoes (draft.cpp:1057-1072)
        printf("%s\n", l.c_str());
        if (l.empty() || l[0] == '>') continue;
        const std::vector<std::string> tok = split_blank(l);
        if (tok.size() < 6) printf("Error! Wrong format.\n");
        if (tok.size() < 4) die("malformed line in the .edges.list file");
    }
    std::vector<Contig> ctgs;            // in the order the reference calls draft_assembly_ctg
    std::vector<std::string> order_log;  // what main() prints between those calls
    {
        Contig cur;
        std::string current_name;
        for (size_t li = 0; li < lines.size(); li++) {
            const std::string& l = lines[li];
            const bool eof = li + 1 == lines.size();
            if (!l.empty() && l[0] == '>') {
                if (!cur.edges.empty()) { cur.name = current_name; ctgs.push_back(cur); order_log.push_back(current_name + "\n"); }
                else order_log.push_back(current_name + "\n#");     // '#': no call follows
                cur = Contig();
                current_name = l;
                continue;
            }
            if (eof) { cur.name = current_name; ctgs.push_back(cur); order_log.push_back(current_name + "\n"); cur = Contig(); continue; }
            const std::vector<std::string> tok = split_blank(l);
            const size_t need = tok.empty() ? 6 : (tok[0] == "O" || tok[0] == "S" || tok[0] == "E" ? 7 : tok[0] == "D" ? 8 : 6);
            if (tok.size() < need) die("malformed line in the .edges.list file");
            Edge e;
            try {
                e.a = std::stoi(tok[1]); e.sa = std::stoi(tok[2]); e.b = std::stoi(tok[3]); e.sb = std::stoi(tok[4]);
                if (tok[0] == "O") { e.w = 0; cur.one = true; }
                else if (tok[0] == "D") { e.w = std::stoi(tok[5]); cur.two = true; }
                else e.w = std::stoi(tok[5]);
                if (tok[0] == "O") { cur.cut_start = std::stoi(tok[5]); cur.cut_end = std::stoi(tok[6]); }
                else if (tok[0] == "S") cur.cut_start = std::stoi(tok[6]);
                else if (tok[0] == "E") cur.cut_end = std::stoi(tok[6]);
                else if (tok[0] == "D") { cur.cut_start = std::stoi(tok[6]); cur.cut_end = std::stoi(tok[7]); }
            } catch (...) { die("malformed line in the .edges.list file"); }
            if (e.a < 0 || e.a >= n_read || e.b < 0 || e.b >= n_read) die("the .edges.list file names a read outside the database");
            cur.edges.push_back(e);
            cur.echo += (tok.size() < 6 ? std::string("Error! Wrong format.\n") : std::string()) + l + "\n";
        }
    }
    tm.mark("paths");

    // ---- phase 1: one- and two-read contigs on the host; for the others the alignment of every edge ---------------------------
    std::vector<hinge_cns_alignment> alns;
    std::vector<uint16_t> trace;
    std::vector<int64_t> map_off{0};
    std::unordered_map<int, int> map_of_rec;      // record -> its slot in the batch
    auto cut_into = [&](Contig& c, const std::string& draft, int cs, int ce_len) {     // contig = draft.substr(cs, ce_len) under the reference's test
        c.text = draft.substr((size_t)cs, (size_t)ce_len);
        c.set = true;
    };
    for (Contig& c : ctgs) {
        logf(c.log, "list size:%lu\n", (unsigned long)c.edges.size());
        if (c.edges.empty()) { c.rc = -1; continue; }
        const Edge& e0 = c.edges[0];
        if (c.one) {
            const std::string draft = B.str(e0.a, e0.sa, 0, db.rlen[(size_t)e0.a]);
            logf(c.log, "%d %d %d\n", c.cut_start, c.cut_end, db.rlen[(size_t)e0.a]);
            if ((size_t)c.cut_start <= draft.size() && (size_t)c.cut_end <= draft.size()) cut_into(c, draft, c.cut_start, c.cut_end - c.cut_start);
            c.rc = 1;
            continue;
        }
        for (const Edge& e : c.edges) {
            int sel = -1, cur = -1;
            for (int r : by_a[(size_t)e.a]) {
                const Rec& q = recs[(size_t)r];
                if (q.b != e.b || q.ae - q.ab + q.be - q.bb != e.w) continue;
                if (sel < 0) sel = r;                 // `selected`: the first that fits (draft.cpp:166-176)
                cur = r;                              // `currentaln`: the last (draft.cpp:266-272)
            }
            if (sel >= 0) c.selected.push_back(sel);
            c.cur.push_back(cur);
        }
        logf(c.log, "selected:%lu\n", (unsigned long)c.selected.size());
        if (c.selected.size() != c.edges.size()) die("an edge of the path file has no alignment of its length in the .las (the reference indexes its lists out of step from here)");
        if (c.two) {
            const Rec& s0 = recs[(size_t)c.selected[0]];
            std::string draft = B.str(e0.a, e0.sa, 0, db.rlen[(size_t)e0.a]);
            const std::string readB = B.str(e0.b, e0.sb, 0, db.rlen[(size_t)e0.b]);
            logf(c.log, "alen blen aend bstart%d %d %d %d\n", db.rlen[(size_t)e0.a], db.rlen[(size_t)e0.b], s0.ae, s0.bb);
            draft = draft.substr(0, (size_t)s0.ae) + readB.substr((size_t)s0.bb);
            logf(c.log, "%d %d %d\n", c.cut_start, c.cut_end, db.rlen[(size_t)e0.a]);
            if ((size_t)c.cut_start <= draft.size() && (size_t)c.cut_end <= draft.size()) cut_into(c, draft, c.cut_start, c.cut_end - c.cut_start);
            c.rc = 2;
            continue;
        }
        for (size_t i = 0; i < c.edges.size(); i++) {
            if (c.cur[i] < 0) die("an edge of the path file has no overlap of its length (the reference exits here)", 1);
            const int r = c.selected[i];
            auto it = map_of_rec.find(r);
            if (it == map_of_rec.end()) {
                const Rec& q = recs[(size_t)r];
                hinge_cns_alignment a;
                a.aread = q.a; a.bread = q.b; a.comp = q.comp; a.abpos = q.ab; a.aepos = q.ae; a.bbpos = q.bb; a.bepos = q.be;
                a.tlen = q.tlen; a.trace_off = (int64_t)trace.size();
                if (q.tbytes == 1) for (int k = 0; k < q.tlen; k++) trace.push_back(q.tr[k]);
                else for (int k = 0; k < q.tlen; k++) trace.push_back((uint16_t)(q.tr[2 * k] | (q.tr[2 * k + 1] << 8)));
                it = map_of_rec.emplace(r, (int)alns.size()).first;
                alns.push_back(a);
                map_off.push_back(map_off.back() + (q.ae - q.ab));
            }
            c.map_id.push_back(it->second);
        }
    }
    tm.mark("edges -> alignments");

    // ---- the GPU, part 1: every edge's A-to-B map ---------------------------------------------------------------------------------
    if (gpu.join() != HINGE_OK) { fprintf(stderr, "draft_assembly: no usable GPU (%s)\n", gpu.ctx ? hinge_last_error(gpu.ctx) : "hinge_ctx_create failed"); quit(2); }
    hinge_ctx* ctx = gpu.ctx;
    tm.mark("hip init");
    auto gdie = [&](const char* what) { fprintf(stderr, "draft_assembly: %s: %s\n", what, hinge_last_error(ctx)); fflush(nullptr); quit(2); };
    for (int which = 0; which < 2; which++)
        if (hinge_consensus_set_db(ctx, which, n_read, db.rlen.data(), db.boff.data(), bps.p, (int64_t)bps.n) != HINGE_OK) gdie("read DB");
    tm.mark("H2D bases");
    std::vector<uint32_t> mapping((size_t)std::max<int64_t>(map_off.back(), 1));
    if (hinge_draft_mappings(ctx, (int64_t)alns.size(), alns.data(), trace.data(), (int64_t)trace.size(), tspace, map_off.data(), mapping.data()) != HINGE_OK) gdie("realignment");
    tm.mark("realign -> maps");

    // ---- phase 2: way points, lanes, ladders of every multi-read contig (draft.cpp:258-556) ------------------------------------
    std::vector<int64_t> rung_off{0}, slot_off{0};
    std::vector<hinge_draft_rung> rungs;
    std::vector<int32_t> tmpl;
    std::vector<std::vector<int>> cov_cache((size_t)n_read);     // getCoverage of a read's pile-up, forward frame, made once
    auto coverage_of = [&](int a) -> const std::vector<int>& {
        std::vector<int>& cov = cov_cache[(size_t)a];
        if (cov.empty() && !by_a[(size_t)a].empty()) {
            std::vector<int> diff((size_t)db.rlen[(size_t)a] + 1, 0);
            for (int r : by_a[(size_t)a]) { diff[(size_t)recs[(size_t)r].ab]++; diff[(size_t)recs[(size_t)r].ae]--; }
            cov.resize((size_t)db.rlen[(size_t)a]);
            int run = 0;
            for (size_t p = 0; p < cov.size(); p++) { run += diff[p]; cov[p] = run; }
        }
        return cov;
    };
    for (Contig& c : ctgs) {
        if (c.rc != -1 || c.edges.empty()) continue;
        const size_t n = c.edges.size();
        struct BEdge { int as, ae, bs, be, alen, blen; };
        std::vector<BEdge> be(n);
        std::vector<std::vector<int>> maps(n);
        int len_overhang = 0;
        for (size_t i = 0; i < n; i++) {
            const Edge& e = c.edges[i];
            const Rec& q = recs[(size_t)c.cur[i]];
            BEdge& b = be[i];
            b.alen = q.alen; b.blen = q.blen;
            if (e.sa == 0) { b.as = q.ab; b.ae = q.ae; } else { b.as = b.alen - q.ae; b.ae = b.alen - q.ab; }
            if (e.sb == 0) { b.bs = q.bbf; b.be = q.bef; } else { b.bs = b.blen - q.bef; b.be = b.blen - q.bbf; }
            len_overhang = b.blen - b.be - (b.alen - b.ae);
            // get_mapping of the tags in the edge's strand frame: forward as the GPU delivers it, or of the reverse-complemented
            // rows: A base L-1-k then has the B bases of the columns BEHIND base k's column in front of it
            const Rec& s = recs[(size_t)c.selected[i]];
            const uint32_t* m = mapping.data() + map_off[(size_t)c.map_id[i]];
            const int L = s.ae - s.ab, nb = s.be - s.bb;
            maps[i].resize((size_t)L);
            if (e.sa == 0) for (int k = 0; k < L; k++) maps[i][(size_t)k] = (int)(m[k] & 0x7fffffffu);
            else for (int k = 0; k < L; k++) maps[i][(size_t)(L - 1 - k)] = nb - ((int)(m[k] & 0x7fffffffu) + ((m[k] >> 31) ? 0 : 1));
        }
        {
            const Edge& el = c.edges.back();
            const int bl = db.rlen[(size_t)el.b];
            if (len_overhang > 0 && len_overhang < bl) c.overhang = B.str(el.b, el.sb, bl - len_overhang, len_overhang);
        }
        logf(c.log, "%lu %lu %lu %lu %lu %lu %lu %lu\n", (unsigned long)n, (unsigned long)n, (unsigned long)n, (unsigned long)n, (unsigned long)n, (unsigned long)n,
             (unsigned long)n, (unsigned long)n);
        std::vector<std::vector<std::pair<int, int>>> lanes;
        std::vector<std::vector<int>> trace_pts(n);
        {
            int start_read = 0, space = 1, offset = 0, rmax = -1;
            const int nb = (int)n;
            while (start_read < nb - 1) {
                int cur = start_read;
                while (be[(size_t)start_read].as + space * TSPACE + offset < be[(size_t)start_read].ae - EDGE_SAFE) {
                    int way = be[(size_t)start_read].as + TSPACE * space + offset;
                    std::vector<std::pair<int, int>> lane;
                    while (way > be[(size_t)cur].as && way < be[(size_t)cur].ae) {
                        trace_pts[(size_t)cur].push_back(way);
                        lane.push_back({cur, way});
                        rmax = std::max(rmax, cur);
                        const int at = way - be[(size_t)cur].as;
                        if (at >= (int)maps[(size_t)cur].size()) die("a way point outside its edge's alignment");
                        way = maps[(size_t)cur][(size_t)at] + be[(size_t)cur].bs;
                        if (++cur >= nb) break;
                    }
                    if (cur < nb && way < be[(size_t)cur].alen) { lane.push_back({cur, way}); rmax = std::max(rmax, cur); }
                    if (cur >= rmax) lanes.push_back(lane);
                    space++;
                    cur = start_read;
                }
                start_read++;
                space = 1;
                offset = trace_pts[(size_t)start_read].empty() ? 0 : trace_pts[(size_t)start_read].back() - be[(size_t)start_read].as;
            }
        }
        for (size_t i = 0; i < n; i++) {
            logf(c.log, "Read %d:", (int)i);
            for (int w : trace_pts[i]) logf(c.log, "%d ", w);
            c.log += "\n";
        }
        for (size_t i = 0; i < lanes.size(); i++) {
            logf(c.log, "Lane %d\n", (int)i);
            for (auto& p : lanes[i]) logf(c.log, "[%d %d] ", p.first, p.second);
            c.log += "\n";
        }
        logf(c.log, "In total %lu lanes\n", (unsigned long)lanes.size());
        if (lanes.empty() || lanes[0].empty() || lanes.back().empty()) die("a path without a lane (the reference reads lanes[0] of an empty list)");
        const int first_start = lanes[0][0].second, last_end = lanes.back().back().second;
        const Edge& e0 = c.edges[0];
        const Edge& el = c.edges.back();
        logf(c.log, "first %d last %d\n", first_start, last_end);
        logf(c.log, "len %d %d\n", db.rlen[(size_t)e0.a], db.rlen[(size_t)el.b]);
        if (first_start < 0 || last_end < 0 || first_start > db.rlen[(size_t)e0.a] || last_end > db.rlen[(size_t)el.a]) die("a lane ends outside its read (the reference's assert)");
        c.prefix = B.str(e0.a, 0, 0, first_start);                                           // FORWARD bases whatever the strand (draft.cpp:524-525)
        c.suffix = B.str(el.a, 0, last_end, db.rlen[(size_t)el.a] - last_end);
        logf(c.log, "last read %d length %d, cut %d\n", el.b, db.rlen[(size_t)el.b], c.cut_end);
        c.cut_end_eff = db.rlen[(size_t)el.b] - c.cut_end;
        for (size_t i = 0; i + 1 < lanes.size(); i++) {
            const auto& l1 = lanes[i]; const auto& l2 = lanes[i + 1];
            Contig::Ladder ld;
            size_t pos = 0;
            for (size_t j = 0; j < l2.size(); j++) {
                while (l1[pos].first != l2[j].first && pos < l1.size() - 1) pos++;
                if (l1[pos].first == l2[j].first) ld.rungs.push_back({l2[j].first, l1[pos].second, l2[j].second});
            }
            for (const auto& g : ld.rungs)
                if (g.start < 0 || g.end < g.start || g.end > db.rlen[(size_t)c.edges[(size_t)g.read].a]) die("a ladder rung outside its read (the reference overruns a buffer here)");
            if (ld.rungs.size() > 1) {
                int mx = 0, maxcoverage = 0;
                for (size_t j = 0; j < ld.rungs.size(); j++) {
                    int mincoverage = 10000;
                    const std::vector<int>& cov = coverage_of(c.edges[(size_t)ld.rungs[j].read].a);
                    for (int p = ld.rungs[j].start; p < ld.rungs[j].end; p++) mincoverage = std::min(mincoverage, cov[(size_t)p]);
                    if (mincoverage > maxcoverage) { maxcoverage = mincoverage; mx = (int)j; }
                }
                ld.mx = mx;
                ld.gpu = (int64_t)tmpl.size();
                for (const auto& g : ld.rungs) rungs.push_back(hinge_draft_rung{c.edges[(size_t)g.read].a, c.edges[(size_t)g.read].sa, g.start, g.end});
                rung_off.push_back((int64_t)rungs.size());
                tmpl.push_back(mx);
                slot_off.push_back(slot_off.back() + 2ll * (ld.rungs[(size_t)mx].end - ld.rungs[(size_t)mx].start + 1));
            }
            c.ladders.push_back(std::move(ld));
        }
        c.rc = 0;
    }
    tm.mark("lanes + ladders");

    // ---- the GPU, part 2: every ladder's consensus ------------------------------------------------------------------------------
    std::vector<char> cns((size_t)std::max<int64_t>(slot_off.back(), 1));
    std::vector<int32_t> cns_len(std::max<size_t>(tmpl.size(), 1));
    if (!tmpl.empty() && hinge_draft_ladders(ctx, (int64_t)tmpl.size(), rung_off.data(), rungs.data(), tmpl.data(), 150, slot_off.data(), cns.data(), cns_len.data()) != HINGE_OK)
        gdie("ladders");
    tm.mark("ladder consensus");

    // ---- the text, in the reference's order; `contig` is one variable for the whole run ----------------------------------------------
    FILE* fa = fopen((out_name + ".fasta").c_str(), "w");
    if (!fa) die("cannot write the FASTA file", 1);
    std::string contig;
    size_t next = 0;
    for (const std::string& ol : order_log) {
        const bool call = ol.empty() || ol.back() != '#';
        if (!call) { fputs(ol.substr(0, ol.size() - 1).c_str(), stdout); continue; }
        Contig& c = ctgs[next++];
        fputs(c.echo.c_str(), stdout);       // (echoed while the lines were read: in front of the name the next '>' line prints)
        fputs(ol.c_str(), stdout);
        if (c.rc == 0) {
            std::string body;
            for (const auto& ld : c.ladders) {
                if (ld.rungs.empty()) { c.log += "low coverage!\n"; continue; }
                if (ld.gpu >= 0) body.append(cns.data() + slot_off[(size_t)ld.gpu], (size_t)cns_len[(size_t)ld.gpu]);
                else body += B.str(c.edges[(size_t)ld.rungs[0].read].a, c.edges[(size_t)ld.rungs[0].read].sa, ld.rungs[0].start, ld.rungs[0].end - ld.rungs[0].start);
            }
            logf(c.log, "0\n%lu\n", (unsigned long)body.size());
            std::string whole = c.prefix + body + c.suffix + c.overhang;
            logf(c.log, "ctg size:%lucut_start:%dcut_end:%d\n", (unsigned long)whole.size(), c.cut_start, c.cut_end_eff);
            if ((size_t)c.cut_start <= whole.size() && (size_t)c.cut_end_eff <= whole.size())
                whole = whole.substr((size_t)c.cut_start, whole.size() - (size_t)c.cut_end_eff - (size_t)c.cut_start);
            c.text.swap(whole);
            c.set = true;
        }
        fputs(c.log.c_str(), stdout);
        if (c.set) contig = c.text;
        // (the '>' branch prints only when the list is not empty - it never is here -, the end-of-file branch always)
        fprintf(fa, "%s\n%s\n", c.name.c_str(), contig.c_str());
    }
    fclose(fa);
    tm.mark("text");
    return finish(ctx, tm, 0);
}
