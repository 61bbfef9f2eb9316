// hinging  ==  `hinge layout --db DB --las LAS[.las] [--mlas] -x PREFIX -o OUT --config nominal.ini`
// Same flags, inputs (.mas .max .repeat.txt .hinges.txt), outputs (.edges.hinges .edges.hinges2 .hinge.list
// .deadends.txt .hgraph .killed.hinges .garbage.txt .edges.greedy .edges.1 .edges.2 .edges.skipped + the
// debug dumps in the cwd) and exit codes as src/layout/hinging.cpp.
//
// GPU: ProcessAlignment of the best one/two overlaps of every maximal x maximal pair (k_trim_classify) and
// GetMatchingPosition of every hinge through every match (k_matching_position).  Host: the (A, B) walk in
// unordered_map order, the std::sort by weight, the hinge graph bookkeeping and the text writers - all on
// the ~10^5 matches that survive, i.e. a few MB.
#include <fstream>
#include <set>
#include <sstream>
#include "pairs.h"

using namespace hh;

namespace {

struct Match {
    int a, b, comp;
    int ab, ae, bb, be;          // raw coordinates (B on the forward strand)
    int a_rs, a_re, b_rs, b_re;  // effective read bounds used for the trim
    int64_t k;                   // index in the part's pile-up arrays
    int part;
    Classified c;
};

struct Hinge {
    int pos, type;
    bool active;
};

void print_overlap(FILE* f, const Match& m) {   // PrintOverlapToFile, hinging.cpp:188-248
    const int t = m.c.type;
    const int hinged = (t == MT_FORWARD || t == MT_BACKWARD) ? -1 : 1;
    if (t == MT_FORWARD_INTERNAL || t == MT_FORWARD)
        fprintf(f, "%d %d %d %d %d %d [%d %d] [%d %d] [%d %d] [%d %d] [%d %d] [%d %d]\n", m.a, m.b, m.c.length, 0, m.comp, hinged, m.c.eff_ab,
                m.c.eff_ae, m.c.eff_bb, m.c.eff_be, m.a_rs, m.a_re, m.b_rs, m.b_re, m.ab, m.ae, m.bb, m.be);
    else if (t == MT_BACKWARD_INTERNAL || t == MT_BACKWARD)
        fprintf(f, "%d %d %d %d %d %d [%d %d] [%d %d] [%d %d] [%d %d] [%d %d] [%d %d]\n", m.b, m.a, m.c.length, m.comp, 0, hinged, m.c.eff_bb,
                m.c.eff_be, m.c.eff_ab, m.c.eff_ae, m.b_rs, m.b_re, m.a_rs, m.a_re, m.ab, m.ae, m.bb, m.be);
}

void print_overlap2(FILE* f, const Match& m, int hinge_pos) {   // PrintOverlapToFile2, hinging.cpp:253-344
    const int t = m.c.type;
    if (t == MT_FORWARD || t == MT_FORWARD_INTERNAL)
        fprintf(f, "%d %d %d %d %d %d %d [%d %d] [%d %d] [%d %d] [%d %d]\n", m.a, m.b, m.c.length, 0, m.comp, t == MT_FORWARD ? 0 : 1,
                t == MT_FORWARD ? -1 : hinge_pos, m.c.eff_ab, m.c.eff_ae, m.c.eff_bb, m.c.eff_be, m.a_rs, m.a_re, m.b_rs, m.b_re);
    else if (t == MT_BACKWARD || t == MT_BACKWARD_INTERNAL)
        fprintf(f, "%d %d %d %d %d %d %d [%d %d] [%d %d] [%d %d] [%d %d]\n", m.b, m.a, m.c.length, m.comp, 0, t == MT_BACKWARD ? 0 : -1,
                t == MT_BACKWARD ? -1 : hinge_pos, m.c.eff_bb, m.c.eff_be, m.c.eff_ab, m.c.eff_ae, m.b_rs, m.b_re, m.a_rs, m.a_re);
}

void print_dump(FILE* f, const Match& m) {   // the 13-field debug line, hinging.cpp:1080-1090 etc.
    fprintf(f, "%d %d %d %d %d [%d %d] [%d %d] [%d %d] [%d %d] \n", m.a, m.b, m.c.length, m.comp, m.c.type, m.c.eff_ab, m.c.eff_ae, m.c.eff_bb,
            m.c.eff_be, m.a_rs, m.a_re, m.b_rs, m.b_re);
}

void print_g(FILE* f, const char* fmt, int x, int y, const Match& m) {
    fprintf(f, fmt, x, y, m.c.length, m.c.eff_ab, m.c.eff_ae, m.c.eff_bb, m.c.eff_be, m.a_rs, m.a_re, m.b_rs, m.b_re);
}

// "<id> <v1> <v2> <v1> <v2> ... \n" lines; pairs with v1 != 0 and v2 != 0 (hinging.cpp:887-936)
void parse_pairs(const std::string& path, std::map<int, std::vector<std::pair<int, int>>>& m, std::vector<int>* order) {
    std::ifstream f(path);
    std::string line;
    while (std::getline(f, line)) {
        std::stringstream ss;
        ss << line << "\n";
        int num = 0;
        ss >> num;
        m[num] = std::vector<std::pair<int, int>>();
        if (order) order->push_back(num);
        while (!ss.eof()) {
            int r1 = 0, r2 = 0;
            ss >> r1 >> r2;
            if (r1 != 0 && r2 != 0) m[num].push_back(std::make_pair(r1, r2));
        }
    }
}

// component sizes are all that reach the output (hinging.cpp:1644-1675): union-find instead of Boost.Graph
std::vector<int> components(int n, const std::vector<std::pair<int, int>>& edges) {
    std::vector<int> parent((size_t)n);
    for (int i = 0; i < n; i++) parent[(size_t)i] = i;
    auto find = [&](int x) { while (parent[(size_t)x] != x) { parent[(size_t)x] = parent[(size_t)parent[(size_t)x]]; x = parent[(size_t)x]; } return x; };
    for (auto& e : edges) { const int a = find(e.first), b = find(e.second); if (a != b) parent[(size_t)a] = b; }
    std::vector<int> comp((size_t)n);
    for (int i = 0; i < n; i++) comp[(size_t)i] = find(i);
    return comp;
}

}  // namespace

// The overlaps of one .las part that layout sends to the GPU, packed: SoA rows in selection order (= A read order)
// and their trace bytes back to back.
struct PackedPart {
    int r_begin = 0, r_end = -1, tbytes = 1;
    bool is_paf = false;
    std::vector<int64_t> row_ptr;
    std::vector<int32_t> a_span, b_span, tlen;
    std::vector<uint32_t> b_flag;
    std::vector<int64_t> trace_off;
    std::vector<uint8_t> trace;
    void build(const LasPart& las, const std::vector<int64_t>& sel, const std::vector<int32_t>& a_of, int n_read) {
        r_begin = las.r_begin; r_end = las.r_end; tbytes = las.tbytes; is_paf = las.is_paf;
        const size_t n = sel.size();
        row_ptr.assign((size_t)n_read + 1, 0);
        for (size_t t = 0; t < n; t++) row_ptr[(size_t)a_of[t] + 1]++;
        for (int i = 0; i < n_read; i++) row_ptr[(size_t)i + 1] += row_ptr[(size_t)i];
        a_span.resize(2 * n); b_span.resize(2 * n); tlen.resize(n); b_flag.resize(n); trace_off.resize(n);
        int64_t bytes = 0;
        for (size_t t = 0; t < n; t++) {
            const size_t k = (size_t)sel[t];
            a_span[2 * t] = las.a_span[2 * k]; a_span[2 * t + 1] = las.a_span[2 * k + 1];
            b_span[2 * t] = las.b_span[2 * k]; b_span[2 * t + 1] = las.b_span[2 * k + 1];
            b_flag[t] = las.b_flag[k];
            tlen[t] = las.tlen[k];
            trace_off[t] = bytes;
            bytes += (int64_t)las.tlen[k] * las.tbytes;
        }
        trace.resize((size_t)std::max<int64_t>(bytes, 1));
        for (size_t t = 0; t < n; t++) {
            const size_t k = (size_t)sel[t];
            if (las.tlen[k] > 0) memcpy(trace.data() + trace_off[t], las.file.p + las.trace_off[k], (size_t)las.tlen[k] * (size_t)las.tbytes);
        }
    }
    int upload(hinge_ctx* ctx) const {
        // (a few thousand selected overlaps, no coverage pass runs on them: no span copy; the facts only gate filter kernels)
        int rc = hinge_set_pileups_packed(ctx, r_begin, r_end, (int64_t)b_flag.size(), row_ptr.data(), a_span.data(), b_span.data(), b_flag.data(), nullptr, 0x7fffffffu, 0, 0);
        if (rc != HINGE_OK) return rc;
        if ((rc = hinge_set_trim(ctx, is_paf ? 0 : 1)) != HINGE_OK) return rc;   // ProcessAlignment(trim = false) without trace points
        return hinge_set_traces(ctx, trace.data(), (int64_t)trace.size(), trace_off.data(), tlen.data(), tbytes, 0);
    }
};

#ifndef HINGE_STAGE_MAIN
#define HINGE_STAGE_MAIN main
#endif
int HINGE_STAGE_MAIN(int argc, char* argv[]) {
    CmdLine cmdp;
    cmdp.add_string("db", 'b', "db file name", false, "");
    cmdp.add_string("las", 'l', "las file name", false, "");
    cmdp.add_string("paf", 'p', "paf file name", false, "");
    cmdp.add_string("config", 'c', "configuration file name", false, "");
    cmdp.add_string("fasta", 'f', "fasta file name", false, "");
    cmdp.add_string("prefix", 'x', "(intermediate output) input file prefix", true, "");
    cmdp.add_string("out", 'o', "final output file name", true, "");
    cmdp.add_string("log", 'g', "log folder name", false, "log");
    cmdp.add_flag("debug", '\0', "debug mode");
    cmdp.add_flag("mlas", '\0', "multiple las files");
    cmdp.parse_check(argc, argv);

    Log console;
    console.open(cmdp.get("log"));
    PhaseTimer tm("layout");
    CtxInit gpu;
    gpu.start();
    console.info("Hinging layout");
    const std::string name_db = cmdp.get("db"), name_las_base = cmdp.get("las"), name_paf = cmdp.get("paf"), name_fasta = cmdp.get("fasta");
    const std::string name_config = cmdp.get("config"), out = cmdp.get("prefix"), out_name = cmdp.get("out");
    const bool mlas = cmdp.exist("mlas");
    FILE* deadend_out = fopen((out_name + ".deadends.txt").c_str(), "w");
    FILE* garbage_out = fopen((out + ".garbage.txt").c_str(), "w");
    const bool db_and_las = !name_db.empty() && !name_las_base.empty(), db_or_las = !name_db.empty() || !name_las_base.empty();
    const bool fa_and_paf = !name_fasta.empty() && !name_paf.empty(), fa_or_paf = !name_fasta.empty() || !name_paf.empty();
    if (db_or_las && fa_or_paf) { console.error("Pass in either a db and a las or a fasta and a paf"); return 1; }
    if (!fa_and_paf && !db_and_las) { console.error("Pass in at least one of the following two combinations: a db and a las or a fasta and a paf"); return 1; }
    if (mlas && !db_and_las) { console.error("--mlas works only with db and las"); return 1; }
    if (!deadend_out || !garbage_out) { console.error("cannot open output files"); return 2; }

    ReadDB db;
    if (fa_and_paf) {
        if (read_fasta_lengths(name_fasta, db.rlen) != 0) { fprintf(stderr, "hinging: cannot read %s\n", name_fasta.c_str()); quit(1); }
    } else if (db.open(name_db) != 0) { fprintf(stderr, "hinging: Could not open database %s\n", name_db.c_str()); quit(1); }
    const int n_read = (int)db.rlen.size();
    console.info("# Reads: %d", n_read);

    Config ini(name_config);
    if (ini.error < 0) { console.warn("Can't load %s", name_config.c_str()); return 1; }
    const int LENGTH_THRESHOLD = (int)ini.get_int("filter", "length_threshold", -1);
    const int ALN_THRESHOLD = (int)ini.get_int("filter", "aln_threshold", -1);
    const int THETA = (int)ini.get_int("filter", "theta", -1);
    const int THETA2 = (int)ini.get_int("filter", "theta2", 0);
    const int HINGE_SLACK = (int)ini.get_int("layout", "hinge_slack", 1000);
    const int HINGE_TOLERANCE = (int)ini.get_int("layout", "hinge_tolerance", 150);
    const int KILL_HINGE_OVERLAP_ALLOWANCE = (int)ini.get_int("layout", "kill_hinge_overlap", 300);
    const int KILL_HINGE_INTERNAL_ALLOWANCE = (int)ini.get_int("layout", "kill_hinge_internal", 40);
    const int MATCHING_HINGE_SLACK = (int)ini.get_int("layout", "matching_hinge_slack", 200);
    const int NUM_EVENTS_TELOMERE = (int)ini.get_int("layout", "num_events_telomere", 7);
    const int MIN_CONNECTED_COMPONENT_SIZE = (int)ini.get_int("layout", "min_connected_component_size", 8);
    const bool USE_TWO_MATCHES = ((int)ini.get_int("layout", "use_two_matches", 1)) != 0;
    const bool KEEP_ONLY_MAX = ((int)ini.get_int("layout", "keep_only_matches_between_maximal_reads", 1)) != 0;
    const bool delete_telomere = ((int)ini.get_int("layout", "del_telomeres", 0)) != 0;   // (sic) layout reads del_telomeres
    console.info("del_telomeres = %d", (int)delete_telomere);

    std::vector<int32_t> eff;
    std::vector<uint8_t> seen;
    if (!read_mas(out + ".mas", n_read, eff, seen)) { console.error("cannot open %s.mas (run hinge filter first)", out.c_str()); return 2; }
    int n_missing_mas = 0;
    for (int i = 0; i < n_read; i++)
        if (!seen[(size_t)i]) {
            // the reference reads uninitialised effective_start / effective_end here: zeroes on a fresh heap, i.e. the read is
            // inactive (0 < LENGTH_THRESHOLD).  Same here ((0, 0) is what read_mas left); HINGE_STRICT_MAS=1 refuses instead.
            if (getenv("HINGE_STRICT_MAS")) { console.error("read %d has no line in %s.mas: the reference reads uninitialised memory here", i, out.c_str()); return 2; }
            n_missing_mas++;
        }
    if (n_missing_mas) console.warn("%d reads have no line in %s.mas (outside the .las' A range): treated as inactive, mask (0, 0)", n_missing_mas, out.c_str());
    std::vector<uint8_t> active((size_t)n_read, 1);

    std::map<int, std::vector<std::pair<int, int>>> marked_repeats, marked_hinges;
    {
        std::vector<int> order;
        parse_pairs(out + ".repeat.txt", marked_repeats, &order);
        for (int num : order)
            if (delete_telomere && (int)marked_repeats[num].size() > NUM_EVENTS_TELOMERE && num >= 0 && num < n_read) active[(size_t)num] = 0;
        parse_pairs(out + ".hinges.txt", marked_hinges, nullptr);
    }
    for (int i = 0; i < n_read; i++)
        if (eff[(size_t)i * 2 + 1] - eff[(size_t)i * 2] < LENGTH_THRESHOLD) { active[(size_t)i] = 0; fprintf(garbage_out, "%d\n", i); }
    fclose(garbage_out);

    const std::string name_las = las_name(name_las_base, mlas);
    std::vector<std::string> las_list;
    if (fa_and_paf) las_list.push_back(name_paf);
    else if (mlas) las_list = las_parts(name_las); else las_list.push_back(name_las);
    // ---- ranks: one host thread + one context per visible GPU for a --mlas run (HINGE_RANKS overrides) --------------------------
    // The front half of a part - ingest, the (A, B) grouping of the active x active pairs, packing, trim + classify on the GPU -
    // runs for the parts of a wave side by side.  The reference's loop carries one thing from part to part: a read found
    // contained ("Should not happen", hinging.cpp:590-600) is inactive for the parts after it.  The front halves therefore work on
    // the activity at the start of the wave and are consumed in part order; should a part deactivate a read, the front halves of
    // the wave's later parts are simply run again (on the then current activity), one after the other.
    const int n_ranks = rank_count(las_list.size(), fa_and_paf);
    PartLoader loader;
    loader.paf = fa_and_paf;
    loader.single = las_list.size() == 1;
    if (!las_list.empty() && n_ranks == 1) loader.preload(las_list[0], db.rlen);
    tm.mark("setup + las ingest (part 1) || HIP init");
    if (gpu.join() != HINGE_OK) { console.error("no usable MI355X / HIP device: this build has no CPU path"); return 2; }
    std::vector<hinge_ctx*> ctxs((size_t)n_ranks, nullptr);
    ctxs[0] = gpu.ctx;
    if (n_ranks > 1) {
        const int ndev = std::max(1, hinge_device_count());
        for (int r = 1; r < n_ranks; r++)
            if (hinge_ctx_create(r % ndev, &ctxs[(size_t)r]) != HINGE_OK) { console.error("cannot create a context on device %d", r % ndev); return 2; }
    }
    hinge_ctx* ctx = ctxs[0];
    for (int r = 0; r < n_ranks; r++) {
        HH_CHECK(ctxs[(size_t)r], hinge_set_reads(ctxs[(size_t)r], n_read, db.rlen.data(), nullptr));
        HH_CHECK(ctxs[(size_t)r], hinge_set_eff_reads(ctxs[(size_t)r], eff.data()));
    }
    tm.mark("setup + ctx_create + set_reads");

    // ---- GetAlignment, hinging.cpp:347-610 ---------------------------------------------------------------
    {
        std::vector<uint8_t> maximal((size_t)n_read, 0);
        std::ifstream mf(out + ".max");
        std::string line;
        while (std::getline(mf, line)) { const int r = atoi(line.c_str()); if (r >= 0 && r < n_read) maximal[(size_t)r] = 1; }
        for (int i = 0; i < n_read; i++) active[(size_t)i] = active[(size_t)i] && maximal[(size_t)i];
    }
    std::vector<std::vector<Match>> matches_forward((size_t)n_read), matches_backward((size_t)n_read);
    // the ranks' classified overlaps reach the one sequential consume pass (hinging.cpp:917-936: matches_forward / _backward in read
    // order, reads dropped on the way) as ONE all-gather over RCCL per wave; HINGE_HOST_EXCHANGE=1 / shared devices: host data
    RowGather gather_rows;
    gather_rows.init(ctxs, "classified matches", console);
    std::vector<LasPart*> parts(las_list.size(), nullptr);
    std::vector<PackedPart*> packed(las_list.size(), nullptr);   // the selected overlaps of every part (GetMatchingPosition needs their traces again)
    int resident_part = -1;            // which packed part the GPU context `ctx` currently holds
    struct Front {   // the front half of one part
        int code = 0;
        std::string error;
        std::vector<std::vector<PairPick>> picks;
        std::vector<Classified> cls;
        size_t n_sel = 0;
    };
#define PART_FAIL(o, c, ...)                                            \
    do {                                                                 \
        char _b[512];                                                    \
        snprintf(_b, sizeof(_b), __VA_ARGS__);                           \
        (o).code = (c); (o).error = _b;                                  \
        return;                                                          \
    } while (0)
    auto front_half = [&](hinge_ctx* cx, size_t part, Front& f, bool timed) {
        f = Front();
        if (!parts[part]) {
            int lrc = 0;
            parts[part] = loader.take(part, las_list[part], db.rlen, lrc);
            if (lrc == -2) PART_FAIL(f, 2, "%s is not sorted by A read", las_list[part].c_str());
            if (lrc == -3) PART_FAIL(f, 1, "%s: a read name without \"/id/\" or an id outside the FASTA (the reference crashes here)", las_list[part].c_str());
            if (lrc != 0) PART_FAIL(f, -1, "hinging: cannot read %s", las_list[part].c_str());
            if (timed) tm.mark("las ingest");
        }
        LasPart& las = *parts[part];
        if (las.novl == 0) PART_FAIL(f, 2, "No alignments!");
        const int r_begin = las.r_begin, r_end = las.r_end;
        const size_t nr = (size_t)(r_end - r_begin + 1);
        // pairs between reads that are active now (the map only ever receives active x active records, hinging.cpp:478-490)
        f.picks.assign(nr, std::vector<PairPick>());
        std::vector<int64_t> sel;
        std::vector<int32_t> a_of;
        parallel_dynamic((int64_t)nr, 64, [&](int64_t k0, int64_t k1) {
            for (int64_t k = k0; k < k1; k++) {
                const int i = r_begin + (int)k;
                if (!active[(size_t)i]) continue;
                pick_pairs(las, i, USE_TWO_MATCHES, 1, [&](int b) { return active[(size_t)b] && KEEP_ONLY_MAX; }, f.picks[(size_t)k]);
            }
        });
        for (int i = r_begin; i <= r_end; i++) {
            if (!active[(size_t)i]) continue;
            for (auto& p : f.picks[(size_t)(i - r_begin)])
                for (int w = 0; w < 2; w++)
                    if (p.pick[w] >= 0) { sel.push_back(p.pick[w]); a_of.push_back(i); }
        }
        if (timed) tm.mark("pick_pairs");
        // Only the selected overlaps (active x active pairs, best one or two each: thousands out of tens of millions) go to
        // the GPU: their SoA rows and trace bytes are packed here; Match.k is the index into this packed part.
        delete packed[part];
        PackedPart* pk = new PackedPart();
        packed[part] = pk;
        pk->build(las, sel, a_of, n_read);
        f.n_sel = sel.size();
        if (!sel.empty()) {
            int rc = pk->upload(cx);
            if (rc != HINGE_OK) PART_FAIL(f, 2, "upload of the selected overlaps failed (%d): %s", rc, hinge_last_error(cx));
            if (cx == ctx) resident_part = (int)part;
        }
        if (timed) tm.mark("pack + upload selected overlaps");
        std::vector<int64_t> sel_packed(sel.size());
        for (size_t t = 0; t < sel.size(); t++) sel_packed[t] = (int64_t)t;
        f.cls.assign(std::max<size_t>(sel.size(), 1), Classified());
        if (!sel.empty()) {
            int rc = hinge_trim_classify(cx, (int64_t)sel.size(), sel_packed.data(), a_of.data(), ALN_THRESHOLD, THETA, THETA2, (int32_t*)f.cls.data());
            if (rc != HINGE_OK) PART_FAIL(f, 2, "hinge_trim_classify failed (%d): %s", rc, hinge_last_error(cx));
        }
        if (timed) tm.mark("trim_classify (GPU)");
    };
    // the back half: matches of the part's reads, in read order; returns true if it deactivated a read
    auto consume = [&](size_t part, Front& f) -> bool {
        LasPart& las = *parts[part];
        const int r_begin = las.r_begin, r_end = las.r_end;
        bool dropped = false;
        size_t c = 0;
        for (int i = r_begin; i <= r_end; i++) {
            // NOTE: `active` is the state BEFORE this loop for the pair filter above (the reference fills idx_ab for the
            // whole part first, hinging.cpp:478-490), but a read dropped here is inactive for the reads after it.
            auto& pk = f.picks[(size_t)(i - r_begin)];
            if (pk.empty() && !active[(size_t)i]) continue;
            if (!active[(size_t)i]) { for (auto& p : pk) for (int w = 0; w < 2; w++) if (p.pick[w] >= 0) c++; continue; }
            bool contained = false;
            for (auto& p : pk)
                for (int w = 0; w < 2; w++) {
                    if (p.pick[w] < 0) continue;
                    const Classified& r = f.cls[c++];
                    const int64_t k = p.pick[w];
                    Match m;
                    m.a = i; m.b = p.b; m.comp = (int)(las.b_flag[(size_t)k] >> 31);
                    m.ab = las.a_span[(size_t)k * 2]; m.ae = las.a_span[(size_t)k * 2 + 1];
                    m.bb = las.b_span[(size_t)k * 2]; m.be = las.b_span[(size_t)k * 2 + 1];
                    m.a_rs = eff[(size_t)i * 2]; m.a_re = eff[(size_t)i * 2 + 1];
                    m.b_rs = eff[(size_t)p.b * 2]; m.b_re = eff[(size_t)p.b * 2 + 1];
                    m.k = (int64_t)(c - 1); m.part = (int)part; m.c = r;   // c - 1: this overlap's index in the packed part
                    if (active[(size_t)p.b]) contained = contained || (r.type == MT_BCOVERA);
                    if (r.type == MT_FORWARD || r.type == MT_FORWARD_INTERNAL) matches_forward[(size_t)i].push_back(m);
                    else if (r.type == MT_BACKWARD || r.type == MT_BACKWARD_INTERNAL) matches_backward[(size_t)i].push_back(m);
                }
            if (contained) { active[(size_t)i] = 0; dropped = true; }   // "[contained] Should not happen"
        }
        return dropped;
    };
    for (size_t w0 = 0; w0 < las_list.size(); w0 += (size_t)n_ranks) {
        const size_t w1 = std::min(las_list.size(), w0 + (size_t)n_ranks), nw = w1 - w0;
        std::vector<Front> fronts(nw);
        if (nw == 1) front_half(ctxs[0], w0, fronts[0], true);
        else {
            std::vector<std::thread> th;
            for (size_t k = 0; k < nw; k++) th.emplace_back([&, k] { front_half(ctxs[k], w0 + k, fronts[k], false); });
            for (auto& t : th) t.join();
            tm.mark("wave: ingest + pick_pairs + upload + classify");
        }
        {   // every rank's classification rows to every rank: consume() below reads the gathered copy
            std::vector<const void*> ptrs(nw);
            std::vector<int64_t> cnts(nw), offs;
            std::vector<char> all;
            bool ok = true;
            for (size_t k = 0; k < nw; k++) { ptrs[k] = fronts[k].cls.data(); cnts[k] = (int64_t)fronts[k].n_sel; ok = ok && fronts[k].code == 0; }
            if (ok && gather_rows.rccl && gather_rows.gather(ptrs, cnts, (int)sizeof(Classified), all, offs)) {
                for (size_t k = 0; k < nw; k++)
                    if (cnts[k]) memcpy(fronts[k].cls.data(), all.data() + (size_t)offs[k] * sizeof(Classified), (size_t)cnts[k] * sizeof(Classified));
                tm.mark("classified rows all-gather");
            }
        }
        bool stale = false;   // a part of this wave deactivated a read: the later fronts were made for an activity that no longer holds
        for (size_t k = 0; k < nw; k++) {
            if (stale) front_half(ctxs[0], w0 + k, fronts[k], false);
            Front& f = fronts[k];
            if (f.code == -1) { fprintf(stderr, "%s\n", f.error.c_str()); quit(1); }
            if (f.code != 0) { console.error("%s", f.error.c_str()); return f.code; }
            if (consume(w0 + k, f)) stale = true;
        }
    }
#undef PART_FAIL

    tm.mark("matches");
    gather_rows.report("classified matches");
    auto by_weight = [](const Match& x, const Match& y) { return x.c.weight > y.c.weight; };   // compare_overlap_weight
    for (int i = 0; i < n_read; i++)
        if (active[(size_t)i]) {
            std::sort(matches_forward[(size_t)i].begin(), matches_forward[(size_t)i].end(), by_weight);
            std::sort(matches_backward[(size_t)i].begin(), matches_backward[(size_t)i].end(), by_weight);
        }

    // debug dumps in the cwd, hinging.cpp:1074-1150
    {
        FILE* g = fopen("edges.g_out.txt", "w");
        for (int i = 0; i < n_read; i++)
            if (active[(size_t)i])
                for (auto& m : matches_forward[(size_t)i]) if (active[(size_t)m.b]) { print_dump(g, m); break; }
        fprintf(g, "bkw\n");
        for (int i = 0; i < n_read; i++)
            if (active[(size_t)i])
                for (auto& m : matches_backward[(size_t)i]) if (active[(size_t)m.b]) { print_dump(g, m); break; }
        fclose(g);
        FILE* ob = fopen("edges.fwd.backup.txt", "w");
        for (int i = 0; i < n_read; i++)
            if (active[(size_t)i]) for (auto& m : matches_forward[(size_t)i]) if (active[(size_t)m.b]) print_dump(ob, m);
        fclose(ob);
        ob = fopen("edges.bkw.backup.txt", "w");
        for (int i = 0; i < n_read; i++)
            if (active[(size_t)i]) for (auto& m : matches_backward[(size_t)i]) if (active[(size_t)m.b]) print_dump(ob, m);
        fclose(ob);
    }

    tm.mark("sort + debug dumps");
    FILE* out_g1 = fopen((out_name + ".edges.1").c_str(), "w");
    FILE* out_g2 = fopen((out_name + ".edges.2").c_str(), "w");
    FILE* out_hg = fopen((out_name + ".edges.hinges").c_str(), "w");
    FILE* out_hg2 = fopen((out_name + ".edges.hinges2").c_str(), "w");
    FILE* out_greedy = fopen((out_name + ".edges.greedy").c_str(), "w");
    FILE* out_skipped = fopen((out_name + ".edges.skipped").c_str(), "w");

    std::vector<std::vector<Hinge>> hinges_vec((size_t)n_read), killed_hinges_vec((size_t)n_read), new_killed_hinges_vec((size_t)n_read);
    for (int i = 0; i < n_read; i++) {
        auto& mh = marked_hinges[i];
        std::set<std::pair<int, int>> surviving(mh.begin(), mh.end());
        for (auto& h : mh) hinges_vec[(size_t)i].push_back(Hinge{h.first, h.second, true});
        for (auto& r : marked_repeats[i])
            if (surviving.find(r) == surviving.end()) killed_hinges_vec[(size_t)i].push_back(Hinge{r.first, r.second, false});
    }
    {
        FILE* ko = fopen((out + ".killed.hinges").c_str(), "w");
        for (int i = 0; i < n_read; i++) {
            fprintf(ko, "%d ", i);
            for (auto& h : killed_hinges_vec[(size_t)i]) fprintf(ko, "%d %d ", h.type, h.pos);
            fprintf(ko, "\n");
        }
        fclose(ko);
    }

    // hinges bridged by a match, hinging.cpp:1262-1321
    for (int i = 0; i < n_read; i++) {
        if (!active[(size_t)i]) continue;
        for (auto& m : matches_forward[(size_t)i])
            if (m.c.active && active[(size_t)m.b])
                for (auto& h : hinges_vec[(size_t)i])
                    if ((((m.c.eff_ab < h.pos + KILL_HINGE_INTERNAL_ALLOWANCE) && (m.c.type == MT_FORWARD_INTERNAL)) ||
                         ((m.c.eff_ab < h.pos - KILL_HINGE_OVERLAP_ALLOWANCE) && (m.c.type == MT_FORWARD))) && (h.type == 1))
                        h.active = false;
        for (auto& m : matches_backward[(size_t)i])
            if (m.c.active && active[(size_t)m.b])
                for (auto& h : hinges_vec[(size_t)i])
                    if ((((m.c.eff_ae > h.pos - KILL_HINGE_INTERNAL_ALLOWANCE) && (m.c.type == MT_BACKWARD_INTERNAL)) ||
                         ((m.c.eff_ae > h.pos + KILL_HINGE_OVERLAP_ALLOWANCE) && (m.c.type == MT_BACKWARD))) && (h.type == -1))
                        h.active = false;
    }

    // ---- hinge graph, hinging.cpp:1325-1640 ------------------------------------------------------------------
    // GetMatchingPosition of every hinge through every live match: batched per .las part on the GPU, consumed in loop order
    std::vector<std::vector<int>> pos_of_query(parts.size());
    {
        std::vector<std::vector<int64_t>> q_ovl(parts.size());
        std::vector<std::vector<int32_t>> q_pos(parts.size());
        for (int i = 0; i < n_read; i++) {
            if (!active[(size_t)i]) continue;
            for (auto& h : hinges_vec[(size_t)i])
                for (int dirn = 0; dirn < 2; dirn++)
                    for (auto& m : (dirn == 0 ? matches_forward : matches_backward)[(size_t)i])
                        if (m.c.active && active[(size_t)m.b]) { q_ovl[(size_t)m.part].push_back(m.k); q_pos[(size_t)m.part].push_back(h.pos); }
        }
        for (size_t p = 0; p < parts.size(); p++) {
            pos_of_query[p].assign(std::max<size_t>(q_ovl[p].size(), 1), 0);
            if (q_ovl[p].empty()) continue;
            if ((int)p != resident_part) { HH_CHECK(ctx, packed[p]->upload(ctx)); resident_part = (int)p; }
            HH_CHECK(ctx, hinge_matching_position(ctx, (int64_t)q_ovl[p].size(), q_ovl[p].data(), q_pos[p].data(), pos_of_query[p].data()));
        }
    }
    int num_hinges = 0;
    std::vector<int> node_base((size_t)n_read + 1, 0);
    for (int i = 0; i < n_read; i++) { node_base[(size_t)i] = num_hinges; num_hinges += (int)hinges_vec[(size_t)i].size(); }
    std::vector<std::pair<int, int>> graph_edges;
    tm.mark("hinges + matching positions");
    FILE* out_hgraph = fopen((out_name + ".hgraph").c_str(), "w");
    FILE* out_debug = fopen((out_name + ".debug").c_str(), "w");
    fclose(fopen("overlap_debug.txt", "w"));
    {
        std::vector<size_t> cursor(parts.size(), 0);
        for (int i = 0; i < n_read; i++) {
            if (!active[(size_t)i]) continue;
            for (int k = 0; k < (int)hinges_vec[(size_t)i].size(); k++) {
                const Hinge hk = hinges_vec[(size_t)i][(size_t)k];
                for (int dirn = 0; dirn < 2; dirn++) {
                    const int own_type = dirn == 0 ? 1 : -1;   // hinge type whose line is written (i, b) in this direction
                    for (auto& m : (dirn == 0 ? matches_forward : matches_backward)[(size_t)i]) {
                        if (!(m.c.active && active[(size_t)m.b])) continue;
                        const int pos_B = pos_of_query[(size_t)m.part][cursor[(size_t)m.part]++];
                        const int rev_int = m.comp == 1 ? 1 : 0;
                        const int req_type = m.comp == 1 ? -hk.type : hk.type;
                        const int b_id = m.b;
                        for (int l = 0; l < (int)hinges_vec[(size_t)b_id].size(); l++) {
                            const Hinge& hb = hinges_vec[(size_t)b_id][(size_t)l];
                            if ((hb.pos < pos_B + MATCHING_HINGE_SLACK) && (hb.pos > pos_B - MATCHING_HINGE_SLACK) && req_type == hb.type) {
                                if (hk.type == own_type) {
                                    graph_edges.push_back(std::make_pair(node_base[(size_t)i] + k, node_base[(size_t)b_id] + l));
                                    fprintf(out_hgraph, "%d %d %d %d %d %d\n", i, b_id, hk.pos, hb.pos, 1, rev_int);
                                } else {
                                    graph_edges.push_back(std::make_pair(node_base[(size_t)b_id] + l, node_base[(size_t)i] + k));
                                    fprintf(out_hgraph, "%d %d %d %d %d %d\n", b_id, i, hb.pos, hk.pos, 1, rev_int);
                                }
                            }
                        }
                        for (auto& kb : killed_hinges_vec[(size_t)b_id]) {
                            if (!((kb.pos < pos_B + MATCHING_HINGE_SLACK) && (kb.pos > pos_B - MATCHING_HINGE_SLACK))) continue;
                            const bool type_ok = req_type == kb.type;
                            if (type_ok) {
                                if (hk.type == own_type) fprintf(out_hgraph, "%d %d %d %d %d %d\n", i, b_id, hk.pos, kb.pos, 0, rev_int);
                                else fprintf(out_hgraph, "%d %d %d %d %d %d\n", b_id, i, kb.pos, hk.pos, 0, rev_int);
                            }
                            if (dirn == 0) {   // forward: inside the type test (hinging.cpp:1467-1493)
                                if (type_ok && m.c.type == MT_FORWARD) {
                                    new_killed_hinges_vec[(size_t)i].push_back(Hinge{hk.pos, hk.type, false});
                                    if (hk.type == -1) {
                                        print_dump(out_debug, m);
                                        fprintf(out_debug, "%d %d %d %d\n", hk.pos, hk.type, kb.pos, kb.type);
                                    }
                                }
                            } else if (m.c.type == MT_BACKWARD) {   // backward: OUTSIDE the type test (hinging.cpp:1612-1622)
                                new_killed_hinges_vec[(size_t)i].push_back(Hinge{hk.pos, hk.type, false});
                            }
                        }
                    }
                }
            }
        }
    }
    fclose(out_hgraph);
    fclose(out_debug);
    {
        std::vector<int> comp = components(num_hinges, graph_edges);
        std::map<int, int> size;
        for (int v : comp) size[v] += 1;
        int node = 0;
        for (int i = 0; i < n_read; i++)
            for (auto& h : hinges_vec[(size_t)i]) { if (size[comp[(size_t)node]] < MIN_CONNECTED_COMPONENT_SIZE) h.active = false; node++; }
    }
    {
        FILE* hl = fopen((out_name + ".hinge.list").c_str(), "w");
        int n = 0;
        for (int i = 0; i < n_read; i++)
            for (size_t j = 0; j < hinges_vec[(size_t)i].size(); j++)
                if (active[(size_t)i] && hinges_vec[(size_t)i][j].active) { fprintf(hl, "%d %d %d\n", i, marked_hinges[i][j].first, marked_hinges[i][j].second); n++; }
        fclose(hl);
        console.info("after filter %d active hinges", n);
    }

    // pure greedy graph, hinging.cpp:1724-1860
    for (int i = 0; i < n_read; i++) {
        if (!active[(size_t)i]) continue;
        for (int dirn = 0; dirn < 2; dirn++) {
            int cnt = 0;
            for (auto& m : (dirn == 0 ? matches_forward : matches_backward)[(size_t)i]) {
                if (!(m.c.active && m.c.type == (dirn == 0 ? MT_FORWARD : MT_BACKWARD) && active[(size_t)m.b])) continue;
                if (cnt < 1) {
                    print_overlap(out_greedy, m);
                    if (m.comp == 0) print_g(out_g1, "%d %d %d [%d %d] [%d %d] [%d %d] [%d %d]\n", m.a, m.b, m);
                    else print_g(out_g1, "%d %d' %d [%d %d] [%d %d] [%d %d] [%d %d]\n", m.a, m.b, m);
                    if (m.comp == 0) print_g(out_g2, "%d' %d' %d [%d %d] [%d %d] [%d %d] [%d %d]\n", m.b, m.a, m);
                    else print_g(out_g2, "%d %d' %d [%d %d] [%d %d] [%d %d] [%d %d]\n", m.b, m.a, m);
                }
                cnt++;
            }
        }
    }

    fclose(fopen("hinge_debug.txt", "w"));
    // best-overlap selection, hinging.cpp:1911-2148: one walk per (read, direction) on the GPU (k_select_edges behind
    // hinge_select_edges); the host packs the classified matches and the hinge tables and prints what was chosen, in read order
    {
        std::vector<int64_t> off_f((size_t)n_read + 1, 0), off_b((size_t)n_read + 1, 0), h_off((size_t)n_read + 1, 0), k_off((size_t)n_read + 1, 0);
        int64_t nm = 0;
        for (int i = 0; i < n_read; i++) nm += (int64_t)matches_forward[(size_t)i].size() + (int64_t)matches_backward[(size_t)i].size();
        std::vector<int32_t> rec((size_t)nm * 9);
        std::vector<const Match*> who((size_t)nm);
        int64_t w = 0;
        for (int dirn = 0; dirn < 2; dirn++) {
            std::vector<int64_t>& off = dirn == 0 ? off_f : off_b;
            for (int i = 0; i < n_read; i++) {
                off[(size_t)i] = w;
                for (auto& m : (dirn == 0 ? matches_forward : matches_backward)[(size_t)i]) {
                    int32_t* r = &rec[(size_t)w * 9];
                    r[0] = m.b; r[1] = m.comp; r[2] = m.c.type; r[3] = m.c.active; r[4] = m.c.weight; r[5] = m.c.eff_bb; r[6] = m.c.eff_be; r[7] = m.bb; r[8] = m.be;
                    who[(size_t)w++] = &m;
                }
            }
            off[(size_t)n_read] = w;
        }
        std::vector<int32_t> h_rec, k_rec;
        for (int i = 0; i < n_read; i++) {
            h_off[(size_t)i] = (int64_t)h_rec.size() / 3;
            k_off[(size_t)i] = (int64_t)k_rec.size() / 2;
            for (auto& h : hinges_vec[(size_t)i]) { h_rec.push_back(h.pos); h_rec.push_back(h.type); h_rec.push_back(h.active ? 1 : 0); }
            for (auto& h : new_killed_hinges_vec[(size_t)i]) { k_rec.push_back(h.pos); k_rec.push_back(h.type); }
        }
        h_off[(size_t)n_read] = (int64_t)h_rec.size() / 3;
        k_off[(size_t)n_read] = (int64_t)k_rec.size() / 2;
        std::vector<int32_t> chosen((size_t)n_read * 2), chosen_hpos((size_t)n_read * 2), poison((size_t)std::max<int64_t>(nm, 1), 0);
        HH_CHECK(ctx, hinge_select_edges(ctx, n_read, active.data(), nm, off_f.data(), off_b.data(), rec.data(), h_off.data(), h_rec.data(), k_off.data(),
                                         k_rec.data(), HINGE_TOLERANCE, HINGE_SLACK, chosen.data(), chosen_hpos.data(), poison.data()));
        for (int i = 0; i < n_read; i++) {
            if (!active[(size_t)i]) continue;
            for (int dirn = 0; dirn < 2; dirn++) {
                const std::vector<int64_t>& off = dirn == 0 ? off_f : off_b;
                for (int64_t j = off[(size_t)i]; j < off[(size_t)i + 1]; j++)        // .edges.skipped: one line per killed hinge that poisoned the match
                    for (int t = 0; t < poison[(size_t)j]; t++) print_overlap(out_skipped, *who[(size_t)j]);
                const int pick = chosen[(size_t)dirn * (size_t)n_read + (size_t)i];
                if (pick >= 0) {
                    print_overlap(out_hg, *who[(size_t)pick]);
                    print_overlap2(out_hg2, *who[(size_t)pick], chosen_hpos[(size_t)dirn * (size_t)n_read + (size_t)i]);
                } else {
                    fprintf(deadend_out, "%d\t matches_%s size: %d\n", i, dirn == 0 ? "forward" : "backward", (int)(off[(size_t)i + 1] - off[(size_t)i]));
                }
            }
        }
    }
    fclose(out_g1); fclose(out_g2); fclose(out_hg); fclose(out_hg2); fclose(out_greedy); fclose(out_skipped); fclose(deadend_out);
    console.info("sort and output finished");
    if (getenv("HINGE_SLOW_EXIT") || pipeline().on) { for (size_t k = 0; k < parts.size(); k++) if (!(loader.shared && k == 0)) delete parts[k]; for (auto* p : packed) delete p; }
    return finish(ctx, tm);
}
