// get_maximal_reads  ==  `hinge maximal --db DB --las LAS[.las] [--mlas] -x PREFIX --config nominal.ini`
// Same flags, inputs, outputs (.max, .contained.txt, rewritten .coverage.txt, emptied .homologous.txt /
// .filtered.fasta) and exit codes as src/maximal/maximal.cpp.  ProcessAlignment of every (A, B) pair's
// best one or two overlaps (maximal.cpp:780-850) runs in k_trim_classify; the order-dependent containment
// resolution (maximal.cpp:805-857: a read is dropped only if its container is still active) stays a
// sequential host pass over 10 ints per classified overlap.
#include "pairs.h"

using namespace hh;

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
    cmdp.add_string("prefix", 'x', "prefix of (intermediate) output", false, "out");
    cmdp.add_string("restrictreads", 'r', "restrict to reads in the file", false, "");
    cmdp.add_string("log", 'g', "log folder name", false, "log");
    cmdp.add_flag("mlas", '\0', "multiple las files");
    cmdp.add_flag("debug", '\0', "debug mode");
    cmdp.parse_check(argc, argv);

    PhaseTimer tm("maximal");
    CtxInit gpu;
    gpu.start();
    Log console;
    console.open(cmdp.get("log"));
    console.info("Getting maximal reads");
    const std::string name_db = cmdp.get("db"), name_las_base = cmdp.get("las"), name_paf = cmdp.get("paf"), name_fasta = cmdp.get("fasta");
    const std::string name_config = cmdp.get("config"), out = cmdp.get("prefix");
    const bool db_and_las = !name_db.empty() && !name_las_base.empty(), db_or_las = !name_db.empty() || !name_las_base.empty();
    const bool fa_and_paf = !name_fasta.empty() && !name_paf.empty(), fa_or_paf = !name_fasta.empty() || !name_paf.empty();
    if (db_or_las && fa_or_paf) { console.error("Pass in either a db and a las or a fasta and a paf"); return 1; }
    if (!fa_and_paf && !db_and_las) { console.error("Pass in at least one of the following two combinations: a db and a las or a fasta and a paf"); return 1; }
    const bool mlas = cmdp.exist("mlas");
    if (mlas && !db_and_las) { console.error("--mlas works only with db and las"); return 1; }

    ReadDB db;
    std::vector<std::string> las_list;
    if (fa_and_paf) {
        if (read_fasta_lengths(name_fasta, db.rlen) != 0) { fprintf(stderr, "get_maximal_reads: cannot read %s\n", name_fasta.c_str()); quit(1); }
        las_list.push_back(name_paf);
    } else {
        if (db.open(name_db) != 0) { fprintf(stderr, "get_maximal_reads: Could not open database %s\n", name_db.c_str()); quit(1); }
        const std::string name_las = las_name(name_las_base, mlas);
        if (mlas) las_list = las_parts(name_las); else las_list.push_back(name_las);
    }
    const int n_read = (int)db.rlen.size();
    console.info("# Reads: %d", n_read);

    Config ini(name_config);
    if (ini.error < 0) { console.warn("Can't load %s", name_config.c_str()); return 1; }
    const int LENGTH_THRESHOLD = (int)ini.get_int("filter", "length_threshold", -1);
    const int ALN_THRESHOLD = (int)ini.get_int("filter", "aln_threshold", -1);
    const int THETA = (int)ini.get_int("filter", "theta", -1);
    const int THETA2 = (int)ini.get_int("filter", "theta2", 0);
    const bool USE_TWO_MATCHES = ((int)ini.get_int("layout", "use_two_matches", 1)) != 0;
    const int reso = 40;

    FILE* f_cov = fopen((out + ".coverage.txt").c_str(), "w");
    fclose(fopen((out + ".homologous.txt").c_str(), "w"));
    fclose(fopen((out + ".filtered.fasta").c_str(), "w"));
    FILE* f_contained = fopen((out + ".contained.txt").c_str(), "w");
    FILE* f_max = fopen((out + ".max").c_str(), "w");
    if (!f_cov || !f_contained || !f_max) { console.error("cannot open output files with prefix %s", out.c_str()); return 2; }

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
    for (int i = 0; i < n_read; i++)
        if (eff[(size_t)i * 2 + 1] - eff[(size_t)i * 2] < LENGTH_THRESHOLD) active[(size_t)i] = 0;

    tm.mark("db + ini + mas");
    // ---- ranks: one host thread + one context per visible GPU for a --mlas run (HINGE_RANKS overrides) --------------------------
    // Everything a part needs up to its list of containment candidates depends on that part only (a read's pile-up lies in its
    // own part; whether a read is examined at all depends on its INITIAL activity), so the parts of a wave run side by side,
    // each on its own GPU.  What the reference does sequentially - a read is dropped only if one of its containers is still
    // active at that moment - stays one host pass over the candidate rows in part order (exchange 4 of the sharded path).
    const int n_ranks = rank_count(las_list.size(), fa_and_paf);
    PartLoader loader;
    loader.paf = fa_and_paf;
    loader.single = las_list.size() == 1;
    if (!las_list.empty() && n_ranks == 1) loader.preload(las_list[0], db.rlen);
    tm.mark("las ingest (part 1) || HIP init");
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
    tm.mark("ctx_create + set_reads");
    // the ranks' containment candidates reach the one sequential resolution pass as ONE all-gather over RCCL per wave (maximal.cpp:805-857
    // is that pass; HINGE_HOST_EXCHANGE=1 or ranks that share a device: host concatenation)
    RowGather gather_rows;
    gather_rows.init(ctxs, "containment candidates", console);

    struct PartOut {
        int code = 0;
        std::string error;
        int r_begin = 0, r_end = -1;
        int64_t novl = 0, n_classified = 0;
        std::vector<int32_t> nb;      // .coverage.txt
        UVec<int32_t> cov;
        std::vector<int32_t> pairs;   // (a, b) rows of the selected overlaps in which B covers A, in selection order
    };
#define PART_FAIL(o, c, ...)                                            \
    do {                                                                 \
        char _b[512];                                                    \
        snprintf(_b, sizeof(_b), __VA_ARGS__);                           \
        (o).code = (c); (o).error = _b;                                  \
        return;                                                          \
    } while (0)
#define PART_CHECK(o, cx, call)                                                                                  \
    do {                                                                                                          \
        int _rc = (call);                                                                                         \
        if (_rc != HINGE_OK) PART_FAIL(o, _rc == HINGE_E_UNDEFINED ? 1 : 2, "%s failed (%d): %s", #call, _rc, hinge_last_error(cx)); \
    } while (0)
    // timed = a single part on the main thread: its .coverage.txt is then written by `cov_writer` and its trace points uploaded
    // by a helper thread WHILE the host threads group the pairs (the GPU is idle in between; 150 of the stage's 700 ms)
    std::thread cov_writer;
    auto part_work = [&](hinge_ctx* cx, size_t part, PartOut& o, bool timed) {
        int lrc = 0;
        const bool part_shared = loader.shared && part == 0;      // `hinge pipeline`: the process's part, not this stage's
        struct Owner { LasPart* p; bool shared; ~Owner() { if (!shared) delete p; } } las_owner{loader.take(part, las_list[part], db.rlen, lrc), part_shared};
        LasPart& las = *las_owner.p;
        if (lrc == -2) PART_FAIL(o, 2, "%s is not sorted by A read", las_list[part].c_str());
        if (lrc == -3) PART_FAIL(o, 1, "%s: a read name without \"/id/\" or an id outside the FASTA (the reference crashes here)", las_list[part].c_str());
        if (lrc != 0) PART_FAIL(o, -1, "get_maximal_reads: cannot read %s", las_list[part].c_str());
        if (timed) tm.mark("las ingest");
        if (las.novl == 0) PART_FAIL(o, 1, "No alignments!");
        o.novl = las.novl;
        const int r_begin = las.r_begin, r_end = las.r_end;
        o.r_begin = r_begin; o.r_end = r_end;
        const size_t nr = (size_t)(r_end - r_begin + 1);
        PART_CHECK(o, cx, hinge_set_pileups_packed(cx, r_begin, r_end, las.n_kept(), las.row_ptr.data(), las.a_span.data(), las.b_span.data(), las.b_flag.data(),
                                                   nullptr, las.max_pile, las.spans_in_range ? 1 : 0, 0));   // no coverage passes here: no span copy
        int trace_rc = HINGE_OK;
        auto upload_traces = [&] {
            static const uint8_t no_trace[1] = {0};   // PAF: no trace points, ProcessAlignment(trim = false)
            trace_rc = hinge_set_trim(cx, las.is_paf ? 0 : 1);
            if (trace_rc != HINGE_OK) return;
            // the part form reads the .las image itself (k_trim_classify_image): one 32-bit offset per overlap goes up instead of
            // trace_off + tlen (12 bytes), and the kernel reads no SoA column.  HINGE_K4_SOA=1 / HINGE_K4_ROWS: the column form
            if (!las.is_paf && !getenv("HINGE_K4_SOA") && !getenv("HINGE_K4_ROWS") && las.build_image_table())
                trace_rc = hinge_set_las_image(cx, las.file.p, (int64_t)las.file.n, las.img_win_base.data(), las.img_rec_rel.data(), las.tbytes, 0);
            else
                trace_rc = hinge_set_traces(cx, las.is_paf ? no_trace : las.file.p, las.is_paf ? 1 : (int64_t)las.file.n, las.trace_off.data(), las.tlen.data(), las.tbytes, 0);
        };
        if (!timed) { upload_traces(); PART_CHECK(o, cx, trace_rc); }
        if (timed) tm.mark("set_pileups (H2D)");
        // .coverage.txt is truncated and rewritten with the same content (maximal.cpp:517,659-685)
        {
            o.nb.resize(nr);
            PART_CHECK(o, cx, hinge_filter_coverage_bins(cx, r_begin, r_end, reso, 0, o.nb.data(), nullptr, 0));
            int64_t tot = 0;
            for (size_t k = 0; k < nr; k++) tot += o.nb[k];
            o.cov.resize((size_t)std::max<int64_t>(tot, 1));   // filled by the copy from the device: no zero fill, huge pages
            PART_CHECK(o, cx, hinge_filter_coverage_bins(cx, r_begin, r_end, reso, 0, o.nb.data(), o.cov.data(), tot));
        }
        if (timed) tm.mark("coverage bins");
        std::thread trace_upload;
        if (timed) {   // from here to the classification the main thread makes no library call: the upload has the context to itself
            cov_writer = std::thread([&o, f_cov, r_begin, reso] { write_coverage_txt(f_cov, r_begin, o.nb, o.cov, reso); });
            trace_upload = std::thread(upload_traces);
        }
        struct JoinGuard { std::thread& t; ~JoinGuard() { if (t.joinable()) t.join(); } } trace_guard{trace_upload};
        // pairs of every read that is active when its turn comes (activity only changes at a read's own turn).
        // Every read's grouping is independent: chunks of 64 reads go to host threads, each chunk emits its selected
        // overlaps (best one or two per (A,B) pair, in the map's iteration order) into its own buffer; the buffers
        // are stitched in read order.  B of a selected overlap is in b_flag, so nothing else has to be kept.
        const int64_t CH = 64, n_chunks = ((int64_t)nr + CH - 1) / CH;
        std::vector<std::vector<int64_t>> chunk_sel((size_t)n_chunks);
        std::vector<int32_t> n_sel_of(nr, 0);
        parallel_dynamic(n_chunks, 1, [&](int64_t c0, int64_t c1) {
            std::vector<PairPick> pp;
            for (int64_t c = c0; c < c1; c++) {
                const int64_t k0 = c * CH, k1 = std::min<int64_t>((int64_t)nr, k0 + CH);
                std::vector<int64_t>& out = chunk_sel[(size_t)c];
                out.reserve((size_t)(las.rec_row_ptr[(size_t)(r_begin + k1)] - las.rec_row_ptr[(size_t)(r_begin + k0)]));
                for (int64_t k = k0; k < k1; k++) {
                    const int i = r_begin + (int)k;
                    if (!active[(size_t)i]) continue;
                    pick_pairs(las, i, USE_TWO_MATCHES, 2, [](int) { return true; }, pp);
                    int cnt = 0;
                    for (auto& p : pp)
                        for (int w = 0; w < 2; w++)
                            if (p.pick[w] >= 0) { out.push_back(p.pick[w]); cnt++; }
                    n_sel_of[(size_t)k] = cnt;
                }
            }
        });
        std::vector<int64_t> chunk_base((size_t)n_chunks + 1, 0);
        for (int64_t c = 0; c < n_chunks; c++) chunk_base[(size_t)c + 1] = chunk_base[(size_t)c] + (int64_t)chunk_sel[(size_t)c].size();
        const int64_t n_sel = chunk_base[(size_t)n_chunks];
        UVec<int64_t> sel;
        UVec<int32_t> a_of;
        UVec<uint8_t> mtype;
        sel.resize((size_t)std::max<int64_t>(n_sel, 1)); a_of.resize((size_t)std::max<int64_t>(n_sel, 1)); mtype.resize((size_t)std::max<int64_t>(n_sel, 1));
        parallel_dynamic(n_chunks, 16, [&](int64_t c0, int64_t c1) {
            for (int64_t c = c0; c < c1; c++) {
                int64_t o2 = chunk_base[(size_t)c];
                if (!chunk_sel[(size_t)c].empty()) memcpy(sel.data() + o2, chunk_sel[(size_t)c].data(), chunk_sel[(size_t)c].size() * sizeof(int64_t));
                const int64_t k0 = c * CH, k1 = std::min<int64_t>((int64_t)nr, k0 + CH);
                for (int64_t k = k0; k < k1; k++)
                    for (int t = 0; t < n_sel_of[(size_t)k]; t++) a_of[(size_t)o2++] = r_begin + (int)k;
                std::vector<int64_t>().swap(chunk_sel[(size_t)c]);
            }
        });
        if (timed) tm.mark("pick_pairs || traces H2D || coverage.txt");
        if (trace_upload.joinable()) trace_upload.join();
        PART_CHECK(o, cx, trace_rc);
        if (timed) tm.mark("traces H2D (rest)");
        // Nearly every overlap is selected (one or two per (A, B) pair), so the whole part is classified in storage order -
        // coalesced, nothing to upload - and the selected ones are picked out of the result.
        {
            UVec<uint8_t> all_types;
            all_types.resize((size_t)std::max<int64_t>(las.n_kept(), 1));
            PART_CHECK(o, cx, hinge_trim_classify_part(cx, ALN_THRESHOLD, THETA, THETA2, all_types.data()));
            parallel_chunks(n_sel, host_threads(), [&](int, int64_t b, int64_t e) {
                for (int64_t c = b; c < e; c++) mtype[(size_t)c] = all_types[(size_t)sel[(size_t)c]];
            });
        }
        if (timed) tm.mark("trim_classify (GPU)");
        // one (a, b) row per selected overlap that classified as BCOVERA (the rows keep the order of `sel`: counted and
        // written in contiguous pieces by the host threads)
        o.n_classified = n_sel;
        const int T = host_threads();
        std::vector<int64_t> piece_rows((size_t)T + 1, 0);
        parallel_chunks(n_sel, T, [&](int c, int64_t b, int64_t e) {
            int64_t m = 0;
            for (int64_t k = b; k < e; k++) m += mtype[(size_t)k] == MT_BCOVERA;
            piece_rows[(size_t)c + 1] = m;
        });
        for (int c = 0; c < T; c++) piece_rows[(size_t)c + 1] += piece_rows[(size_t)c];
        o.pairs.resize((size_t)(2 * piece_rows[(size_t)T]));
        parallel_chunks(n_sel, T, [&](int c, int64_t b, int64_t e) {
            int32_t* out = o.pairs.data() + 2 * piece_rows[(size_t)c];
            for (int64_t k = b; k < e; k++)
                if (mtype[(size_t)k] == MT_BCOVERA) {
                    *out++ = a_of[(size_t)k];
                    *out++ = (int32_t)(las.b_flag[(size_t)sel[(size_t)k]] & 0x7fffffffu);
                }
        });
        if (timed) tm.mark("candidate rows");
    };

    for (size_t w0 = 0; w0 < las_list.size(); w0 += (size_t)n_ranks) {
        const size_t w1 = std::min(las_list.size(), w0 + (size_t)n_ranks), nw = w1 - w0;
        std::vector<PartOut> outs(nw);
        if (nw == 1) {
            part_work(ctxs[0], w0, outs[0], true);
            tm.mark("part teardown");   // (121 ms for a 3.5 GB part: its columns and the mapped .las go back; left to _exit() instead, the kernel spends the same at exit - measured, round 5)
        }
        else {
            std::vector<std::thread> th;
            for (size_t k = 0; k < nw; k++) th.emplace_back([&, k] { part_work(ctxs[k], w0 + k, outs[k], false); });
            for (auto& t : th) t.join();
            tm.mark("wave: ingest + H2D + bins + pick_pairs + classify");
        }
        // every rank's candidate rows (A, B) to every rank; the resolution below reads the gathered copy, part by part
        std::vector<char> all_pairs;
        std::vector<int64_t> pair_off;
        {
            std::vector<const void*> ptrs(nw);
            std::vector<int64_t> cnts(nw);
            bool ok = true;
            for (size_t k = 0; k < nw; k++) { ptrs[k] = outs[k].pairs.data(); cnts[k] = (int64_t)(outs[k].pairs.size() / 2); ok = ok && outs[k].code == 0; }
            if (ok) { gather_rows.gather(ptrs, cnts, 2 * (int)sizeof(int32_t), all_pairs, pair_off); if (gather_rows.rccl) tm.mark("candidate rows all-gather"); }
        }
        for (size_t k = 0; k < nw; k++) {
            PartOut& o = outs[k];
            console.info("name of las: %s", las_list[w0 + k].c_str());
            if (o.code != 0 && cov_writer.joinable()) cov_writer.join();
            if (o.code == -1) { fprintf(stderr, "%s\n", o.error.c_str()); quit(1); }
            if (o.code != 0) { console.error("%s", o.error.c_str()); return o.code; }
            const int r_begin = o.r_begin, r_end = o.r_end;
            const size_t nr = (size_t)(r_end - r_begin + 1);
            if (!cov_writer.joinable()) { write_coverage_txt(f_cov, r_begin, o.nb, o.cov, reso); tm.mark("coverage.txt"); }
            // sequential containment resolution, maximal.cpp:780-858
            std::vector<int32_t> containing((size_t)n_read);
            const int32_t* part_pairs = all_pairs.empty() ? o.pairs.data() : (const int32_t*)all_pairs.data() + 2 * pair_off[k];
            if (hinge_resolve_containment(n_read, active.data(), (int64_t)(o.pairs.size() / 2), part_pairs, containing.data()) != HINGE_OK) {
                console.error("containment resolution: malformed candidate list");
                if (cov_writer.joinable()) cov_writer.join();
                return 2;
            }
            for (int i = r_begin; i <= r_end; i++)
                if (containing[(size_t)i] >= 0) fprintf(f_contained, "%d\t%d\n", i, containing[(size_t)i]);
            int n_active = 0;
            for (int i = r_begin; i <= r_end; i++)
                if (active[(size_t)i]) { n_active++; fprintf(f_max, "%d\n", i); }
            tm.mark("containment + max txt");
            if (cov_writer.joinable()) { cov_writer.join(); tm.mark("coverage.txt (rest)"); }
            console.info("classified %lld overlaps; removed contained reads, active reads: %d / %zu", (long long)o.n_classified, n_active, nr);
        }
    }
    fclose(f_cov); fclose(f_contained); fclose(f_max);
    gather_rows.report("containment candidates");
    return finish(ctx, tm);
}
