// Reads_filter  ==  `hinge filter --db DB --las LAS[.las] [--mlas] -x PREFIX --config nominal.ini`
// Same flags, same inputs, same output files, same exit codes as src/filter/filter.cpp; the pile-up
// arithmetic (filter.cpp:529-1070) runs in the HIP kernels behind include/hinge_hip.h.
#include <functional>
#include <set>
#include "host_common.h"

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

    PhaseTimer tm("filter");
    CtxInit gpu;
    gpu.start();
    Log console;
    console.open(cmdp.get("log"));
    console.info("Reads filtering");
    const std::string name_db = cmdp.get("db"), name_las_base = cmdp.get("las"), name_paf = cmdp.get("paf"), name_fasta = cmdp.get("fasta");
    const std::string name_config = cmdp.get("config"), out = cmdp.get("prefix");
    const bool db_and_las = !name_db.empty() && !name_las_base.empty(), db_or_las = !name_db.empty() || !name_las_base.empty();
    const bool fa_and_paf = !name_fasta.empty() && !name_paf.empty(), fa_or_paf = !name_fasta.empty() || !name_paf.empty();
    if (db_or_las && fa_or_paf) { console.error("Pass in either a db and a las or a fasta and a paf"); return 1; }
    if (!fa_and_paf && !db_and_las) { console.error("Pass in at least one of the following two combinations: a db and a las or a fasta and a paf"); return 1; }
    const bool mlas = cmdp.exist("mlas");
    if (mlas && !db_and_las) { console.error("--mlas works only with db and las"); return 1; }

    ReadDB db;
    std::vector<std::vector<uint8_t>> qv;
    bool has_qv = false;
    std::vector<std::string> las_list;
    int64_t novl0 = 0;
    int tspace0 = 100;
    if (fa_and_paf) {   // reads from the FASTA, one "part" = the PAF, no QV track (filter.cpp:289-291,467-468)
        if (read_fasta_lengths(name_fasta, db.rlen) != 0) { fprintf(stderr, "Reads_filter: cannot read %s\n", name_fasta.c_str()); quit(1); }
        las_list.push_back(name_paf);
    } else {
        if (db.open(name_db) != 0) { fprintf(stderr, "Reads_filter: Could not open database %s\n", name_db.c_str()); quit(1); }
        has_qv = db.load_qual(qv);
        const std::string name_las = las_name(name_las_base, mlas);
        if (mlas) las_list = las_parts(name_las); else las_list.push_back(name_las);
        if (las_list.empty()) { console.error("No alignments!"); return 1; }
        if (LasPart::header(las_list[0], novl0, tspace0) != 0) { fprintf(stderr, "Reads_filter: cannot open %s\n", las_list[0].c_str()); quit(1); }
    }
    const int n_read = (int)db.rlen.size();
    console.info("# Reads: %d", n_read);
    std::vector<int32_t> qvm;
    if (has_qv) qv_masks(qv, tspace0, qvm);

    Config ini(name_config);
    if (ini.error < 0) { console.warn("Can't load %s", name_config.c_str()); return 1; }
    hinge_filter_params P = filter_params_from(ini, has_qv);
    console.info("use_qv_mask set to %d", P.use_qv_mask);
    console.info("MIN_COV = %d CUT_OFF = %d THETA = %d EST_COV = %d", P.min_cov, P.cut_off, P.theta, P.est_cov);

    tm.mark("db + qual + ini");
    // --restrictreads FILE: one read id per line (filter.cpp:300-316)
    std::set<int> reads_to_keep;
    const std::string name_restrict = cmdp.get("restrictreads");
    if (!name_restrict.empty()) {
        FILE* rf = fopen(name_restrict.c_str(), "r");
        char line[256];
        while (rf && fgets(line, sizeof(line), rf)) reads_to_keep.insert(atoi(line));   // `ss >> num` of an unparsable line gives 0
        if (rf) fclose(rf);
        console.info("Restricting to %zu reads", reads_to_keep.size());
    }
    // ---- how many ranks: one per visible GPU for a --mlas run (HINGE_RANKS overrides; more ranks than devices share them) -----
    // A rank = one host thread + one hinge_ctx.  The parts of a --mlas run go to the ranks in waves of n_ranks consecutive parts;
    // what the reference's sequential loop carries from part to part is exchanged between the ranks of a wave (see run_wave):
    // the running MIN_COV (a prefix maximum over the parts) and the mask table (a part sees the masks of the parts before it).
    const int n_ranks = rank_count(las_list.size(), !reads_to_keep.empty() || fa_and_paf);   // --restrictreads grows its read set from part to part: sequential
    PartLoader loader;
    loader.pairs = !reads_to_keep.empty();   // the neighbours of the listed reads need the per-record B column
    loader.paf = fa_and_paf;
    loader.span16 = true;
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
    // one RCCL communicator over the ranks' GPUs; refused (ranks share a device, no librccl, HINGE_HOST_EXCHANGE=1): host exchanges
    const bool use_rccl = n_ranks > 1 && hinge_comm_create(ctxs.data(), n_ranks) == HINGE_OK;
    if (n_ranks > 1) console.info("%d ranks, mask rows %s", n_ranks, use_rccl ? "over RCCL (one all-gather per wave)" : "through the host");
    for (int r = 0; r < n_ranks; r++) {
        HH_CHECK(ctxs[(size_t)r], hinge_set_reads(ctxs[(size_t)r], n_read, db.rlen.data(), has_qv ? qvm.data() : nullptr));
        HH_CHECK(ctxs[(size_t)r], hinge_filter_set_min_cov(ctxs[(size_t)r], P.min_cov));
    }

    tm.mark("ctx_create + set_reads");
    FILE* f_cov = fopen((out + ".coverage.txt").c_str(), "w");
    fclose(fopen((out + ".homologous.txt").c_str(), "w"));
    FILE* f_rep = fopen((out + ".repeat.txt").c_str(), "w");
    fclose(fopen((out + ".filtered.fasta").c_str(), "w"));
    FILE* f_hg = fopen((out + ".hinges.txt").c_str(), "w");
    FILE* f_mask = fopen((out + ".mas").c_str(), "w");
    FILE* f_cmask = fopen((out + ".cmas").c_str(), "w");
    FILE* f_covflag = fopen((out + ".cov.flag").c_str(), "w");
    FILE* f_selfflag = fopen((out + ".self.flag").c_str(), "w");
    if (!f_cov || !f_rep || !f_hg || !f_mask || !f_cmask || !f_covflag || !f_selfflag) { console.error("cannot open output files with prefix %s", out.c_str()); return 2; }

    // Everything one part produces, kept until its turn to be written comes (files are written in part order).
    struct PartOut {
        int code = 0;                 // 0 ok, else the exit code of the program
        std::string error;            // console.error text for a non-zero code
        std::unique_ptr<LasPart> las;
        int r_begin = 0, r_end = -1;
        hinge_cov_estimate est{};
        std::set<int> self_match;
        std::vector<int32_t> mask, cmask, nb, pos, type;
        std::vector<uint8_t> flags, is_hinge;
        std::vector<int64_t> coff, off;
        UVec<int32_t> cov;
    };
    int running_min_cov = P.min_cov;   // MIN_COV as the sequential loop carries it (filter.cpp:677-678)
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
    // phase A of a part: ingest, upload, coverage statistics, the part's own median
    auto phase_a = [&](hinge_ctx* cx, size_t part, PartOut& o) {
        int lrc = 0;
        o.las.reset(loader.take(part, las_list[part], db.rlen, lrc));   // (`hinge pipeline`, single .las: the process's part - the wave that holds it is never freed, see below)
        LasPart& las = *o.las;
        if (lrc == -2) PART_FAIL(o, 2, "%s is not sorted by A read", las_list[part].c_str());
        if (lrc == -3) PART_FAIL(o, 1, "%s: a read name without \"/id/\" or an id outside the FASTA (the reference crashes here)", las_list[part].c_str());
        if (lrc != 0) PART_FAIL(o, -1, "Reads_filter: cannot read %s", las_list[part].c_str());
        if (las.novl == 0) PART_FAIL(o, 1, "No alignments!");
        o.r_begin = las.r_begin; o.r_end = las.r_end;
        PART_CHECK(o, cx, hinge_set_pileups_packed(cx, las.r_begin, las.r_end, las.n_kept(), las.row_ptr.data(), las.a_span.data(), las.b_span.data(),
                                                   las.b_flag.data(), las.span16_ptr(), las.max_pile, las.spans_in_range ? 1 : 0, 0));
        if (P.reso == 40 && las.nbins40.size() == (size_t)(las.r_end - las.r_begin + 1))   // the ingest's per-read bin counts: the one-sweep pass launches no sweep before it
            PART_CHECK(o, cx, hinge_set_pile_bins(cx, 40, las.nbins40.data(), 0));
        PART_CHECK(o, cx, hinge_filter_coverage_out(cx, 1));   // K2 also stores the cutoff-0 bins: .coverage.txt needs no sweep of its own
        if (P.delete_telomere) {   // self_match_reads, filter.cpp:552-561 (float accumulation in record order)
            std::map<int, float> cov;
            for (size_t k = 0; k < las.self_a.size(); k++) {
                float& c = cov[las.self_a[k]];
                c += las.self_span[4 * k + 1] - las.self_span[4 * k];
                c += las.self_span[4 * k + 3] - las.self_span[4 * k + 2];
            }
            for (auto& it : cov) {
                float c = it.second / float(db.rlen[(size_t)it.first]);
                if ((c > 4.5) && (db.rlen[(size_t)it.first] > 10000)) o.self_match.insert(it.first);
            }
        }
        if (!reads_to_keep.empty()) {   // + every B the listed reads have an alignment with (idx_ab, filter.cpp:680-694); cumulative over parts
            const std::set<int> initial = reads_to_keep;
            for (int i : initial) {
                if (i < 0 || i >= n_read) continue;
                for (int64_t j = las.rec_row_ptr[(size_t)i]; j < las.rec_row_ptr[(size_t)i + 1]; j++) reads_to_keep.insert(las.rec_b[(size_t)j]);
            }
            console.info("After accounting for neighbours of reads selected, have %zu reads", reads_to_keep.size());
            std::vector<uint8_t> keep((size_t)n_read, 0);
            for (int i : reads_to_keep) if (i >= 0 && i < n_read) keep[(size_t)i] = 1;
            PART_CHECK(o, cx, hinge_set_read_restriction(cx, keep.data()));
        }
        PART_CHECK(o, cx, hinge_filter_stats_median(cx, &P, nullptr, &o.est));   // K1 + the part's own median (filter.cpp:642-678), one launch
    };
    // phase B: the mask / annotation pass under the MIN_COV this part sees; its mask rows come back to the host
    auto phase_b = [&](hinge_ctx* cx, int min_cov_seen, PartOut& o) {
        const size_t nr = (size_t)(o.r_end - o.r_begin + 1);
        PART_CHECK(o, cx, hinge_filter_set_min_cov(cx, min_cov_seen));
        PART_CHECK(o, cx, hinge_filter_mask_annotate(cx, &P));
        o.mask.resize(2 * nr); o.cmask.resize(2 * nr); o.flags.resize(nr);
        PART_CHECK(o, cx, hinge_filter_get_masks(cx, o.mask.data(), o.cmask.data(), o.flags.data()));
    };
    // phase C: hinge calling (every mask this part may see is in the context's table by now) and the rest of the results
    auto phase_c = [&](hinge_ctx* cx, PartOut& o) {
        const size_t nr = (size_t)(o.r_end - o.r_begin + 1);
        PART_CHECK(o, cx, hinge_filter_hinges(cx, &P));
        o.coff.resize(nr + 1); o.nb.resize(nr);
        PART_CHECK(o, cx, hinge_filter_get_coverage(cx, o.coff.data(), nullptr, nullptr, 0));
        o.cov.resize((size_t)std::max<int64_t>(o.coff[nr], 1));   // filled by the copy from the device: no zero fill, huge pages
        PART_CHECK(o, cx, hinge_filter_get_coverage(cx, o.coff.data(), o.nb.data(), o.cov.data(), o.coff[nr]));
        o.off.resize(nr + 1);
        PART_CHECK(o, cx, hinge_filter_get_annotations(cx, o.off.data(), nullptr, nullptr, nullptr));
        const size_t na = (size_t)std::max<int64_t>(o.off[nr], 1);
        o.pos.resize(na); o.type.resize(na); o.is_hinge.resize(na);
        PART_CHECK(o, cx, hinge_filter_get_annotations(cx, o.off.data(), o.pos.data(), o.type.data(), o.is_hinge.data()));
    };
    // the text of one part, in the reference's order (filter.cpp:599-602,775-788,1076-1098)
    auto write_part = [&](size_t part, PartOut& o) {
        console.info("part: %zu  name of las: %s", part, las_list[part].c_str());
        console.info("# Alignments: %lld", (long long)o.las->novl);
        console.info("Estimated mean coverage: %lld", (long long)(o.est.num_slot ? o.est.total_cov / o.est.num_slot : 0));
        console.info("Estimated median coverage: %d", P.est_cov != 0 ? P.est_cov : o.est.cov_est);
        const int r_begin = o.r_begin;
        const size_t nr = (size_t)(o.r_end - o.r_begin + 1);
        // .coverage.txt (150 MB of text for an E. coli part) is formatted and written by its own thread (which fans out to the
        // host threads) while this one writes the small files
        std::thread cov_writer([&] { write_coverage_txt(f_cov, r_begin, o.nb, o.cov, P.reso, &o.coff); });
        for (size_t k = 0; k < nr; k++) {
            const int i = r_begin + (int)k;
            if (P.delete_telomere) {
                if (o.flags[k] & 1) fprintf(f_covflag, "%d\n", i);
                if (o.self_match.count(i)) fprintf(f_selfflag, "%d\n", i);
            }
            fprintf(f_cmask, "%d %d %d\n", i, o.cmask[2 * k], o.cmask[2 * k + 1]);
            fprintf(f_mask, "%d %d %d\n", i, o.mask[2 * k], o.mask[2 * k + 1]);
        }
        fclose(fopen("debug.txt", "w"));   // filter.cpp:837
        if (f_rep) {   // closed inside the part loop, filter.cpp:1086: later parts write nothing
            for (size_t k = 0; k < nr; k++) {
                fprintf(f_rep, "%d ", r_begin + (int)k);
                for (int64_t t = o.off[k]; t < o.off[k + 1]; t++) fprintf(f_rep, "%d %d ", o.pos[(size_t)t], o.type[(size_t)t]);
                fprintf(f_rep, "\n");
            }
            fclose(f_rep);
            f_rep = nullptr;
        }
        int hg_cnt = 0;
        for (size_t k = 0; k + 1 < nr; k++) {   // i < r_end, filter.cpp:1091
            fprintf(f_hg, "%d ", r_begin + (int)k);
            for (int64_t t = o.off[k]; t < o.off[k + 1]; t++)
                if (o.is_hinge[(size_t)t]) { fprintf(f_hg, "%d %d ", o.pos[(size_t)t], o.type[(size_t)t]); hg_cnt++; }
            fprintf(f_hg, "\n");
        }
        tm.mark("mas/cmas/repeat/hinges txt");
        cov_writer.join();
        tm.mark("coverage.txt (rest)");
        console.info("Number of hinges before filtering: %lld", (long long)o.off[nr]);
        console.info("Number of hinges: %d", hg_cnt);
    };
    auto report = [&](const PartOut& o) -> int {   // a failed part ends the program the way the sequential loop did
        if (o.code == 0) return 0;
        if (o.code == -1) { fprintf(stderr, "%s\n", o.error.c_str()); quit(1); }
        console.error("%s", o.error.c_str());
        return o.code;
    };
    auto seen_min_cov = [&](const PartOut& o) {   // filter.cpp:671-678
        const int cov_est = P.est_cov != 0 ? P.est_cov : o.est.cov_est;
        if (running_min_cov < cov_est / 3) running_min_cov = cov_est / 3;
        return running_min_cov;
    };

    for (size_t w0 = 0; w0 < las_list.size(); w0 += (size_t)n_ranks) {
        const size_t w1 = std::min(las_list.size(), w0 + (size_t)n_ranks);
        const size_t nw = w1 - w0;
        // (on the heap, and the LAST wave's is never freed: the process leaves through _exit() right after its text is written, and
        // unmapping a part's columns, bins and .las first costs ~70 ms of a 0.4-s run)
        std::vector<PartOut>* outs_heap = new std::vector<PartOut>(nw);
        std::vector<PartOut>& outs = *outs_heap;
        auto on_ranks = [&](const std::function<void(size_t)>& f) {   // f(k) for the parts of the wave, rank k on its own thread
            if (nw == 1) { f(0); return; }
            std::vector<std::thread> th;
            for (size_t k = 0; k < nw; k++) th.emplace_back(f, k);
            for (auto& t : th) t.join();
        };
        on_ranks([&](size_t k) { phase_a(ctxs[k], w0 + k, outs[k]); });
        tm.mark("wave: ingest + H2D + statistics + median");
        // exchange 1: the MIN_COV each part sees is a running maximum in part order (8 bytes per part; in one process: a host loop)
        std::vector<int> seen(nw, running_min_cov);
        size_t n_good = 0;
        for (; n_good < nw && outs[n_good].code == 0; n_good++) seen[n_good] = seen_min_cov(outs[n_good]);
        on_ranks([&](size_t k) { if (k < n_good) phase_b(ctxs[k], seen[k], outs[k]); });
        for (size_t k = 0; k < n_good; k++) if (outs[k].code != 0) { n_good = k; break; }
        // exchange 2: mask rows.  While part p is in hinge calling, the table holds the masks of the parts up to p and (0, 0) for
        // the later ones (filter.cpp:534 / :778-787 in a sequential loop); afterwards every context gets the rest of the wave.
        // Ranks on distinct GPUs: ONE ncclAllGather over xGMI per wave (hinge_comm_exchange_mask_rows); ranks that share a device
        // (HINGE_RANKS on a small box, the tests) or a ragged / failed wave: through the host.
        const bool wave_rccl = use_rccl && nw == (size_t)n_ranks && n_good == nw;
        std::vector<int32_t> row_lo(nw, 0), row_hi(nw, -1);
        for (size_t k = 0; k < n_good; k++) { row_lo[k] = outs[k].r_begin; row_hi[k] = outs[k].r_end; }
        auto rccl_rows = [&](int phase) {
            if (hinge_comm_exchange_mask_rows(ctxs.data(), n_ranks, row_lo.data(), row_hi.data(), phase) == HINGE_OK) return true;
            console.error("mask rows over RCCL: %s", hinge_last_error(ctxs[0]));
            return false;
        };
        if (n_ranks > 1) {
            if (wave_rccl) { if (!rccl_rows(0)) return 2; }
            else on_ranks([&](size_t k) {
                if (k >= n_good) return;
                for (size_t q = 0; q < k; q++)
                    PART_CHECK(outs[k], ctxs[k], hinge_set_mask_rows(ctxs[k], outs[q].r_begin, outs[q].r_end, outs[q].mask.data()));
            });
        }
        on_ranks([&](size_t k) { if (k < n_good && outs[k].code == 0) phase_c(ctxs[k], outs[k]); });
        if (n_ranks > 1) {
            if (wave_rccl) { if (!rccl_rows(1)) return 2; }
            else on_ranks([&](size_t k) {
                for (size_t q = k + 1; q < n_good; q++)
                    if (outs[k].code == 0) PART_CHECK(outs[k], ctxs[k], hinge_set_mask_rows(ctxs[k], outs[q].r_begin, outs[q].r_end, outs[q].mask.data()));
            });   // (ranks without a part in this - the last - wave are not used again)
        }
        tm.mark("wave: masks, exchange, hinges, results");
        for (size_t k = 0; k < nw; k++) {
            const int rc = report(outs[k]);
            if (rc) return rc;
            write_part(w0 + k, outs[k]);
        }
        if (w1 < las_list.size() || (getenv("HINGE_SLOW_EXIT") && !loader.shared)) delete outs_heap;
    }
    if (f_rep) fclose(f_rep);
    fclose(f_cov); fclose(f_hg); fclose(f_mask); fclose(f_cmask); fclose(f_covflag); fclose(f_selfflag);
    return finish(ctx, tm);
}
