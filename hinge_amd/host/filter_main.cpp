// Reads_filter  ==  `hinge filter --db DB --las LAS[.las] [--mlas] -x PREFIX --config nominal.ini`
// Same flags, same inputs, same output files, same exit codes as src/filter/filter.cpp; the pile-up
// arithmetic (filter.cpp:529-1070) runs in the HIP kernels behind include/hinge_hip.h.
#include <set>
#include "host_common.h"

using namespace hh;

int main(int argc, char* argv[]) {
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
    PartLoader loader;
    loader.pairs = !reads_to_keep.empty();   // the neighbours of the listed reads need the per-record B column
    loader.paf = fa_and_paf;
    loader.span16 = true;
    if (!las_list.empty()) loader.preload(las_list[0], db.rlen);
    tm.mark("las ingest (part 1) || HIP init");
    if (gpu.join() != HINGE_OK) { console.error("no usable MI355X / HIP device: this build has no CPU path"); return 2; }
    hinge_ctx* ctx = gpu.ctx;
    HH_CHECK(ctx, hinge_set_reads(ctx, n_read, db.rlen.data(), has_qv ? qvm.data() : nullptr));
    HH_CHECK(ctx, hinge_filter_set_min_cov(ctx, P.min_cov));

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

    for (size_t part = 0; part < las_list.size(); part++) {
        console.info("part: %zu  name of las: %s", part, las_list[part].c_str());
        int lrc = 0;
        std::unique_ptr<LasPart> las_owner(loader.take(part, las_list[part], db.rlen, lrc));
        LasPart& las = *las_owner;
        if (lrc == -2) { console.error("%s is not sorted by A read", las_list[part].c_str()); return 2; }
        if (lrc == -3) { console.error("%s: a read name without \"/id/\" or an id outside the FASTA (the reference crashes here)", las_list[part].c_str()); return 1; }
        if (lrc != 0) { fprintf(stderr, "Reads_filter: cannot read %s\n", las_list[part].c_str()); quit(1); }
        tm.mark("las ingest");
        console.info("# Alignments: %lld", (long long)las.novl);
        if (las.novl == 0) { console.error("No alignments!"); return 1; }
        const int r_begin = las.r_begin, r_end = las.r_end;
        const size_t nr = (size_t)(r_end - r_begin + 1);
        HH_CHECK(ctx, hinge_set_pileups_packed(ctx, r_begin, r_end, las.n_kept(), las.row_ptr.data(), las.a_span.data(), las.b_span.data(), las.b_flag.data(),
                                               las.span16_ptr(), las.max_pile, las.spans_in_range ? 1 : 0, 0));
        HH_CHECK(ctx, hinge_filter_coverage_out(ctx, 1));   // K2 also stores the cutoff-0 bins: .coverage.txt needs no sweep of its own

        tm.mark("set_pileups (H2D)");
        // self_match_reads, filter.cpp:552-561 (float accumulation in record order)
        std::set<int> self_match;
        if (P.delete_telomere) {
            std::map<int, float> cov;
            for (size_t k = 0; k < las.self_a.size(); k++) {
                float& c = cov[las.self_a[k]];
                c += las.self_span[4 * k + 1] - las.self_span[4 * k];
                c += las.self_span[4 * k + 3] - las.self_span[4 * k + 2];
            }
            for (auto& it : cov) {
                float c = it.second / float(db.rlen[(size_t)it.first]);
                if ((c > 4.5) && (db.rlen[(size_t)it.first] > 10000)) self_match.insert(it.first);
            }
        }

        tm.mark("self matches");
        if (!reads_to_keep.empty()) {   // + every B the listed reads have an alignment with (idx_ab, filter.cpp:680-694); cumulative over parts
            const std::set<int> initial = reads_to_keep;
            for (int i : initial) {
                if (i < 0 || i >= n_read) continue;
                for (int64_t j = las.rec_row_ptr[(size_t)i]; j < las.rec_row_ptr[(size_t)i + 1]; j++) reads_to_keep.insert(las.rec_b[(size_t)j]);
            }
            console.info("After accounting for neighbours of reads selected, have %zu reads", reads_to_keep.size());
            std::vector<uint8_t> keep((size_t)n_read, 0);
            for (int i : reads_to_keep) if (i >= 0 && i < n_read) keep[(size_t)i] = 1;
            HH_CHECK(ctx, hinge_set_read_restriction(ctx, keep.data()));
        }
        hinge_cov_estimate est;
        HH_CHECK(ctx, hinge_filter_stats(ctx, &P));
        HH_CHECK(ctx, hinge_filter_median(ctx, &P, r_begin, r_end, &est));
        console.info("Estimated mean coverage: %lld", (long long)(est.num_slot ? est.total_cov / est.num_slot : 0));
        console.info("Estimated median coverage: %d", P.est_cov != 0 ? P.est_cov : est.cov_est);
        HH_CHECK(ctx, hinge_filter_mask_annotate(ctx, &P));
        HH_CHECK(ctx, hinge_filter_hinges(ctx, &P));

        tm.mark("kernels (4 passes)");
        // .coverage.txt, filter.cpp:599-602: the bins K2 stored
        {
            std::vector<int64_t> coff(nr + 1);
            HH_CHECK(ctx, hinge_filter_get_coverage(ctx, coff.data(), nullptr, nullptr, 0));
            std::vector<int32_t> nb(nr);
            UVec<int32_t> cov;   // filled by the copy from the device: no zero fill, huge pages
            cov.resize((size_t)std::max<int64_t>(coff[nr], 1));
            HH_CHECK(ctx, hinge_filter_get_coverage(ctx, coff.data(), nb.data(), cov.data(), coff[nr]));
            write_coverage_txt(f_cov, r_begin, nb, cov, P.reso, &coff);
        }
        tm.mark("coverage.txt");
        std::vector<int32_t> mask(2 * nr), cmask(2 * nr);
        std::vector<uint8_t> flags(nr);
        HH_CHECK(ctx, hinge_filter_get_masks(ctx, mask.data(), cmask.data(), flags.data()));
        for (size_t k = 0; k < nr; k++) {
            const int i = r_begin + (int)k;
            if (P.delete_telomere) {
                if (flags[k] & 1) fprintf(f_covflag, "%d\n", i);
                if (self_match.count(i)) fprintf(f_selfflag, "%d\n", i);
            }
            fprintf(f_cmask, "%d %d %d\n", i, cmask[2 * k], cmask[2 * k + 1]);
            fprintf(f_mask, "%d %d %d\n", i, mask[2 * k], mask[2 * k + 1]);
        }
        fclose(fopen("debug.txt", "w"));   // filter.cpp:837

        std::vector<int64_t> off(nr + 1);
        HH_CHECK(ctx, hinge_filter_get_annotations(ctx, off.data(), nullptr, nullptr, nullptr));
        const size_t na = (size_t)std::max<int64_t>(off[nr], 1);
        std::vector<int32_t> pos(na), type(na);
        std::vector<uint8_t> is_hinge(na);
        HH_CHECK(ctx, hinge_filter_get_annotations(ctx, off.data(), pos.data(), type.data(), is_hinge.data()));
        if (f_rep) {   // closed inside the part loop, filter.cpp:1086: later parts write nothing
            for (size_t k = 0; k < nr; k++) {
                fprintf(f_rep, "%d ", r_begin + (int)k);
                for (int64_t t = off[k]; t < off[k + 1]; t++) fprintf(f_rep, "%d %d ", pos[(size_t)t], type[(size_t)t]);
                fprintf(f_rep, "\n");
            }
            fclose(f_rep);
            f_rep = nullptr;
        }
        int hg_cnt = 0;
        for (size_t k = 0; k + 1 < nr; k++) {   // i < r_end, filter.cpp:1091
            fprintf(f_hg, "%d ", r_begin + (int)k);
            for (int64_t t = off[k]; t < off[k + 1]; t++)
                if (is_hinge[(size_t)t]) { fprintf(f_hg, "%d %d ", pos[(size_t)t], type[(size_t)t]); hg_cnt++; }
            fprintf(f_hg, "\n");
        }
        tm.mark("mas/cmas/repeat/hinges txt");
        console.info("Number of hinges before filtering: %lld", (long long)off[nr]);
        console.info("Number of hinges: %d", hg_cnt);
    }
    if (f_rep) fclose(f_rep);
    fclose(f_cov); fclose(f_hg); fclose(f_mask); fclose(f_cmask); fclose(f_covflag); fclose(f_selfflag);
    return finish(ctx, tm);
}
