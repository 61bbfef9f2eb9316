// consensus  ==  `hinge consensus DRAFT_DB READ_DB LAS OUT.fasta nominal.ini`
// Same five positional arguments, input files, output FASTA and stdout text as src/consensus/consensus.cpp:77-288
// (demo/ecoli_demo/run.sh:38-42).  Host side: the two DAZZ_DBs with their bases, the draft-vs-reads .las, the per-contig
// std::sort(compare_overlap_aln) and remove_multialign's count (consensus.cpp:126-150), the text.  Everything base-level -
// realignment between trace points, the gapped columns, chop_end, the vote and the calls - runs behind the C ABI
// (hinge_consensus_*, include/hinge_hip.h) on the GPU.
#include "host_common.h"

#include <unordered_set>

using namespace hh;

int main(int argc, char* argv[]) {
    if (argc < 6) {
        fprintf(stderr, "usage: consensus <draft db> <read db> <las> <out.fasta> <nominal.ini>\n");
        return 1;
    }
    PhaseTimer tm("consensus");
    CtxInit gpu;
    gpu.start();
    const std::string name_db1 = argv[1], name_db2 = argv[2], name_las = argv[3], name_out = argv[4], name_config = argv[5];
    FILE* out = fopen(name_out.c_str(), "w");      // (std::ofstream out(name_out) is the program's first statement)
    Config ini(name_config);
    if (ini.error < 0) { printf("Can't load %s\n", name_config.c_str()); fflush(stdout); gpu.join(); return 1; }
    const int LENGTH_THRESHOLD = (int)ini.get_int("consensus", "min_length", -1);
    printf("length threshold:%d\n", LENGTH_THRESHOLD);

    ReadDB db1, db2;
    if (db1.open(name_db1) != 0 || db2.open(name_db2) != 0) { fprintf(stderr, "consensus: Could not open database\n"); quit(1); }
    printf("%d files\n%d files\n", db1.nfiles, db2.nfiles);
    Mapped bps1, bps2;
    const bool has1 = bps1.open(db1.dir + "/." + db1.root + ".bps"), has2 = bps2.open(db2.dir + "/." + db2.root + ".bps");
    if ((!has1 && !db1.rlen.empty()) || (!has2 && !db2.rlen.empty())) { fprintf(stderr, "consensus: cannot read the .bps file of a database\n"); quit(1); }
    const int n_contigs = (int)db1.rlen.size(), n_reads = (int)db2.rlen.size();
    printf("# Contigs:%d\n# Reads:%d\n", n_contigs, n_reads);

    // ---- the .las: every record whose A read passes getAlignment(res, 0, n_alns)'s range filter (A reads 1 .. n_alns, 1-based) ----
    Mapped las;
    if (!las.open(name_las) || las.n < 12) { fprintf(stderr, "consensus: cannot read %s\n", name_las.c_str()); quit(1); }
    const int64_t novl = rd<int64_t>(las.p);
    const int tspace = rd<int32_t>(las.p + 8);
    const int tbytes = tspace <= 125 ? 1 : 2;
    printf("# Alignments:%d\n", (int)novl);
    std::vector<hinge_cns_alignment> recs;
    std::vector<uint16_t> trace;
    {
        size_t p = 12;
        for (int64_t j = 0; j < novl; j++) {
            if (p + 40 > las.n) break;
            const uint8_t* r = las.p + p;
            const int tlen = rd<int32_t>(r);
            const size_t tb = (size_t)std::max(tlen, 0) * (size_t)tbytes;
            if (tlen < 0 || p + 40 + tb > las.n) break;
            p += 40 + tb;
            hinge_cns_alignment a;
            a.aread = rd<int32_t>(r + 28); a.bread = rd<int32_t>(r + 32);
            if (!(a.aread + 1 >= 1 && a.aread + 1 <= novl)) continue;
            if (a.aread >= n_contigs || a.bread < 0 || a.bread >= n_reads) { fprintf(stderr, "consensus: alignment %lld names a read outside its database\n", (long long)j); quit(1); }
            a.comp = (int)(rd<uint32_t>(r + 24) & 1u);
            a.abpos = rd<int32_t>(r + 8); a.bbpos = rd<int32_t>(r + 12); a.aepos = rd<int32_t>(r + 16); a.bepos = rd<int32_t>(r + 20);
            a.tlen = tlen;
            a.trace_off = (int64_t)trace.size();
            const uint8_t* t = r + 40;
            if (tbytes == 1) for (int k = 0; k < tlen; k++) trace.push_back(t[k]);
            else for (int k = 0; k < tlen; k++) trace.push_back((uint16_t)(t[2 * k] | (t[2 * k + 1] << 8)));
            recs.push_back(a);
        }
    }
    printf("%lu\n", (unsigned long)recs.size());
    tm.mark("ingest");

    // ---- per contig: libstdc++'s order of std::sort(compare_overlap_aln) and how many of its first alignments are used ----
    std::vector<std::vector<int>> idx((size_t)n_contigs);
    for (size_t i = 0; i < recs.size(); i++) idx[(size_t)recs[i].aread].push_back((int)i);
    std::vector<int> seq_count((size_t)n_contigs, 0);
    std::vector<hinge_cns_alignment> used;        // contig by contig, in the order of use
    std::vector<int64_t> first_used((size_t)n_contigs + 1, 0);
    for (int i = 0; i < n_contigs; i++) {
        std::vector<int>& v = idx[(size_t)i];
        const int n = (int)v.size();
        if (n > 1) {
            std::vector<int64_t> key((size_t)n);
            std::vector<int32_t> perm((size_t)n);
            for (int k = 0; k < n; k++) { const hinge_cns_alignment& a = recs[(size_t)v[(size_t)k]]; key[(size_t)k] = (int64_t)(a.aepos - a.abpos) + (a.bepos - a.bbpos); }
            if (hinge_sort_order_desc(n, key.data(), 1, perm.data()) != HINGE_OK) { fprintf(stderr, "consensus: sort order\n"); quit(2); }
            std::vector<int> sorted((size_t)n);
            for (int k = 0; k < n; k++) sorted[(size_t)k] = v[(size_t)perm[(size_t)k]];
            v.swap(sorted);
        }
        printf("%d %lu\n", i, (unsigned long)v.size());
        // remove_multialign works on a COPY of the vector (consensus.cpp:62): only its count reaches main()
        std::unordered_set<int> seen_b;
        int r = 0;
        for (int k = 0; k < n; k++) {
            const hinge_cns_alignment& a = recs[(size_t)v[(size_t)k]];
            if (a.aepos - a.abpos >= LENGTH_THRESHOLD && seen_b.insert(a.bread).second) r++;
        }
        seq_count[(size_t)i] = r;
        first_used[(size_t)i] = (int64_t)used.size();
        for (int k = 0; k < r; k++) used.push_back(recs[(size_t)v[(size_t)k]]);
    }
    first_used[(size_t)n_contigs] = (int64_t)used.size();
    printf("Getting read lengths\n");
    for (int i = 0; i < n_contigs; i++) printf("%d\t%lu\n", i, (unsigned long)db1.rlen[(size_t)i]);
    printf("Building consensus sequences...\n");
    tm.mark("select");

    // ---- the GPU: realign, vote, call ----
    if (gpu.join() != HINGE_OK) { fprintf(stderr, "consensus: no usable GPU (%s)\n", gpu.ctx ? hinge_last_error(gpu.ctx) : "hinge_ctx_create failed"); quit(2); }
    hinge_ctx* ctx = gpu.ctx;
    tm.mark("hip init");
    auto die = [&](const char* what) { fprintf(stderr, "consensus: %s: %s\n", what, hinge_last_error(ctx)); quit(2); };
    if (hinge_consensus_set_db(ctx, 0, n_contigs, db1.rlen.data(), db1.boff.data(), bps1.p, (int64_t)bps1.n) != HINGE_OK) die("draft DB");
    if (hinge_consensus_set_db(ctx, 1, n_reads, db2.rlen.data(), db2.boff.data(), bps2.p, (int64_t)bps2.n) != HINGE_OK) die("read DB");
    tm.mark("H2D bases");
    // Contigs go to the GPU in batches: one hinge_consensus_run holds an indel slot per (segment, recorded difference) - 240 MB for an
    // E. coli-sized draft at 30x, beyond 2^32 slots for a few hundred Mb - so a batch ends where its estimate reaches
    // HINGE_CNS_SLOT_BUDGET (default 1.5e9 slots = 6 GB); a batch is whole contigs, their results are kept, the text is written at the end
    // in contig order.  (A contig without voting alignments comes out of every run as its own lower-case bases.)
    long long budget = 1500000000ll;
    if (const char* g = getenv("HINGE_CNS_SLOT_BUDGET")) budget = std::max(1ll, atoll(g));
    auto slots_of = [&](const hinge_cns_alignment& a) {
        int dmax = 0;
        for (int d = 0; d < a.tlen; d += 2) dmax = std::max(dmax, (int)trace[(size_t)a.trace_off + (size_t)d]);
        return (long long)std::max(a.tlen / 2, 1) * (dmax + 16);
    };
    std::vector<std::string> text((size_t)n_contigs);
    std::vector<hinge_cns_stats> stats((size_t)n_contigs);
    std::vector<int32_t> offsets(std::max<size_t>(used.size(), 1));
    int n_batches = 0;
    for (int c0 = 0; c0 < n_contigs;) {
        int c1 = c0;
        long long est = 0;
        do {
            for (int64_t k = first_used[(size_t)c1]; k < first_used[(size_t)c1 + 1]; k++) est += slots_of(used[(size_t)k]);
            c1++;
        } while (c1 < n_contigs && est < budget);
        const int64_t k0 = first_used[(size_t)c0], k1 = first_used[(size_t)c1];
        if (hinge_consensus_run(ctx, k1 - k0, used.data() + k0, trace.data(), (int64_t)trace.size(), tspace) != HINGE_OK) die("consensus");
        if (k1 > k0 && hinge_consensus_get_offsets(ctx, offsets.data() + k0) != HINGE_OK) die("offsets");
        for (int i = c0; i < c1; i++) {
            int64_t len = 0;
            if (hinge_consensus_get_contig(ctx, i, nullptr, 0, &len, &stats[(size_t)i]) != HINGE_OK) die("contig");
            text[(size_t)i].resize((size_t)len);
            if (len && hinge_consensus_get_contig(ctx, i, &text[(size_t)i][0], len, &len, &stats[(size_t)i]) != HINGE_OK) die("contig");
        }
        n_batches++;
        c0 = c1;
    }
    tm.mark("realign + vote + call");

    for (int i = 0; i < n_contigs; i++) {
        printf("Contig %d: %d reads\n", i, seq_count[(size_t)i]);
        const hinge_cns_stats& st = stats[(size_t)i];
        fprintf(out, ">Consensus%d\n", i);
        fwrite(text[(size_t)i].data(), 1, text[(size_t)i].size(), out);
        fputc('\n', out);
        if (seq_count[(size_t)i] == 0) continue;      // (printed as it is, consensus.cpp:158-162: no statistics)
        for (int64_t k = first_used[(size_t)i]; k < first_used[(size_t)i + 1]; k++) printf("%d\n", offsets[(size_t)k]);
        const int alen = st.contig_length;
        printf("Average coverage: %f\n", (1.0 * st.sum_coverage) / alen);
        printf("Good bases: %d/%d\n", st.good_bases, alen);
        printf("Insertions: %d/%d\n", st.insertions, alen);
        printf("Deletions: %d/%d\n", st.deletions, alen);
        printf("Low coverage bases: %d/%d\n", st.low_coverage_bases, alen);
        printf("Consensus length: %d\n", st.consensus_length);
    }
    if (getenv("HINGE_HOST_TIMING")) fprintf(stderr, "[timing] consensus: %d contig batch(es)\n", n_batches);
    fclose(out);
    tm.mark("text");
    return finish(ctx, tm, 0);
}
