// TEST INFRASTRUCTURE - NOT PRODUCT CODE.
// CPU restatement ("oracle") of `hinge consensus`: /root/reference/src/consensus/consensus.cpp:77-288 and
// the LAInterface functions under it.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
//
// PINNED: unlike the three graph stages this one needs neither spdlog nor Boost, so oracle/Makefile builds the reference's
// own consensus.cpp + library sources UNMODIFIED into oracle/_ref/consensus; tests/test_consensus_oracle.py holds this
// restatement against that binary (FASTA and stdout, byte for byte) live where the reference tree exists and against the
// golden files it produced (tests/golden/consensus_*.fasta) everywhere.
//
// What is restated, and from where:
//   * DB stub / index / 2-bit bases, trimming       lib/DB.c:395-683 (Open_DB, Trim_DB), :1380-1500 (Load_Read / Load_Subread)
//   * .las records, A-read range filter             lib/LAInterface.cpp:1753-2135 (getAlignment, LAlignment form)
//   * per-contig sort, remove_multialign quirk      consensus.cpp:62-75,126-150,4925-4927 (the vector is passed BY VALUE, so only
//                                                   its return value - a count - reaches main(): the first `count` alignments
//                                                   of the SORTED list are used, whatever their length or B read)
//   * recoverAlignment -> computeTracePTS -> iter_np lib/LAInterface.cpp:4125-4244, :3410-3506, :3152-3404 (Myers' O(np) wave
//                                                   algorithm per trace-point segment, trace-back with re-sliding)
//   * getAlignmentTags                              lib/LAInterface.cpp:3709-3905 (gapped strings from the indel list)
//   * chop_end, the column vote, the base calls     consensus.cpp:27-45,163-283
#include "cns_core.h"

extern "C" {

long oracle_consensus_neg_slides() { return g_neg_slides; }

// 0 ok; 1 = unreadable config (the reference's "return 1"); -1 unreadable DB / .las (reference exit(1)); -3 = ids out of range.
// log_path (may be NULL): what the reference prints on stdout.  dump_path (may be NULL): per USED alignment, in the order of
// use, int32 {contig, position in the .las, chop offset, n} + n int32 indel-list entries (test aid for the device kernels).
int oracle_consensus(const char* name_db1, const char* name_db2, const char* name_las, const char* name_out, const char* name_config,
                     const char* log_path, const char* dump_path) {
    Out o;
    if (log_path) o.log = fopen(log_path, "w");
    FILE* dump = dump_path ? fopen(dump_path, "wb") : nullptr;
    struct Closer { Out& o; FILE* d; ~Closer() { if (o.log) fclose(o.log); if (d) fclose(d); } } closer{o, dump};
    FILE* out = fopen(name_out, "w");       // (std::ofstream out(name_out) comes first, consensus.cpp:85)
    Ini ini(name_config);
    if (ini.error < 0) { o.pf("Can't load %s\n", name_config); if (out) fclose(out); return 1; }
    const int LENGTH_THRESHOLD = (int)ini.get_int("consensus", "min_length", -1);
    o.pf("length threshold:%d\n", LENGTH_THRESHOLD);
    SeqDB d1, d2;
    if (open_seq_db(name_db1, d1) != 0 || open_seq_db(name_db2, d2) != 0) { if (out) fclose(out); return -1; }
    o.pf("%d files\n%d files\n", d1.nfiles, d2.nfiles);
    const int n_contigs = (int)d1.rlen.size(), n_reads = (int)d2.rlen.size();
    o.pf("# Contigs:%d\n# Reads:%d\n", n_contigs, n_reads);
    std::vector<Aln> res;
    int tspace = 0; int64_t n_alns = 0;
    {
        const int rc = load_alignments(name_las, d1, d2, res, tspace, n_alns);
        if (rc != 0) { if (out) fclose(out); return rc; }
    }
    o.pf("# Alignments:%d\n", (int)n_alns);
    o.pf("%lu\n", (unsigned long)res.size());
    std::vector<std::vector<Aln*>> idx((size_t)n_contigs);
    std::vector<int> las_pos(res.size());
    for (size_t i = 0; i < res.size(); i++) idx[(size_t)res[i].a].push_back(&res[i]);
    for (int i = 0; i < n_contigs; i++) {
        std::sort(idx[(size_t)i].begin(), idx[(size_t)i].end(), by_aligned_length);   // (libstdc++'s introsort: the tie order is part of the result)
        o.pf("%d %lu\n", i, (unsigned long)idx[(size_t)i].size());
    }
    o.pf("Getting read lengths\n");
    for (int i = 0; i < n_contigs; i++) o.pf("%d\t%lu\n", i, (unsigned long)d1.rlen[(size_t)i]);
    o.pf("Building consensus sequences...\n");
    static const char lower[4] = {'a', 'c', 'g', 't'}, upper[4] = {'A', 'C', 'G', 'T'};
    Waves w;
    std::string ra, rb;
    for (int i = 0; i < n_contigs; i++) {
        std::vector<Aln*>& v = idx[(size_t)i];
        // remove_multialign on a COPY of the vector: only the count comes back (consensus.cpp:62-75,148)
        int seq_count = 0;
        {
            std::vector<Aln*> c = v;
            for (size_t x = 0; x < c.size(); x++)
                if (c[x]->ae - c[x]->ab >= LENGTH_THRESHOLD) {
                    int y;
                    for (y = 0; y < seq_count; y++) if (c[(size_t)y]->b == c[x]->b) break;
                    if (y == seq_count) c[(size_t)seq_count++] = c[x];
                }
        }
        o.pf("Contig %d: %d reads\n", i, seq_count);
        const int alen = d1.rlen[(size_t)i];
        if (seq_count == 0) {
            fprintf(out, ">Consensus%d\n", i);
            for (int j = 0; j < alen; j++) fputc(lower[d1.base(i, j)], out);   // (Load_Read with ascii = 1: lower case)
            fputc('\n', out);
            continue;
        }
        std::vector<int> score((size_t)alen * 5, 0), ins_score((size_t)alen, 0), ins_base((size_t)alen * 5, 0), depth((size_t)alen, 0);
        for (int j = 0; j < seq_count; j++) {
            Aln& al = *v[(size_t)j];
            recover(d1, d2, al, tspace, w);
            tags(d1, d2, al, ra, rb);
            const int offset = chop_end(ra, rb, 100);
            o.pf("%d\n", offset);
            if (dump) {
                const int32_t hd[4] = {i, (int32_t)(&al - res.data()), offset, (int32_t)al.trace.size()};
                fwrite(hd, 4, 4, dump);
                if (!al.trace.empty()) fwrite(al.trace.data(), 4, al.trace.size(), dump);
            }
            int pos = al.ab + offset;
            for (size_t m = 0; m < ra.size(); m++) {
                int base = -1;
                switch (rb[m]) { case 'A': base = 0; break; case 'C': base = 1; break; case 'G': base = 2; break; case 'T': base = 3; break; case '-': base = 4; break; }
                if (ra[m] != '-') {
                    if (base != -1) { score[(size_t)pos * 5 + (size_t)base]++; depth[(size_t)pos]++; }
                    pos++;
                } else if (base != -1) {
                    ins_score[(size_t)pos]++; ins_base[(size_t)pos * 5 + (size_t)base]++;
                }
            }
        }
        int good = 0, insertions = 0, deletions = 0, clen = 0, low_cov = 0;
        long sum_cov = 0;
        fprintf(out, ">Consensus%d\n", i);
        for (int j = 0; j < alen; j++) {
            sum_cov += depth[(size_t)j];
            if (depth[(size_t)j] < 3) { low_cov++; fputc(lower[d1.base(i, j)], out); continue; }
            if (ins_score[(size_t)j] > depth[(size_t)j] / 2) {
                int mb = 0;
                for (int b = 1; b < 4; b++) if (ins_base[(size_t)j * 5 + (size_t)b] > ins_base[(size_t)j * 5 + (size_t)mb]) mb = b;
                fputc(upper[mb], out); clen++; insertions++;
            }
            int mb = 0;
            for (int b = 1; b < 5; b++) if (score[(size_t)j * 5 + (size_t)b] > score[(size_t)j * 5 + (size_t)mb]) mb = b;
            if (mb < 4) { fputc(upper[mb], out); good++; clen++; } else deletions++;
        }
        fputc('\n', out);
        o.pf("Average coverage: %f\n", (1.0 * sum_cov) / alen);
        o.pf("Good bases: %d/%d\n", good, alen);
        o.pf("Insertions: %d/%d\n", insertions, alen);
        o.pf("Deletions: %d/%d\n", deletions, alen);
        o.pf("Low coverage bases: %d/%d\n", low_cov, alen);
        o.pf("Consensus length: %d\n", clen);
    }
    fclose(out);
    return 0;
}

}  // extern "C"
