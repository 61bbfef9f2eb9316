// Test helper (CPU only): loads a DAZZ_DB + .las through the executables' ingest code and dumps the SoA columns.
// usage: ingest_dump DB LAS OUT      exit code = LasPart::load() result (0, 255 for -1, 254 for -2)
#include "../../hinge_amd/host/host_common.h"

using namespace hh;

template <typename V> static void put(FILE* f, const V& v) {
    const int64_t n = (int64_t)v.size();
    fwrite(&n, 8, 1, f);
    if (n) fwrite(v.data(), sizeof(*v.data()), (size_t)n, f);
}

// usage: ingest_dump DB LAS OUT | ingest_dump --paf FASTA PAF OUT | ingest_dump --fasta FASTA OUT (read lengths as int32)
int main(int argc, char** argv) {
    if (argc == 4 && std::string(argv[1]) == "--fasta") {
        std::vector<int32_t> rlen;
        if (read_fasta_lengths(argv[2], rlen) != 0) return 3;
        FILE* f = fopen(argv[3], "wb");
        if (!f) return 4;
        if (!rlen.empty()) fwrite(rlen.data(), 4, rlen.size(), f);
        fclose(f);
        return 0;
    }
    const bool paf = argc == 5 && std::string(argv[1]) == "--paf";
    if (paf) { argv++; argc--; }
    if (argc != 4) return 2;
    ReadDB db;
    if (paf) { if (read_fasta_lengths(argv[1], db.rlen) != 0) return 3; }
    else if (db.open(argv[1]) != 0) return 3;
    LasPart las;
    las.want_span16 = true;
    const int rc = paf ? las.load_paf(argv[2], db.rlen) : las.load(argv[2], db.rlen);
    if (rc != 0) return rc & 255;
    printf("%s\n", las.indexed_in_pieces ? "pieces" : "sequential");
    FILE* f = fopen(argv[3], "wb");
    if (!f) return 4;
    const int64_t hdr[4] = {las.novl, las.tspace, las.r_begin, las.r_end};
    fwrite(hdr, 8, 4, f);
    put(f, las.row_ptr); put(f, las.a_span); put(f, las.b_span); put(f, las.b_flag); put(f, las.trace_off); put(f, las.tlen);
    put(f, las.rec_row_ptr); put(f, las.rec_b); put(f, las.rec_kept); put(f, las.self_a); put(f, las.self_span);
    put(f, las.span16);
    const std::vector<int64_t> facts = {(int64_t)las.max_pile, las.spans_in_range ? 1 : 0};
    put(f, facts);
    // what hinge_set_las_image gets besides the mapped file (round 5): the 64-overlap windows of the kept overlaps
    const bool img = !paf && las.build_image_table();
    const std::vector<int64_t> img_ok = {img ? 1 : 0};
    put(f, img_ok); put(f, las.img_win_base); put(f, las.img_rec_rel);
    fclose(f);
    return 0;
}
