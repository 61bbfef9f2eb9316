// Test helper (CPU only): loads a DAZZ_DB + .las through the executables' ingest code and dumps the SoA columns.
// usage: ingest_dump DB LAS OUT      exit code = LasPart::load() result (0, 255 for -1, 254 for -2)
#include "../../hinge_amd/host/host_common.h"

using namespace hh;

template <typename V> static void put(FILE* f, const V& v) {
    const int64_t n = (int64_t)v.size();
    fwrite(&n, 8, 1, f);
    if (n) fwrite(v.data(), sizeof(*v.data()), (size_t)n, f);
}

int main(int argc, char** argv) {
    if (argc != 4) return 2;
    ReadDB db;
    if (db.open(argv[1]) != 0) return 3;
    LasPart las;
    const int rc = las.load(argv[2], db.rlen);
    if (rc != 0) return rc & 255;
    printf("%s\n", las.indexed_in_pieces ? "pieces" : "sequential");
    FILE* f = fopen(argv[3], "wb");
    if (!f) return 4;
    const int64_t hdr[4] = {las.novl, las.tspace, las.r_begin, las.r_end};
    fwrite(hdr, 8, 4, f);
    put(f, las.row_ptr); put(f, las.a_span); put(f, las.b_span); put(f, las.b_flag); put(f, las.trace_off); put(f, las.tlen);
    put(f, las.rec_row_ptr); put(f, las.rec_b); put(f, las.rec_kept); put(f, las.self_a); put(f, las.self_span);
    fclose(f);
    return 0;
}
