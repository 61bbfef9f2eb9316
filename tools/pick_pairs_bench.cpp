#include "../hinge_amd/host/pairs.h"
using namespace hh;
int main(int argc, char** argv) {
    ReadDB db; if (db.open(argv[1]) != 0) return 3;
    LasPart las; if (las.load(argv[2], db.rlen) != 0) return 4;
    const int nr = las.r_end - las.r_begin + 1;
    std::vector<std::vector<PairPick>> picks((size_t)nr);
    auto t0 = std::chrono::steady_clock::now();
    parallel_dynamic((int64_t)nr, 64, [&](int64_t k0, int64_t k1) {
        for (int64_t k = k0; k < k1; k++) pick_pairs(las, las.r_begin + (int)k, true, 2, [](int) { return true; }, picks[(size_t)k]);
    });
    auto t1 = std::chrono::steady_clock::now();
    size_t tot = 0; for (auto& p : picks) tot += p.size();
    printf("threads %d: pick_pairs %.1f ms, %zu pairs\n", host_threads(), std::chrono::duration<double, std::milli>(t1 - t0).count(), tot);
}
