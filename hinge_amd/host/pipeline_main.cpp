// hinge_pipeline  ==  `hinge pipeline --db DB --las LAS[.las] [--mlas] -x PREFIX --config nominal.ini [-o OUT]`
// = `hinge filter`, `hinge maximal`, `hinge layout` with these arguments, one after the other IN ONE PROCESS
// (demo/ecoli_demo/run.sh:21-25 runs them as three).  The stages are the very programs of filter_main.cpp, maximal_main.cpp
// and layout_main.cpp, compiled in as functions: same files, byte for byte (tests/test_cli_gpu.py); what one process saves is
// two of the three HIP start-ups and process teardowns (0.3-0.45 s each) and two of the three ingests of
// a single .las (the part is loaded once, with what every stage needs, and handed on).  Measured and not kept: the .las' bytes
// uploaded by a helper thread while `hinge filter` works, so that `hinge maximal` finds its trace points resident - maximal
// 247 -> 173 ms, filter 374 -> 423 ms (the 3.5 GB copy from pageable memory competes with filter's text output), same total.  The three separate executables stay as
// they are.  Exit code: the first stage's that is not 0.
#include <fstream>
#include <functional>
#include <set>
#include <sstream>
#include <unordered_set>
#include "host_common.h"
#include "pairs.h"

#define HINGE_STAGE_MAIN filter_stage
namespace stage_filter {
#include "filter_main.cpp"
}
#undef HINGE_STAGE_MAIN
#undef PART_FAIL
#undef PART_CHECK
#define HINGE_STAGE_MAIN maximal_stage
namespace stage_maximal {
#include "maximal_main.cpp"
}
#undef HINGE_STAGE_MAIN
#undef PART_FAIL
#undef PART_CHECK
#define HINGE_STAGE_MAIN layout_stage
namespace stage_layout {
#include "layout_main.cpp"
}
#undef HINGE_STAGE_MAIN

int main(int argc, char* argv[]) {
    // the stages' own parsers take the flags; `-o` / `--out` (layout's output name) is layout's alone
    std::vector<char*> common{argv[0]}, layout_only;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if ((a == "-o" || a == "--out") && i + 1 < argc) { layout_only.push_back(argv[i]); layout_only.push_back(argv[++i]); }
        else common.push_back(argv[i]);
    }
    hh::pipeline().on = true;
    auto run = [&](int (*stage)(int, char**), bool with_layout_args) {
        std::vector<char*> av(common);
        if (with_layout_args) av.insert(av.end(), layout_only.begin(), layout_only.end());
        av.push_back(nullptr);
        return stage((int)av.size() - 1, av.data());
    };
    int rc = run(stage_filter::filter_stage, false);
    if (rc == 0) rc = run(stage_maximal::maximal_stage, false);
    if (rc == 0) rc = run(stage_layout::layout_stage, true);
    fflush(nullptr);
    if (!getenv("HINGE_SLOW_EXIT")) _exit(rc);   // (as the stages do on their own: no unmapping of the .las, no runtime teardown)
    return rc;
}
