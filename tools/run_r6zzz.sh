# The round's LAST tree: default profile set, config 3, draft counters, the GPU suite (through gpurun)
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r6zzz > gpurun_out/r6zzz_round.log 2>&1; tail -1 gpurun_out/r6zzz/r6zzz_bench.json | cut -c1-160
bash tools/profile_cfg3.sh r6zzz > gpurun_out/r6zzz_cfg3.log 2>&1; tail -1 gpurun_out/r6zzz_cfg3/r6zzz_bench_cfg3.json | cut -c1-160
bash tools/profile_draft.sh r6zzz > gpurun_out/r6zzz_draft.log 2>&1; tail -1 gpurun_out/r6zzz_draft/r6zzz_draft_bench.json | cut -c1-300
timeout 2000 python -m pytest tests -x -q -m gpu > gpurun_out/r6zzz_gpu_tests.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed" gpurun_out/r6zzz_gpu_tests.log | tail -1
