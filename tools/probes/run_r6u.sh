cd $GRAFT_REPO_ROOT
bash tools/profile_cfg3.sh r6zzzz > gpurun_out/r6zzzz_cfg3.log 2>&1; tail -1 gpurun_out/r6zzzz_cfg3/r6zzzz_bench_cfg3.json | cut -c1-160; grep hinge gpurun_out/r6zzzz_cfg3/r6zzzz_cfg3_rocprofv3_kernel_stats.csv | cut -c1-60,170-230
python bench.py > gpurun_out/r6zzzz_bench.json 2> gpurun_out/r6zzzz_bench.err; tail -1 gpurun_out/r6zzzz_bench.json | cut -c1-160
timeout 2000 python -m pytest tests -x -q -m gpu > gpurun_out/r6zzzz_gpu_tests.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed" gpurun_out/r6zzzz_gpu_tests.log | tail -1
