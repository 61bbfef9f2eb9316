cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6t
timeout 1500 python -m pytest tests/test_filter_gpu.py tests/test_fuzz_gpu.py tests/test_full_size_gpu.py tests/test_one_sweep_gpu.py tests/test_cli_gpu.py -x -q -m gpu > gpurun_out/r6t/tests.txt 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r6t/tests.txt
for r in 1 2; do for g in 0 1; do
  HINGE_CALL_LEAN=$g HINGE_BENCH_NO_ASSERT=1 python bench.py --workload cfg3_nctc --parts 2 --no-cpu-baseline --no-e2e --steps 20 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('lean=$g', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['roofline']['kernels_ms_per_step'].items() if 'hinge' in k})"
done; done
python bench.py --workload cfg3_nctc --parts 2 --no-cpu-baseline --no-e2e --steps 5 2>&1 | tail -1 | cut -c1-120
python bench.py --steps 20 --warmup 3 --no-e2e --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cfg2', round(d['ms_per_step'],4), d['checks']['hinges_and_digests_match_cpu_oracle'])"
timeout 1200 python tools/fuzz_pipeline.py --seed 1501 --cases 60 --paths 2>&1 | tail -1
