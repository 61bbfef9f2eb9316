cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6g
timeout 1200 python -m pytest tests/test_draft_gpu.py tests/test_filter_gpu.py tests/test_fuzz_gpu.py tests/test_one_sweep_gpu.py -x -q -m gpu > gpurun_out/r6g/tests.txt 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r6g/tests.txt
for g in 0 1; do
  HINGE_CALL_GROUP=$g HINGE_BENCH_NO_ASSERT=1 python bench.py --workload cfg3_nctc --parts 2 --no-cpu-baseline --no-e2e --steps 20 2> gpurun_out/r6g/cfg3_g$g.err | tail -1 > gpurun_out/r6g/cfg3_g$g.json
  python -c "
import json,sys
d=json.loads(open('gpurun_out/r6g/cfg3_g$g.json').read())
print('group=$g', 'ms_per_step', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['roofline']['kernels_ms_per_step'].items()}, d.get('checks',{}).get('hinges_and_digests_match_cpu_oracle'))
"
done
HINGE_BENCH_NO_ASSERT=0 python bench.py --workload cfg3_nctc --parts 2 --no-cpu-baseline --no-e2e --steps 5 2>&1 | tail -1 | cut -c1-200
bash tools/profile_draft.sh r6g 2>&1 | tail -12
