cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6h
timeout 1500 python -m pytest tests/test_filter_gpu.py tests/test_fuzz_gpu.py tests/test_ref_direct_gpu.py tests/test_full_size_gpu.py -x -q -m gpu > gpurun_out/r6h/tests.txt 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r6h/tests.txt
for r in 1 2; do for v in nocoop coop; do
  HINGE_LIB=$GRAFT_REPO_ROOT/build_ab/$v.so HINGE_BENCH_NO_ASSERT=1 python bench.py --workload cfg3_nctc --parts 2 --no-cpu-baseline --no-e2e --steps 20 2> gpurun_out/r6h/cfg3_$v.err | tail -1 > gpurun_out/r6h/cfg3_$v.json
  python -c "
import json,sys
d=json.loads(open('gpurun_out/r6h/cfg3_$v.json').read())
print('$v', 'ms_per_step', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['roofline']['kernels_ms_per_step'].items()})
"
done; done
HINGE_DEBUG_FORCE_EXACT=2 timeout 600 python -m pytest tests/test_filter_gpu.py -x -q -m gpu -k "exact or tie or route" 2>&1 | tail -2
timeout 900 python tools/fuzz_pipeline.py --seed 91 --cases 40 --paths 2>&1 | tail -3
