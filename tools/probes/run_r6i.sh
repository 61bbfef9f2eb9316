cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6i
run() { # name env...
  name=$1; shift
  line=$(env "$@" python bench.py --steps 20 --warmup 3 --no-e2e --no-cpu-baseline 2>>gpurun_out/r6i/err.txt | tail -1)
  echo "$line" > gpurun_out/r6i/$name.json
  echo "$name: $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_per_step", round(d["ms_per_step"],4), "k2_ms", round(d["roofline"].get("avg_launch_ms"),4), "ppl", d["roofline"]["parts_per_launch"], "checks", d["checks"]["hinges_and_digests_match_cpu_oracle"])' 2>&1 | tail -1)"
}
for r in 1 2; do
run base HINGE_STEP_HALVES=0
run halves HINGE_STEP_HALVES=1
run halves_w7 HINGE_STEP_HALVES=1 HINGE_K2_WGS=1792
run halves_w7_prio HINGE_STEP_HALVES=1 HINGE_K2_WGS=1792 HINGE_STEP_SIDE_PRIORITY=-1
run halves_w6 HINGE_STEP_HALVES=1 HINGE_K2_WGS=1536

done
python bench.py --steps 20 --warmup 3 --no-e2e --no-cpu-baseline --parts 8 2>>gpurun_out/r6i/err.txt | tail -1 > gpurun_out/r6i/parts8.json; python -c "
import json; d=json.loads(open('gpurun_out/r6i/parts8.json').read()); print('parts8', d['value'], d['ms_per_step'])"
HINGE_STEP_HALVES=1 HINGE_K2_WGS=1792 python bench.py --steps 20 --warmup 3 --no-e2e --no-cpu-baseline --parts 8 2>>gpurun_out/r6i/err.txt | tail -1 > gpurun_out/r6i/parts8_halves.json; python -c "
import json; d=json.loads(open('gpurun_out/r6i/parts8_halves.json').read()); print('parts8 halves', d['value'], d['ms_per_step'])"
timeout 900 python -m pytest tests/test_consensus_gpu.py tests/test_draft_gpu.py tests/test_capi_library.py -x -q -m gpu > gpurun_out/r6i/tests.txt 2>&1; echo "cns tests rc=$?"; tail -3 gpurun_out/r6i/tests.txt
timeout 900 python tools/fuzz_consensus.py --cases 30 2>&1 | tail -2
python tools/cns_bench.py --no-cpu 2>gpurun_out/r6i/cns.err | tail -1 > gpurun_out/r6i/cns_bench.json; python -c "
import json; d=json.loads(open('gpurun_out/r6i/cns_bench.json').read()); print('cns', d['run_call_ms'], d['kernels_ms'])"
