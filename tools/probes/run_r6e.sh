cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6e
timeout 900 python -m pytest tests/test_capi_library.py tests/test_zz_comm_multi_gpu.py "tests/test_cli_gpu.py::test_cli_rows_through_a_one_rank_communicator" tests/test_cli_gpu.py -x -q -m gpu > gpurun_out/r6e/tests.txt 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r6e/tests.txt
timeout 600 python tools/mlas_rccl_check.py --one-rank > gpurun_out/r6e/mlas.txt 2>&1; echo "mlas rc=$?"; tail -8 gpurun_out/r6e/mlas.txt
bash tools/profile_draft.sh r6e 2>&1 | tail -30
timeout 1200 python bench.py > gpurun_out/r6e/bench.json 2> gpurun_out/r6e/bench.err; echo "bench rc=$?"; tail -1 gpurun_out/r6e/bench.json | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
print(json.dumps(d['cpu_baseline'])[:3000])
print({k:v for k,v in d['e2e'].items() if 'speedup' in k and not k.endswith('note')})
"
