cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in w12 w10 w9; do
  HINGE_LIB=$GRAFT_REPO_ROOT/build_ab/$v.so python tools/cns_bench.py --no-cpu 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['run_call_ms'], d['kernels_ms'])"
done; done
HINGE_CNS_WGS_PER_CU=8 HINGE_LIB=$GRAFT_REPO_ROOT/build_ab/w10.so python tools/cns_bench.py --no-cpu 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('w10 grid 8/cu', d['run_call_ms'], d['kernels_ms'])"
timeout 600 python -m pytest tests/test_consensus_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python tools/fuzz_consensus.py --cases 20 2>&1 | tail -1
