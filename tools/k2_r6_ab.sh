#!/bin/bash
# Round 6: correctness of the working tree's K2 (and of build_ab variants named in $4), then an A/B of every build_ab/*.so on the
# bench's one-sweep step (through gpurun).     tools/k2_r6_ab.sh <tag> [rounds] [tests: 0/1] ["v_c v_d": variants to test too]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r6x}
ROUNDS=${2:-2}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
if [ "${3:-1}" = 1 ]; then
  timeout 900 python -m pytest tests/test_filter_gpu.py tests/test_one_sweep_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu > $OUT/tests.txt 2>&1
  echo "tests rc=$?"; tail -3 $OUT/tests.txt
  for v in ${4:-}; do
    HINGE_LIB=$R/build_ab/$v.so timeout 900 python -m pytest tests/test_filter_gpu.py tests/test_one_sweep_gpu.py -x -q -m gpu > $OUT/tests_$v.txt 2>&1
    echo "tests $v rc=$?"; tail -3 $OUT/tests_$v.txt
  done
fi
for r in $(seq $ROUNDS); do
  for lib in $R/build_ab/*.so; do
    line=$(HINGE_LIB=$lib timeout 300 python bench.py --steps 20 --warmup 3 --no-e2e --no-cpu-baseline 2>>$OUT/bench_err.txt | tail -1)
    echo "$line" >> $OUT/bench_$(basename $lib .so).json
    echo "$(basename $lib .so): $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_per_step", round(d["ms_per_step"],4), "k2_ms", round(d["roofline"].get("avg_launch_ms"),4), "frac", round(d["roofline"]["frac"],4))' 2>&1 | tail -1)"
  done
done
