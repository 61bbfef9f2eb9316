#!/bin/bash
# A/B of library builds on ONE box (box-to-box differences are ~1 us on k_mask_annotate_q20): every build_ab/*.so runs
# tools/k2_bench.py in turn, ROUNDS times.    tools/k2_ab.sh [rounds]     (through gpurun; build the variants here first:
#   hipcc ... -shared -o build_ab/<name>.so hinge_amd/csrc/hinge_capi.hip)
R=${GRAFT_REPO_ROOT:-/root/repo}
ROUNDS=${1:-2}
for r in $(seq $ROUNDS); do
  for lib in $R/build_ab/*.so; do
    echo "$(basename $lib .so): $(HINGE_LIB=$lib timeout 200 python $R/tools/k2_bench.py --cov-out --reps 20 --only default 2>&1 | tail -1)"
  done
done
