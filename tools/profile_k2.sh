#!/bin/bash
# SQ counters of the two K2 kernels side by side (tools/k2_bench.py --only ...), two --pmc passes:  tools/profile_k2.sh <tag>
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/k2_bench.py --cov-out --reps 3 --only default"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/k2_sq1 -o $TAG --output-format csv -- python $R/tools/k2_bench.py --cov-out --reps 3 --only "default" > $OUT/k2_sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_SCA -d $OUT/k2_sq2 -o $TAG --output-format csv -- python $R/tools/k2_bench.py --cov-out --reps 3 --only "default" > $OUT/k2_sq2.log 2>&1
cd $R
python tools/pmc_summary.py $(find $OUT/k2_sq1 $OUT/k2_sq2 -name "*counter_collection.csv") > $OUT/${TAG}_k2_sq_summary.csv
grep -E "k_mask_annotate|k_cov_stats" $OUT/${TAG}_k2_sq_summary.csv
