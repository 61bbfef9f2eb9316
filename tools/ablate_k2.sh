#!/bin/bash
# Where the instructions of k_mask_annotate_q20 go: an ablation build of the library (-DHINGE_ABLATE: the kernel leaves every read
# after phase k) under rocprofv3 counters, one run per phase.  tools/ablate_k2.sh <tag>   (through gpurun; results in gpurun_out/<tag>/)
#    9 = rows + histogram                1 = + hot-word fold     6 = all but the prefix scan
#    7 = all but the mask pass          2 = up to the mask outputs (no gate / candidates)   3 = + gate   4 = + candidates   0 = whole kernel
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
if [ "${2:-}" = trace ]; then   # per-read time stamps (their own build: the stamps change the kernel)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -I$R/include -DHINGE_ABLATE -DHINGE_K2_TRACE -shared -o $OUT/libhinge_hip_trace.so $R/hinge_amd/csrc/hinge_capi.hip || exit 1
  HINGE_LIB=$OUT/libhinge_hip_trace.so python $R/tools/k2_trace.py > $OUT/k2_trace.txt 2>&1; cat $OUT/k2_trace.txt
  exit 0
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -I$R/include -DHINGE_ABLATE -shared -o $OUT/libhinge_hip_ablate.so $R/hinge_amd/csrc/hinge_capi.hip || exit 1
cd /tmp && export TMPDIR=/tmp
for ph in 9 1 6 7 2 3 4 0; do
  HINGE_LIB=$OUT/libhinge_hip_ablate.so HINGE_ABLATE_PHASE=$ph timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVE_CYCLES -d $OUT/abl_$ph -o $TAG --output-format csv -- python $R/tools/k2_bench.py --cov-out --reps 3 --only "default" --no-check > $OUT/abl_$ph.log 2>&1
  echo "phase $ph: $(grep "^default" $OUT/abl_$ph.log | tail -1)"
  python $R/tools/pmc_summary.py $(find $OUT/abl_$ph -name "*counter_collection.csv") | grep q20 | sed "s/^/  /"
done
