#!/bin/bash
# rocprofv3 summaries of the kernels bench.py does not run (K4 = k_trim_classify_stream / _rows / list form, k_matching_position,
# k_coverage_bins, k_select_edges):  tools/profile_stages.sh <tag>     (through gpurun; copy gpurun_out/<tag>/<tag>_stages_* to profiles/)
#  * tools/k4_bench.py (every overlap of the bench part through K4, both kernels): kernel stats + FETCH_SIZE / WRITE_SIZE + SQ counters
#  * the three executables on the full bench data set: kernel stats (what each stage launches, and for how long)
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
K4="python $R/tools/k4_bench.py --reps 3"
rocprofv3 --kernel-trace --stats -d $OUT/k4_trace -o $TAG --output-format csv -- $K4 > $OUT/k4_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/k4_fetch -o $TAG --output-format csv -- $K4 > $OUT/k4_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/k4_write -o $TAG --output-format csv -- $K4 > $OUT/k4_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/k4_sq -o $TAG --output-format csv -- $K4 > $OUT/k4_sq.log 2>&1
cd $R
python tools/pmc_summary.py $(find $OUT/k4_fetch $OUT/k4_write -name "*counter_collection.csv") | sed '1s/mean_value/mean_value_KB/' > $OUT/${TAG}_stages_k4_pmc_summary.csv
python tools/pmc_summary.py $(find $OUT/k4_sq -name "*counter_collection.csv") > $OUT/${TAG}_stages_k4_sq_summary.csv
cp $(find $OUT/k4_trace -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_stages_k4_kernel_stats.csv
grep -E "image|stream|rows|identical" $OUT/k4_trace.log > $OUT/${TAG}_stages_k4_bench.txt
# the executables on the bench data set
D=/tmp/hinge_stage_data
python tools/e2e_bench.py --genome 4600000 --exact-config --dir $D > $OUT/${TAG}_stages_e2e.json 2> $OUT/e2e.err
cd /tmp
for st in filter maximal layout; do
  extra=""; [ $st = layout ] && extra="-o G"
  case $st in filter) exe=Reads_filter;; maximal) exe=get_maximal_reads;; layout) exe=hinging;; esac
  # (HINGE_SLOW_EXIT=1: the executables normally leave through _exit(), which would skip the profiler's output)
  (cd $D/hip && HINGE_SLOW_EXIT=1 rocprofv3 --kernel-trace --stats -d $OUT/cli_$st -o $TAG --output-format csv -- $R/hinge_amd/bin/$exe --db G --las G.las -x G --config nominal.ini $extra > $OUT/cli_$st.log 2>&1)
  cp $(find $OUT/cli_$st -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_stages_cli_${st}_kernel_stats.csv 2>/dev/null
done
rm -rf $D
cat $OUT/${TAG}_stages_k4_bench.txt; head -8 $OUT/${TAG}_stages_k4_kernel_stats.csv; cat $OUT/${TAG}_stages_k4_pmc_summary.csv; tail -1 $OUT/${TAG}_stages_e2e.json | cut -c1-600
for st in filter maximal layout; do echo "== $st"; head -12 $OUT/${TAG}_stages_cli_${st}_kernel_stats.csv; done
