# The config-3 (repeat-rich) kernel-level line:  tools/profile_cfg3.sh <tag>   (through gpurun; copy gpurun_out/<tag>_cfg3/* to profiles/)
TAG=${1:-rXX}
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/${TAG}_cfg3; O=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_cfg3
cd $GRAFT_REPO_ROOT && HINGE_BENCH_NO_ASSERT=1 python bench.py --workload cfg3_nctc --parts 2 --no-cpu-baseline --no-e2e --steps 20 > $O/${TAG}_bench_cfg3.json 2> $O/bench_cfg3.err
cd /tmp && export TMPDIR=/tmp; HINGE_BENCH_NO_ASSERT=1 timeout 420 rocprofv3 --kernel-trace --stats -d $O/trace -o cfg3 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload cfg3_nctc --parts 2 --no-cpu-baseline --no-e2e --steps 20 > $O/trace.log 2>&1
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/${TAG}_cfg3_rocprofv3_kernel_stats.csv; rm -rf $O/trace
tail -1 $O/${TAG}_bench_cfg3.json | cut -c1-300; head -12 $O/${TAG}_cfg3_rocprofv3_kernel_stats.csv | cut -c1-150
