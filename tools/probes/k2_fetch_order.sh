R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3l; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in default storage; do
  if [ $v = storage ]; then export HINGE_K2_ORDER_BP=100000000; else unset HINGE_K2_ORDER_BP; fi
  rocprofv3 --pmc FETCH_SIZE -d $O/fetch_$v -o x --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $O/fetch_$v.log 2>&1
  python $R/tools/pmc_summary.py $(find $O/fetch_$v -name "*counter_collection.csv") | grep -i "mask_annotate\|cov_stats"
done
