# Where k_mask_annotate_q20's FETCH_SIZE beyond its span copy comes from: with / without the coverage-bin output.
# (one counter per pass: FETCH_SIZE and WRITE_SIZE together exceed what the hardware collects at once, and rocprofv3 then hangs)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3l; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in nocov cov; do
  flag=""; [ $v = cov ] && flag="--cov-out"
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/k2_$v -o x --output-format csv -- python $R/tools/k2_bench.py --only default --reps 5 $flag > $O/k2_$v.log 2>&1
  echo "== $v"; python $R/tools/pmc_summary.py $(find $O/k2_$v -name "*counter_collection.csv") | grep -i "mask_annotate\|cov_stats"
done
