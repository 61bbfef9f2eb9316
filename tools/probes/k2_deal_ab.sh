# XCD-contiguous deal of k_mask_annotate_q20's reads (HINGE_K2_DEAL=1) against the longest-first draw: step time, then FETCH_SIZE.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3o; mkdir -p $O
cd $R
HINGE_K2_DEAL=1 timeout 600 python -m pytest tests/test_filter_gpu.py -x -q -k "matches_oracle or packed_route or long_reads" 2>&1 | tail -2
for d in 0 1 0 1; do HINGE_K2_DEAL=$d python bench.py --steps 30 --warmup 5 --no-e2e 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('deal', $d, '%.4f' % d['ms_per_step'], 'K2 %.1f us' % (1e3*d['roofline']['avg_launch_ms']), d['checks']['parts_checked'])"; done
cd /tmp; export TMPDIR=/tmp
for d in 0 1; do
  HINGE_K2_DEAL=$d timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/fetch_deal$d -o x --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $O/fetch_deal$d.log 2>&1
  echo "deal $d"; python $R/tools/pmc_summary.py $(find $O/fetch_deal$d -name "*counter_collection.csv") | grep -i "mask_annotate"
done
