# What k_hinge_count waits for: L2 hit rate, texture-addresser busy, memory-wait share (separate --pmc passes, kernel stats only)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for set in "TCC_HIT_sum TCC_MISS_sum" "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_BUSY_CYCLES" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum"; do
  rm -rf /tmp/hcp; timeout 300 rocprofv3 --pmc $set -d /tmp/hcp -o x --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > /tmp/hcp.log 2>&1 || tail -2 /tmp/hcp.log
  python $R/tools/pmc_summary.py $(find /tmp/hcp -name "*counter_collection.csv") 2>/dev/null | grep "hinge_count\|q20_batch"
done
