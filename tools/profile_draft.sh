# `hinge draft`'s ladder kernels (tools/draft_bench.py): the bench line, the kernel trace and two counter passes.
#   tools/profile_draft.sh <tag>     (through gpurun; copy gpurun_out/<tag>_draft/<tag>_draft_* to profiles/)
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${TAG}_draft; mkdir -p $O
cd $R && python tools/draft_bench.py ${DRAFT_ARGS:-} > $O/${TAG}_draft_bench.json 2> $O/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/draft_bench.py --no-cpu --steps 2 ${DRAFT_ARGS:-}"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o d --output-format csv -- $CMD > $O/trace.log 2>&1
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/${TAG}_draft_rocprofv3_kernel_stats.csv; rm -rf $O/trace
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $O/sq1 -o d --output-format csv -- $CMD > $O/sq1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS -d $O/sq2 -o d --output-format csv -- $CMD > $O/sq2.log 2>&1
cd $R && python tools/pmc_summary.py $(find $O/sq1 $O/sq2 -name "*counter_collection.csv") | grep -E "kernel,|k_draft" > $O/${TAG}_draft_rocprofv3_sq_summary.csv
rm -rf $O/sq1 $O/sq2
tail -1 $O/${TAG}_draft_bench.json | cut -c1-1200; grep k_draft $O/${TAG}_draft_rocprofv3_kernel_stats.csv | cut -c1-160; cat $O/${TAG}_draft_rocprofv3_sq_summary.csv
