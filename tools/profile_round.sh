#!/bin/bash
# One profile set of `python bench.py` on the GPU box:  tools/profile_round.sh <tag>
# (run through gpurun; results land in gpurun_out/<tag>/, copy the four summaries into profiles/<tag>_*).
# Separate passes: kernel trace + stats, FETCH_SIZE, WRITE_SIZE, SQ counters (PMC never shares a pass with tracing).
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e"
PMCB="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e"      # counter passes serialise the kernels: few steps
(cd $R && python bench.py > $OUT/bench.json 2> $OUT/bench.err)
timeout 420 rocprofv3 --kernel-trace --stats -d $OUT/trace -o $TAG --output-format csv -- $BENCH > $OUT/trace.log 2>&1
timeout 420 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o $TAG --output-format csv -- $PMCB > $OUT/pmc_fetch.log 2>&1
timeout 420 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o $TAG --output-format csv -- $PMCB > $OUT/pmc_write.log 2>&1
timeout 420 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/pmc_sq -o $TAG --output-format csv -- $PMCB > $OUT/pmc_sq.log 2>&1
cd $R
python tools/pmc_summary.py $(find $OUT/pmc_fetch $OUT/pmc_write -name "*counter_collection.csv") | sed '1s/mean_value/mean_value_KB/' > $OUT/${TAG}_rocprofv3_pmc_summary.csv
python tools/pmc_summary.py $(find $OUT/pmc_sq -name "*counter_collection.csv") > $OUT/${TAG}_rocprofv3_sq_summary.csv
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_rocprofv3_kernel_stats.csv
cp $OUT/bench.json $OUT/${TAG}_bench.json
# `hinge consensus`: its kernels under the tracer, E. coli-sized draft
(cd /tmp && timeout 420 rocprofv3 --kernel-trace --stats -d $OUT/trace_cns -o ${TAG}_cns --output-format csv -- python $R/tools/cns_bench.py --no-cpu --steps 5 > $OUT/trace_cns.log 2>&1)
cp $(find $OUT/trace_cns -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_cns_rocprofv3_kernel_stats.csv 2>/dev/null
# ... and their instruction counters (the roofline tools/cns_bench.py states for k_cns_realign is vector-instruction issue: it reads the newest of these)
(cd /tmp && timeout 420 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/pmc_cns -o ${TAG}_cns --output-format csv -- python $R/tools/cns_bench.py --no-cpu --steps 2 > $OUT/pmc_cns.log 2>&1)
python $R/tools/pmc_summary.py $(find $OUT/pmc_cns -name "*counter_collection.csv") > $OUT/${TAG}_cns_rocprofv3_sq_summary.csv 2>/dev/null
tail -1 $OUT/bench.json | cut -c1-400
head -12 $OUT/${TAG}_rocprofv3_kernel_stats.csv
cat $OUT/${TAG}_rocprofv3_pmc_summary.csv
