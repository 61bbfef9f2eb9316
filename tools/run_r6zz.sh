# after the last kernel change of the round (k_cns_vote_tiles): the consensus counters + the default bench line + the GPU suite again
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6zz; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 420 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace_cns -o r6zz_cns --output-format csv -- python $GRAFT_REPO_ROOT/tools/cns_bench.py --no-cpu --steps 5 > $GRAFT_REPO_ROOT/$O/trace_cns.log 2>&1
cp $(find $GRAFT_REPO_ROOT/$O/trace_cns -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/r6zz_cns_rocprofv3_kernel_stats.csv
timeout 420 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $GRAFT_REPO_ROOT/$O/pmc_cns -o r6zz_cns --output-format csv -- python $GRAFT_REPO_ROOT/tools/cns_bench.py --no-cpu --steps 2 > $GRAFT_REPO_ROOT/$O/pmc_cns.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $(find $O/pmc_cns -name "*counter_collection.csv") > $O/r6zz_cns_rocprofv3_sq_summary.csv
rm -rf $O/trace_cns $O/pmc_cns
python bench.py > $O/r6zz_bench.json 2> $O/bench.err; tail -1 $O/r6zz_bench.json | cut -c1-200
timeout 2000 python -m pytest tests -x -q -m gpu > $O/r6zz_gpu_tests.log 2>&1; echo "suite rc=$?"; tail -3 $O/r6zz_gpu_tests.log
