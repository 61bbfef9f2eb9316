set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5j; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
K4="python $R/tools/k4_bench.py --reps 3"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/k4_sq -o r5j --output-format csv -- $K4 > $OUT/k4_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA -d $OUT/k4_sq2 -o r5j --output-format csv -- $K4 > $OUT/k4_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/k4_fetch -o r5j --output-format csv -- $K4 > $OUT/k4_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/k4_write -o r5j --output-format csv -- $K4 > $OUT/k4_write.log 2>&1
cd $R
python tools/pmc_summary.py $(find $OUT/k4_sq $OUT/k4_sq2 -name "*counter_collection.csv") > $OUT/r5j_k4_sq_summary.csv
python tools/pmc_summary.py $(find $OUT/k4_fetch $OUT/k4_write -name "*counter_collection.csv") > $OUT/r5j_k4_pmc_summary.csv
rm -rf $OUT/k4_sq $OUT/k4_sq2 $OUT/k4_fetch $OUT/k4_write
grep -i "image\|stream" $OUT/r5j_k4_sq_summary.csv $OUT/r5j_k4_pmc_summary.csv | cut -c1-260
