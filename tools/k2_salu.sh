# Instruction counts of k_mask_annotate_q20 on the bench part (one counter pass):  tools/k2_salu.sh   (through gpurun)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/k2sq
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d /tmp/k2sq -o x --output-format csv -- python $R/tools/k2_bench.py --cov-out --reps 3 --only default > /tmp/k2sq.log 2>&1
echo "rc=$?"
python $R/tools/pmc_summary.py $(find /tmp/k2sq -name "*counter_collection.csv") | grep q20
