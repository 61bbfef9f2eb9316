# What a stage spends between the last line of main() and the parent's wait() returning (HINGE_HOST_TIMING's TOTAL vs wall clock),
# on the bench data set:  tools/exit_probe.sh   (through gpurun)
R=${GRAFT_REPO_ROOT:-/root/repo}
D=/tmp/hinge_stage_data
cd $R
python tools/e2e_bench.py --genome 4600000 --exact-config --dir $D > /tmp/e2e.json 2> /tmp/e2e.err
tail -1 /tmp/e2e.json | cut -c1-560
cd $D/hip
for exe in hinging Reads_filter get_maximal_reads; do
  extra=""; [ $exe = hinging ] && extra="-o G"
  for k in 1 2 3; do
    t0=$(date +%s.%N)
    HINGE_HOST_TIMING=1 $R/hinge_amd/bin/$exe --db G --las G.las -x G --config nominal.ini $extra > /tmp/p.log 2>&1
    python3 -c "import time,sys; print(\"$exe wall %.3f s\" % (time.time() - float(sys.argv[1])))" $t0
    grep " TOTAL\|HIP init" /tmp/p.log | grep -v "las.load"
  done
done
