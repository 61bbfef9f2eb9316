# Host-phase timing (HINGE_HOST_TIMING=1) of the layout and filter executables on the bench data set:  tools/stage_timing.sh  (through gpurun)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
D=/tmp/hinge_stage_data
cd $R
python tools/e2e_bench.py --genome 4600000 --exact-config --dir $D > /tmp/e2e.json 2> /tmp/e2e.err
tail -1 /tmp/e2e.json | cut -c1-500
cd $D/hip
for k in 1 2 3; do
  t0=$(date +%s.%N)
  HINGE_HOST_TIMING=1 $R/hinge_amd/bin/hinging --db G --las G.las -x G --config nominal.ini -o G > /tmp/lay.$k.log 2>&1
  python3 -c "import time,sys; print(\"layout wall %.3f s\" % (time.time() - float(sys.argv[1])))" $t0
  grep -i "ms\| s$" /tmp/lay.$k.log | tail -30
done
for k in 1 2; do
  t0=$(date +%s.%N)
  HINGE_HOST_TIMING=1 $R/hinge_amd/bin/Reads_filter --db G --las G.las -x G --config nominal.ini > /tmp/fil.$k.log 2>&1
  python3 -c "import time,sys; print(\"filter wall %.3f s\" % (time.time() - float(sys.argv[1])))" $t0
  grep -i "ms\| s$" /tmp/fil.$k.log | tail -30
done
