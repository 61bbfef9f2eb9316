D=/tmp/hinge_pipe_data
python tools/stage_times.py --rounds 1 --dir $D > /dev/null 2>&1
cd $D
for k in 1 2 3; do
  t0=$(date +%s.%N)
  HINGE_HOST_TIMING=1 $GRAFT_REPO_ROOT/hinge_amd/bin/hinge pipeline --db G --las G.las -x P --config nominal.ini -o P > /tmp/pipe.$k.log 2>&1
  python3 -c "import time,sys; print('pipeline wall %.3f s' % (time.time() - float(sys.argv[1])))" $t0
done
grep "timing" /tmp/pipe.3.log | sed 's/\[timing\] //' | tr '\n' '|' | cut -c1-3000
