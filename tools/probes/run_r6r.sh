cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for v in cnt_base cnt_packed cnt_w5; do
  HINGE_LIB=$GRAFT_REPO_ROOT/build_ab/$v.so python bench.py --steps 20 --warmup 3 --no-e2e --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['roofline']['kernels_ms_per_step'].items() if 'hinge' in k}, d['checks']['hinges_and_digests_match_cpu_oracle'])"
done; done
