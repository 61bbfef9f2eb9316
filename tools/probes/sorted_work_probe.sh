# Does the ORDER in which k_hinge_count visits the pile-ups matter?  A probe build (-DHINGE_PROBE_SORTED_WORK, tools/k2_ab_build.sh)
# rewrites every part's work list on the host before the kernel: 0 dense in the order K2 filed it, 1 sorted by row (storage
# order), 2 dearest first.  Only the kernel's own time in the breakdown means anything (the step contains the rewrite).
for m in none 0 1 2; do if [ $m = none ]; then unset HINGE_PROBE_SORTED_WORK; else export HINGE_PROBE_SORTED_WORK=$m; fi
HINGE_LIB=$PWD/build_ab/probe_sorted.so python bench.py --no-cpu-baseline --no-e2e --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('work list: $m', d['checks']['hinges_and_digests_match_cpu_oracle'], {k: round(v*1e3,1) for k,v in r['kernels_ms_per_step'].items() if 'hinge' in k})"; done
