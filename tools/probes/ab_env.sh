# A/B of one environment knob through the default bench:  KNOB=HINGE_CALL_LIGHT VALUES="1 0" bash tools/probes/ab_env.sh
for r in 1 2; do for v in ${VALUES:-1 0}; do env ${KNOB:-HINGE_CALL_LIGHT}=$v HINGE_DEBUG_PATHS=${PATHS:-} python bench.py --no-cpu-baseline --no-e2e --steps 10 2>/tmp/ab_env.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('${KNOB:-HINGE_CALL_LIGHT}=$v', round(d['ms_per_step'],4), d['checks']['hinges_and_digests_match_cpu_oracle'], {k: round(v*1e3,1) for k,v in r['kernels_ms_per_step'].items()})"; grep "hinge-call paths" /tmp/ab_env.err | tail -1; done; done
