# A/B of the batched k_mask_annotate_q20 launch: BATCH:STEAL pairs (HINGE_K2_BATCH, HINGE_K2_STEAL) through the default bench
for r in 1 2; do for b in ${K2_AB:-1:2 1:1 0:0}; do HINGE_K2_BATCH=${b%%:*} HINGE_K2_STEAL=${b##*:} python bench.py --no-cpu-baseline --no-e2e --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('batch:steal=$b', round(d['ms_per_step'],4), round(r['avg_launch_ms']*1e3,2), r['launches_timed'], round(r['frac'],4), d['checks']['hinges_and_digests_match_cpu_oracle'], {k: round(v*1e3,1) for k,v in r['kernels_ms_per_step'].items()})"; done; done
