# A/B of library builds under build_ab/ (tools/k2_ab_build.sh) through the default bench:  LIBS="a b c" bash tools/probes/ab_libs.sh
for r in 1 2; do for lib in ${LIBS:-base}; do HINGE_LIB=$PWD/build_ab/$lib.so python bench.py --no-cpu-baseline --no-e2e --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$lib', round(d['ms_per_step'],4), d['checks']['hinges_and_digests_match_cpu_oracle'], {k: round(v*1e3,1) for k,v in r['kernels_ms_per_step'].items()})"; done; done
