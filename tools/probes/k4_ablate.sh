# timing-only ablation builds of the trim / classify walks (results are WRONG by construction): 1 no forward walk, 2 no backward walk, 4 no advance sum
for a in 1 2 3 4 7; do echo "== ablate=$a"; K4_NOASSERT=1 HINGE_LIB=$PWD/build_ab/k4a$a.so timeout 600 python tools/k4_bench.py --reps 4 2>&1 | grep "^stream   " ; done
