# The round's final evidence set (through gpurun): tools/run_r6z.sh
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r6z > gpurun_out/r6z_round.log 2>&1; tail -3 gpurun_out/r6z_round.log | cut -c1-300
bash tools/profile_cfg3.sh r6z > gpurun_out/r6z_cfg3.log 2>&1; tail -4 gpurun_out/r6z_cfg3.log | cut -c1-200
bash tools/profile_draft.sh r6z > gpurun_out/r6z_draft.log 2>&1; tail -3 gpurun_out/r6z_draft.log | cut -c1-200
bash tools/profile_stages.sh r6z > gpurun_out/r6z_stages.log 2>&1; tail -3 gpurun_out/r6z_stages.log | cut -c1-200
SCALE_SMOKE_DRY=1 OUT=$GRAFT_REPO_ROOT/gpurun_out/r6z_scale_smoke bash tools/scale_smoke.sh 1 > gpurun_out/r6z_scale_smoke.log 2>&1; cat gpurun_out/r6z_scale_smoke/summary.txt | cut -c1-250
python bench.py --steps 20 --warmup 3 --no-e2e --no-cpu-baseline --parts 8 2>/dev/null | tail -1 > gpurun_out/r6z_bench_parts8.json
