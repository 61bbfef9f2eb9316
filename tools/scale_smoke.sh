#!/bin/bash
# First contact with an N-GPU node: ONE command that yields a scaling curve AND a correctness verdict.
#   tools/scale_smoke.sh [N]        N = GPUs to use (default: all visible)        output: gpurun_out/scale_smoke/ (or $OUT)
# What it runs, in order, and what each step proves:
#   1. pytest -m gpu tests/test_zz_comm_multi_gpu.py     the C++ hosts' RCCL branch (hinge_comm_create / hinge_comm_exchange_mask_rows /
#                                                     hinge_comm_allgather_rows) between DISTINCT devices, every device order, results against the host exchange
#   2. bench.py --gpus n (weak) for n = 1, 2, 4 .. N  one process per GPU over RCCL; every run asserts per part and rank the hinge
#                                                     counts + digests against the CPU oracle's expectations (tests/golden/bench_expect.json)
#   3. bench.py --gpus N --scaling strong             the same total work cut N ways
#   4. tools/mlas_rccl_check.py + the cfg4 digest test `hinge filter | maximal | layout --mlas` with one rank per visible GPU: the logs must say "mask rows /
#                                                     containment candidates / classified matches over RCCL" and the files must equal a
#                                                     HINGE_HOST_EXCHANGE=1 run's; then all three stages, 8 blocks, 20 files against the CPU oracle's committed digests
#   5. tests/test_dist_gpu.py                         hinge_amd/dist.py's sharded filter / maximal / layout over real RCCL
# Every step's exit code goes to summary.txt; the JSON lines of 2-3 to scale.jsonl (efficiency is the reader's to compute).
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
NVIS=$(python - <<'PY'
from hinge_amd import capi
print(capi.load_library().hinge_device_count())
PY
)
N=${1:-$NVIS}
OUT=${OUT:-$R/gpurun_out/scale_smoke}
mkdir -p "$OUT"
: > "$OUT/summary.txt"; : > "$OUT/scale.jsonl"
note() { echo "$1" | tee -a "$OUT/summary.txt"; }
note "scale_smoke: $NVIS visible GPU(s), using $N; tree $(git rev-parse --short HEAD 2>/dev/null || echo '?')"
if [ "$N" -lt 2 ] && [ "${SCALE_SMOKE_DRY:-0}" != "1" ]; then note "fewer than 2 GPUs: the multi-device steps would only repeat the single-GPU suite - nothing to do (SCALE_SMOKE_DRY=1 runs the script's steps on one GPU anyway: a rehearsal of the script, not a scaling result)"; exit 3; fi

python -m pytest tests/test_zz_comm_multi_gpu.py -x -q -m gpu > "$OUT/1_comm.log" 2>&1; note "1 comm between devices: rc=$?"

PORT=29610
n=1
while [ "$n" -le "$N" ]; do
  if [ "$n" -eq 1 ]; then python bench.py --gpus 1 --no-e2e > "$OUT/2_bench_n1.json" 2> "$OUT/2_bench_n1.err"
  else python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus "$n" --no-e2e > "$OUT/2_bench_n$n.json" 2> "$OUT/2_bench_n$n.err"; fi
  rc=$?; tail -1 "$OUT/2_bench_n$n.json" >> "$OUT/scale.jsonl"
  note "2 bench weak n=$n: rc=$rc $(tail -1 "$OUT/2_bench_n$n.json" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("value=%.4g ms_per_step=%.4f checks=%s" % (d["value"], d["ms_per_step"], d["checks"]["hinges_and_digests_match_cpu_oracle"]))' 2>/dev/null)"
  PORT=$((PORT + 1)); n=$((n * 2))
done
if [ "$N" -ge 2 ]; then python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus "$N" --scaling strong --no-e2e > "$OUT/3_bench_strong_n$N.json" 2> "$OUT/3_bench_strong_n$N.err"
else python bench.py --gpus 1 --scaling strong --no-e2e > "$OUT/3_bench_strong_n$N.json" 2> "$OUT/3_bench_strong_n$N.err"; fi
rc=$?; tail -1 "$OUT/3_bench_strong_n$N.json" >> "$OUT/scale.jsonl"; note "3 bench strong n=$N: rc=$rc"

if [ "$N" -ge 2 ]; then python tools/mlas_rccl_check.py > "$OUT/4_mlas_rccl.log" 2>&1; else python tools/mlas_rccl_check.py --one-rank > "$OUT/4_mlas_rccl.log" 2>&1; fi
note "4a filter / maximal / layout --mlas, RCCL == host exchange: rc=$? ($(tail -1 "$OUT/4_mlas_rccl.log"))"
python -m pytest "tests/test_full_size_gpu.py::test_configs_at_their_full_size_against_committed_digests[cfg4_yeast]" -x -q -m gpu > "$OUT/4_cfg4_digests.log" 2>&1
note "4b cfg4 (8 blocks, --mlas, one rank per visible GPU) against the oracle's digests: rc=$?"

python -m pytest tests/test_dist_gpu.py -x -q -m gpu > "$OUT/5_dist.log" 2>&1; note "5 dist.py over RCCL: rc=$?"
cat "$OUT/summary.txt"
