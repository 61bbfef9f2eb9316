# Cost of the batched exchanges on ONE rank (HINGE_FORCE_COLLECTIVES=1: every collective really goes through RCCL) against the
# plain run, and bench.py's N = 2 path on the full workload with both ranks on this GPU (gloo transport; results asserted
# against the CPU oracle's N = 2 expectations, times meaningless).   bash tools/coll_overhead.sh [outdir]   (through gpurun)
O=${1:-gpurun_out/coll}; mkdir -p $O
python bench.py --steps 40 --warmup 5 --no-e2e > $O/plain.json 2> $O/plain.err || tail -3 $O/plain.err
for g in 1 2; do
  HINGE_FORCE_COLLECTIVES=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 40 --warmup 5 --no-e2e --gather-groups $g > $O/forced_g$g.json 2> $O/forced_g$g.err || tail -3 $O/forced_g$g.err
done
python - <<P
import json
for n in ("plain", "forced_g1", "forced_g2"):
    try:
        d = json.loads(open("$O/%s.json" % n).read().strip().splitlines()[-1])
        print("%-10s ms_per_step %.4f  collectives/step %s  checked %s" % (n, d["ms_per_step"], d["config"]["collectives_per_step"], d["checks"]["parts_checked"]))
    except Exception as e:
        print(n, "failed", e)
P
if [ "${RANKS2:-1}" = 1 ]; then
  HINGE_BENCH_BACKEND=gloo HINGE_BENCH_ONE_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 3 --warmup 1 > $O/n2_gloo.json 2> $O/n2_gloo.err || tail -5 $O/n2_gloo.err
  python -c "
import json; d=json.loads(open('$O/n2_gloo.json').read().strip().splitlines()[-1]); print('N=2 (gloo, one device):', d['checks'])"
fi
