# bench.py's N-rank path with all ranks on ONE device (gloo transport): every part of every rank must reproduce the CPU oracle's
# expectations for that N (tests/golden/bench_expect.json); times mean nothing.   bash tools/bench_ranks_one_device.sh "2 4 8" outdir
O=${2:-gpurun_out/ranks}; mkdir -p $O
for n in ${1:-2 4}; do
  HINGE_BENCH_BACKEND=gloo HINGE_BENCH_ONE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29540 + n)) bench.py --gpus $n --steps 2 --warmup 1 > $O/n${n}_gloo.json 2> $O/n${n}_gloo.err || tail -5 $O/n${n}_gloo.err
  python -c "
import json; d=json.loads(open('$O/n${n}_gloo.json').read().strip().splitlines()[-1]); c=d['checks']; print('N=$n', c['parts_checked'], c['hinges_and_digests_match_cpu_oracle'], c['mask_tables_identical_on_all_ranks'], c['hinges_per_part_and_rank'])"
done
