#!/bin/bash
# round-3 batch 20: bench.py twice (CPU baseline stability), and the N > 1 code path of bench.py rehearsed on one GPU (2 and 4 ranks
# sharing device 0, gloo instead of RCCL)
set -e
out=gpurun_out/r03_batch20
mkdir -p $out
python bench.py --steps 3 --warmup 1 > $out/bench_a.json 2> $out/bench_a.err
python bench.py --steps 3 --warmup 1 > $out/bench_b.json 2> $out/bench_b.err
python - <<'PY'
import json
for f in ("gpurun_out/r03_batch20/bench_a.json", "gpurun_out/r03_batch20/bench_b.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["cpu_baseline"]["value"], d["cpu_baseline"]["sample"][:230])
PY
for n in 2 4; do
  RT_BENCH_DEVICE=0 RT_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2950$n bench.py --gpus $n --steps 2 --warmup 1 > $out/rehearsal_n$n.json 2> $out/rehearsal_n$n.err || { tail -20 $out/rehearsal_n$n.err; exit 1; }
  tail -1 $out/rehearsal_n$n.json | cut -c1-1500
done
