#!/bin/bash
# end-of-round pass E: bench.py's N > 1 code path rehearsed on one GPU (2 and 4 ranks sharing the device, gloo) with the final build
set -e
for n in 2 4; do
  RT_BENCH_DEVICE=0 RT_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2950$n bench.py --gpus $n --steps 2 --warmup 1 > gpurun_out/r03z_rehearsal_gloo_n${n}_one_gpu.json 2> gpurun_out/r03z_rehearsal_n$n.err || { tail -20 gpurun_out/r03z_rehearsal_n$n.err; exit 1; }
  tail -1 gpurun_out/r03z_rehearsal_gloo_n${n}_one_gpu.json | cut -c1-600
done
