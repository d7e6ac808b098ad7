#!/bin/bash
# bench.py on the headline (with the CPU baseline) and on the other BASELINE configurations; one JSON line each.
# Usage: tools/bench_all.sh <tag>   -> gpurun_out/<tag>_bench_n1.json, gpurun_out/<tag>_bench_configs.jsonl
set -e
tag=$1
python bench.py > gpurun_out/${tag}_bench_n1.json 2> gpurun_out/${tag}_bench_n1.err
: > gpurun_out/${tag}_bench_configs.jsonl
python bench.py --scene cornell --nx 600 --ny 600 --ns 1000 --no-cpu-baseline >> gpurun_out/${tag}_bench_configs.jsonl 2>> gpurun_out/${tag}_bench_n1.err
python bench.py --scene cornell_smoke --nx 600 --ny 600 --ns 1000 --no-cpu-baseline >> gpurun_out/${tag}_bench_configs.jsonl 2>> gpurun_out/${tag}_bench_n1.err
python bench.py --scene final --nx 800 --ny 800 --ns 200 --no-cpu-baseline >> gpurun_out/${tag}_bench_configs.jsonl 2>> gpurun_out/${tag}_bench_n1.err
python bench.py --scene book1 --nx 1200 --ny 800 --ns 100 --no-cpu-baseline >> gpurun_out/${tag}_bench_configs.jsonl 2>> gpurun_out/${tag}_bench_n1.err
python bench.py --nx 1920 --ny 1080 --ns 500 --no-cpu-baseline >> gpurun_out/${tag}_bench_configs.jsonl 2>> gpurun_out/${tag}_bench_n1.err
