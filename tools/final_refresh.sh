#!/bin/bash
# End-of-round measurement pass on the GPU box: profiles (kernel trace + PMC) of the three BASELINE scenes, every bench
# configuration, the partition tables, the whole GPU test suite.  Usage: tools/final_refresh.sh <tag>
set -e
tag=$1
bash tools/profile_bench.sh ${tag}_headline --steps 4 --warmup 1
bash tools/profile_bench.sh ${tag}_cornell --scene cornell --nx 600 --ny 600 --ns 1000 --steps 4 --warmup 1
bash tools/profile_bench.sh ${tag}_final --scene final --nx 800 --ny 800 --ns 200 --steps 4 --warmup 1
echo "profiles done"
bash tools/bench_all.sh $tag
echo "bench done"
python tools/partition_time.py > gpurun_out/${tag}_partition_random_1200x800_500.log 2>&1
NX=1920 NY=1080 python tools/partition_time.py > gpurun_out/${tag}_partition_random_1920x1080_500.log 2>&1
SCENE=cornell NX=600 NY=600 NS=1000 python tools/partition_time.py > gpurun_out/${tag}_partition_cornell_600x600_1000.log 2>&1
SCENE=final NX=800 NY=800 NS=200 python tools/partition_time.py > gpurun_out/${tag}_partition_final_800x800_200.log 2>&1
echo "partitions done"
python tools/scene_create_time.py > gpurun_out/${tag}_scene_create.log 2>&1
python -m pytest tests -q -m gpu > gpurun_out/${tag}_gpu_tests.log 2>&1; tail -2 gpurun_out/${tag}_gpu_tests.log
