#!/bin/bash
# round-3 batch 32: tail hand-off as the default: GPU suite, shares of the headline and of 1920x1080 with and without, other frames
set -e
out=gpurun_out/r03_batch32
mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1 || { tail -40 $out/gpu_tests.log; exit 1; }
tail -2 $out/gpu_tests.log
python tools/partition_time.py 1 2 4 8 > $out/partition_on.log 2>&1; grep "==" $out/partition_on.log
RT_OPTS=handoff=0 python tools/partition_time.py 1 2 4 8 > $out/partition_off.log 2>&1; grep "==" $out/partition_off.log
NX=1920 NY=1080 python tools/partition_time.py 1 8 > $out/partition_hd_on.log 2>&1; grep "==" $out/partition_hd_on.log
NX=1920 NY=1080 RT_OPTS=handoff=0 python tools/partition_time.py 1 8 > $out/partition_hd_off.log 2>&1; grep "==" $out/partition_hd_off.log
for sc in "final 800 800 200" "book1 1200 800 100" "bouncing 1200 800 100" "cornell_smoke 600 600 200" "earth 1200 800 500" "checkered 1200 800 500"; do
  set -- $sc
  python tools/sweep.py --scene $1 --nx $2 --ny $3 --ns $4 --rounds 3 "handoff=0" "" > $out/scene_$1.log 2>&1 || true
  echo "$sc"; grep -v amdgpu $out/scene_$1.log | cut -c1-200
done
