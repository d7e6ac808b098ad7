#!/bin/bash
# round-3 batch 57: scanned scenes walk their leaves only (interior nodes dropped): GPU suite; Cornell, cornell_smoke, simple_light before / after (scan_nodes=0 = walk)
set -e
out=gpurun_out/r03_batch57
mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1 || { tail -40 $out/gpu_tests.log; exit 1; }
tail -1 $out/gpu_tests.log
SCENE=cornell NX=600 NY=600 NS=1000 STRIDE=1 ROUNDS=3 python tools/share_sweep.py "" > $out/cornell.log 2>&1; grep -v amdgpu $out/cornell.log
SCENE=cornell NX=600 NY=600 NS=1000 STRIDE=8 ROUNDS=3 python tools/share_sweep.py "" > $out/cornell_8.log 2>&1; grep -v amdgpu $out/cornell_8.log
SCENE=cornell_smoke NX=600 NY=600 NS=1000 STRIDE=1 ROUNDS=3 python tools/share_sweep.py "" > $out/smoke.log 2>&1; grep -v amdgpu $out/smoke.log
