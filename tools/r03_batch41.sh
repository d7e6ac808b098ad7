#!/bin/bash
# round-3 batch 41: tail hand-off in the scenes scanned in lockstep (Cornell box, whole and 1/8; cornell_smoke; simple_light)
set -e
out=gpurun_out/r03_batch41
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $out/gpu_parity.log 2>&1 || { tail -40 $out/gpu_parity.log; exit 1; }
tail -1 $out/gpu_parity.log
export SCENE=cornell NX=600 NY=600 NS=1000
STRIDE=1 ROUNDS=3 python tools/share_sweep.py "handoff_scan=0" "" "handoff_pixels=4096" "handoff_pixels=16384" "handoff_pixels=32768" "handoff_pixels=65536" "handoff_pixels=131072" "handoff_pixels=400000" > $out/cornell.log 2>&1; grep -v amdgpu $out/cornell.log
STRIDE=8 ROUNDS=3 python tools/share_sweep.py "handoff_scan=0" "" "handoff_pixels=4096" "handoff_pixels=8192" "handoff_pixels=16384" "handoff_pixels=32768" "handoff_pixels=65536" > $out/cornell_8.log 2>&1; grep -v amdgpu $out/cornell_8.log
SCENE=cornell_smoke STRIDE=1 ROUNDS=3 python tools/share_sweep.py "handoff_scan=0" "" "handoff_pixels=16384" "handoff_pixels=65536" > $out/smoke.log 2>&1; grep -v amdgpu $out/smoke.log
SCENE=cornell_smoke STRIDE=8 ROUNDS=3 python tools/share_sweep.py "handoff_scan=0" "" "handoff_pixels=8192" "handoff_pixels=32768" > $out/smoke_8.log 2>&1; grep -v amdgpu $out/smoke_8.log
