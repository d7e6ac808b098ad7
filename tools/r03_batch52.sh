#!/bin/bash
# round-3 batch 52: tier-1 size of the middle row (0.69 .. 1.375 pixels per lane) with the hand-off: 1/4 of 1200x800, 1/8 and 1/4 of 1920x1080
set -e
out=gpurun_out/r03_batch52
mkdir -p $out
mk() { echo "tier_auto=0,tier1_pixels=$1,tier1_factor_x10=$2,tier1_depth=4,heavy_factor_x10=15,sparse_factor_x10=$2,sparse_wg_percent=80,sparse_work_percent=40"; }
C=("" "$(mk 8192 20)" "$(mk 6144 20)" "$(mk 4096 20)" "$(mk 6144 25)" "$(mk 4096 25)" "$(mk 3072 30)")
STRIDE=4 ROUNDS=3 python tools/share_sweep.py "${C[@]}" > $out/q_1200.log 2>&1; grep -v amdgpu $out/q_1200.log
STRIDE=4 FIRST=2 ROUNDS=3 python tools/share_sweep.py "${C[@]}" > $out/q_1200_rank2.log 2>&1; grep -v amdgpu $out/q_1200_rank2.log
NX=1920 NY=1080 STRIDE=8 ROUNDS=3 python tools/share_sweep.py "${C[@]}" > $out/e_1920.log 2>&1; grep -v amdgpu $out/e_1920.log
NX=1920 NY=1080 STRIDE=8 FIRST=5 ROUNDS=3 python tools/share_sweep.py "${C[@]}" > $out/e_1920_rank5.log 2>&1; grep -v amdgpu $out/e_1920_rank5.log
NX=1920 NY=1080 STRIDE=4 ROUNDS=3 python tools/share_sweep.py "${C[@]}" > $out/q_1920.log 2>&1; grep -v amdgpu $out/q_1920.log
