#!/bin/bash
# round-3 batch 34: tier sizes of the shares re-fitted with the tail hand-off on (rank 0 of 2 / 4 / 8 of the headline frame)
set -e
out=gpurun_out/r03_batch34
mkdir -p $out
mk() { echo "tier_auto=0,tier1_pixels=$1,tier1_factor_x10=$2,tier1_depth=$3,heavy_factor_x10=$4,sparse_factor_x10=$5,sparse_wg_percent=80,sparse_work_percent=$6"; }
STRIDE=8 ROUNDS=3 python tools/share_sweep.py "" "$(mk 16384 15 4 15 15 60)" "$(mk 8192 15 4 15 15 60)" "$(mk 8192 20 4 15 20 60)" "$(mk 4096 20 4 15 20 60)" "$(mk 4096 30 4 15 30 60)" "$(mk 8192 20 4 15 20 40)" "$(mk 8192 20 2 15 20 40)" "$(mk 16384 15 2 15 15 60)" "$(mk 4096 30 4 20 30 20)" "$(mk 2048 40 4 20 40 20)" > $out/share_8.log 2>&1; grep -v amdgpu $out/share_8.log
STRIDE=4 ROUNDS=3 python tools/share_sweep.py "" "$(mk 8192 20 4 15 20 40)" "$(mk 4096 20 4 15 20 40)" "$(mk 4096 30 4 15 25 40)" "$(mk 4096 30 4 15 25 5)" "$(mk 2048 30 4 15 30 5)" "$(mk 2048 40 3 20 40 5)" "$(mk 16384 15 4 15 15 40)" "$(mk 8192 15 4 15 15 60)" > $out/share_4.log 2>&1; grep -v amdgpu $out/share_4.log
STRIDE=2 ROUNDS=3 python tools/share_sweep.py "" "$(mk 4096 30 4 15 25 5)" "$(mk 2048 30 4 15 25 5)" "$(mk 2048 40 3 20 40 5)" "$(mk 1536 40 3 20 40 5)" "$(mk 8192 20 4 15 20 5)" "$(mk 8192 20 4 15 20 40)" "$(mk 4096 20 4 15 20 20)" > $out/share_2.log 2>&1; grep -v amdgpu $out/share_2.log
STRIDE=1 ROUNDS=3 python tools/share_sweep.py "" "tier1_pixels=1024" "tier1_pixels=2048" "tier1_pixels=3072" "tier1_pixels=2048,tier1_depth=4" "tier1_factor_x10=30" "tier1_factor_x10=50" "sparse_work_percent=10" "sparse_work_percent=3" > $out/share_1.log 2>&1; grep -v amdgpu $out/share_1.log
