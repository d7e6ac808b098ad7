#!/bin/bash
# round-3 batch 35: tier sizes of the shares with the tail hand-off on, second pass (rank 0 of 8 / 2; rank 3 of 8 as a check)
set -e
out=gpurun_out/r03_batch35
mkdir -p $out
mk() { echo "tier_auto=0,tier1_pixels=$1,tier1_factor_x10=$2,tier1_depth=$3,heavy_factor_x10=$4,sparse_factor_x10=$5,sparse_wg_percent=$7,sparse_work_percent=$6"; }
STRIDE=8 ROUNDS=3 python tools/share_sweep.py "" "$(mk 8192 20 4 15 20 40 80)" "$(mk 8192 25 4 15 25 40 80)" "$(mk 6144 20 4 15 20 40 80)" "$(mk 12288 20 4 15 20 40 80)" "$(mk 8192 20 4 20 20 40 80)" "$(mk 8192 20 4 15 30 40 80)" "$(mk 8192 20 4 15 20 20 80)" "$(mk 8192 20 4 15 20 40 50)" "$(mk 8192 17 4 15 17 40 80)" > $out/share_8.log 2>&1; grep -v amdgpu $out/share_8.log
STRIDE=8 FIRST=3 ROUNDS=3 python tools/share_sweep.py "" "$(mk 8192 20 4 15 20 40 80)" "$(mk 8192 25 4 15 25 40 80)" "$(mk 6144 20 4 15 20 40 80)" > $out/share_8_rank3.log 2>&1; grep -v amdgpu $out/share_8_rank3.log
STRIDE=2 ROUNDS=3 python tools/share_sweep.py "" "tier_auto=0" "$(mk 1536 40 3 20 40 5 80)" "$(mk 1536 50 3 20 50 5 80)" "$(mk 1024 40 3 20 40 5 80)" "$(mk 2048 40 3 20 40 5 80)" "$(mk 1536 40 3 20 40 5 35)" "$(mk 1536 40 3 20 40 10 80)" "$(mk 1536 40 2 20 40 5 80)" > $out/share_2.log 2>&1; grep -v amdgpu $out/share_2.log
STRIDE=4 ROUNDS=3 python tools/share_sweep.py "" "$(mk 8192 20 4 15 20 40 80)" "$(mk 8192 25 4 15 25 40 80)" "$(mk 6144 20 4 15 20 40 80)" "$(mk 8192 20 4 20 20 40 80)" "$(mk 8192 20 4 15 20 20 80)" "$(mk 3072 30 4 20 30 10 80)" > $out/share_4.log 2>&1; grep -v amdgpu $out/share_4.log
