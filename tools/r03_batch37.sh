#!/bin/bash
# round-3 batch 37: Book-2 final 800x800 @ 200 with the tail hand-off: tier sizes and hand-off threshold, whole frame and 1/8 share
set -e
out=gpurun_out/r03_batch37
mkdir -p $out
mk() { echo "tier_auto=0,tier1_pixels=$1,tier1_factor_x10=$2,tier1_depth=$3,heavy_factor_x10=$4,sparse_factor_x10=$5,sparse_wg_percent=$7,sparse_work_percent=$6"; }
export SCENE=final NX=800 NY=800 NS=200
STRIDE=1 ROUNDS=2 python tools/share_sweep.py "" "handoff=0" "handoff_pixels=2048" "handoff_pixels=4096" "handoff_pixels=8192" "handoff_pixels=16384" "handoff_pixels=32768" "handoff_poll=4" "handoff_poll=8" "tier_kernel=0" "tier_kernel=0,handoff=0" "$(mk 256 70 1 20 40 5 35)" "$(mk 64 100 1 20 40 5 35)" "$(mk 1024 40 2 20 40 5 35)" > $out/whole.log 2>&1; grep -v amdgpu $out/whole.log
STRIDE=8 ROUNDS=2 python tools/share_sweep.py "" "handoff=0" "handoff_pixels=1024" "handoff_pixels=4096" "handoff_pixels=8192" "$(mk 8192 20 4 15 15 40 80)" "$(mk 4096 30 4 20 30 20 80)" "$(mk 2048 40 4 20 40 5 80)" "$(mk 256 70 1 20 40 5 35)" "$(mk 8192 20 4 15 15 40 80),handoff_pixels=8192" "$(mk 2048 40 4 20 40 5 80),handoff_pixels=8192" > $out/eighth.log 2>&1; grep -v amdgpu $out/eighth.log
