#!/bin/bash
# A/B of two library builds on the three BASELINE scenes (+ cornell_smoke): tools/ab_scenes.sh other.so
# Experiment tooling (timings quoted in DESIGN.md come from bench.py).
set -e
other=$1
python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/ab_parity.log 2>&1 || { tail -30 gpurun_out/ab_parity.log; exit 1; }
tail -1 gpurun_out/ab_parity.log
bash tools/ab_variants.sh '--scene cornell --nx 600 --ny 600 --ns 1000 --rounds 2' $other shipped > gpurun_out/ab_cornell.log 2>&1
bash tools/ab_variants.sh '--scene cornell_smoke --nx 600 --ny 600 --ns 200 --rounds 2' $other shipped > gpurun_out/ab_smoke.log 2>&1
bash tools/ab_variants.sh '--scene final --nx 800 --ny 800 --ns 200 --rounds 2' $other shipped > gpurun_out/ab_final.log 2>&1
bash tools/ab_variants.sh '--ns 500 --rounds 3' $other shipped > gpurun_out/ab_head.log 2>&1
grep -h "==\|min" gpurun_out/ab_cornell.log gpurun_out/ab_smoke.log gpurun_out/ab_final.log gpurun_out/ab_head.log | cut -c1-12,55-135
