#!/bin/bash
# round-3 batch 22: progressive windows and binary PPM: GPU test suite, the drop-in executable's new flags
set -e
out=gpurun_out/r03_batch22
mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1 || { tail -40 $out/gpu_tests.log; exit 1; }
tail -2 $out/gpu_tests.log
E=accelerated-ray-tracer_amd/lib/rayTracer
$E --scene bouncing --nx 96 --ny 64 --ns 24 > $out/one.ppm 2> $out/one.err
$E --scene bouncing --nx 96 --ny 64 --ns 24 --progressive 7 > $out/prog.ppm 2> $out/prog.err
cmp $out/one.ppm $out/prog.ppm && echo "progressive PPM identical"
$E --scene bouncing --nx 96 --ny 64 --ns 24 --p6 > $out/one6.ppm 2>> $out/one.err
head -c 15 $out/one6.ppm | head -2; ls -la $out/*.ppm; tail -3 $out/prog.err
python tools/sweep.py --ns 500 --rounds 3 "" > $out/headline.log 2>&1; grep min $out/headline.log
