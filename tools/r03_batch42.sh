#!/bin/bash
# round-3 batch 42: tail hand-off, final defaults: GPU suite; slowest-rank tables of the BASELINE frames
set -e
out=gpurun_out/r03_batch42
mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1 || { tail -40 $out/gpu_tests.log; exit 1; }
tail -1 $out/gpu_tests.log
python tools/partition_time.py 1 2 4 8 > $out/partition.log 2>&1; grep "==" $out/partition.log
NX=1920 NY=1080 python tools/partition_time.py 1 2 4 8 > $out/partition_hd.log 2>&1; grep "==" $out/partition_hd.log
SCENE=final NX=800 NY=800 NS=200 python tools/partition_time.py 1 2 4 8 > $out/partition_final.log 2>&1; grep "==" $out/partition_final.log
SCENE=cornell NX=600 NY=600 NS=1000 python tools/partition_time.py 1 2 4 8 > $out/partition_cornell.log 2>&1; grep "==" $out/partition_cornell.log
SCENE=book1 NS=100 python tools/partition_time.py 1 8 > $out/partition_book1.log 2>&1; grep "==" $out/partition_book1.log
