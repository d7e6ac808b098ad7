#!/bin/bash
# end-of-round measurement pass, part B: every bench configuration, the partition tables, scene creation, per-launch timelines,
# wave-end records, tier pace, the whole GPU test suite
set -e
tag=r03z
export TMPDIR=/tmp
bash tools/bench_all.sh $tag
python bench.py > gpurun_out/${tag}_bench_n1_run2.json 2>> gpurun_out/${tag}_bench_n1.err
echo "bench done"; cat gpurun_out/${tag}_bench_n1.json | cut -c1-400
python tools/partition_time.py > gpurun_out/${tag}_partition_random_1200x800_500.log 2>&1
NX=1920 NY=1080 python tools/partition_time.py > gpurun_out/${tag}_partition_random_1920x1080_500.log 2>&1
SCENE=cornell NX=600 NY=600 NS=1000 python tools/partition_time.py > gpurun_out/${tag}_partition_cornell_600x600_1000.log 2>&1
SCENE=final NX=800 NY=800 NS=200 python tools/partition_time.py > gpurun_out/${tag}_partition_final_800x800_200.log 2>&1
echo "partitions done"; grep -h "==" gpurun_out/${tag}_partition_*.log
python tools/scene_create_time.py > gpurun_out/${tag}_scene_create.log 2>&1
for cfg in "whole:1" "half:2" "quarter:4" "eighth:8"; do
  t=${cfg%%:*}; st=${cfg#*:}
  STRIDE=$st rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${tag}_trace_$t -- python3 tools/one_frame.py > gpurun_out/${tag}_one_frame_$t.log 2>&1
  python3 tools/timeline_from_trace.py gpurun_out/${tag}_trace_$t > gpurun_out/${tag}_timeline_$t.txt 2>&1 || true
  rm -rf gpurun_out/${tag}_trace_$t
done
D=accelerated-ray-tracer_amd/lib/librt_mi355x_diag.so
RT_LIB_OVERRIDE=$D python tools/diag_wave_ends.py 500 > gpurun_out/${tag}_wave_ends_whole.txt 2>&1
STRIDE=8 RT_LIB_OVERRIDE=$D python tools/diag_wave_ends.py 500 > gpurun_out/${tag}_wave_ends_eighth.txt 2>&1
RT_LIB_OVERRIDE=$D python tools/diag_tier_pace.py > gpurun_out/${tag}_tier_pace.txt 2>&1
STRIDE=8 RT_LIB_OVERRIDE=$D python tools/diag_tier_pace.py >> gpurun_out/${tag}_tier_pace.txt 2>&1
SCENE=final NX=800 NY=800 NS=200 STRIDE=8 RT_LIB_OVERRIDE=$D python tools/diag_tier_pace.py >> gpurun_out/${tag}_tier_pace.txt 2>&1
RT_LIB_OVERRIDE=$D python tools/diag_stages.py 200 > gpurun_out/${tag}_diag_stage_cycles.txt 2>&1
echo "diag done"
python -m pytest tests -q -m gpu > gpurun_out/${tag}_gpu_tests.log 2>&1; tail -2 gpurun_out/${tag}_gpu_tests.log
