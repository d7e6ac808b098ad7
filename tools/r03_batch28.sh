#!/bin/bash
# round-3 batch 28: tail hand-off: per-launch timelines of the headline frame without it, with threshold 0, with the default
set -e
out=gpurun_out/r03_batch28
mkdir -p $out
export TMPDIR=/tmp
for cfg in "off:handoff=0" "zero:handoff_pixels=0" "auto:" "p16k:handoff_pixels=16384"; do
  tag=${cfg%%:*}; opts=${cfg#*:}
  RT_OPTS=$opts rocprofv3 --kernel-trace --output-format csv -d $out/trace_$tag -- python3 tools/one_frame.py > $out/$tag.log 2>&1
  python3 tools/timeline_from_trace.py $out/trace_$tag > $out/timeline_$tag.txt 2>&1 || true
  tail -1 $out/$tag.log; cat $out/timeline_$tag.txt
  rm -rf $out/trace_$tag
done
