#!/bin/bash
# A/B of library builds in one process-per-build, interleaved: tools/ab_variants.sh "<sweep args>" lib1.so lib2.so ...
# ("shipped" = the in-tree library).  Experiment tooling; the timings quoted in DESIGN.md come from bench.py.
set -e
args="$1"; shift
for round in 1 2; do
  for lib in "$@"; do
    echo "== $lib"
    if [ "$lib" = shipped ]; then python tools/sweep.py $args ""; else RT_LIB_OVERRIDE=$lib python tools/sweep.py $args ""; fi
  done
done
