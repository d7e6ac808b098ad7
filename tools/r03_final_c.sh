#!/bin/bash
# end-of-round pass C (after the last source change): PMC records + summaries of the three scenes, then the bench lines
set -e
bash tools/r03_final_a.sh
tag=r03z
bash tools/bench_all.sh $tag
python bench.py > gpurun_out/${tag}_bench_n1_run2.json 2>> gpurun_out/${tag}_bench_n1.err
cut -c1-300 gpurun_out/${tag}_bench_n1.json
