#!/bin/bash
# end-of-round pass D: the bench lines once more, now that profiles/pmc_*.json are the records of these sources
set -e
tag=r03z
bash tools/bench_all.sh $tag
python bench.py > gpurun_out/${tag}_bench_n1_run2.json 2>> gpurun_out/${tag}_bench_n1.err
python - <<'PY'
import json
for f in ("gpurun_out/r03z_bench_n1.json", "gpurun_out/r03z_bench_n1_run2.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["pmc_record_is_of_this_build"], d["cpu_baseline"]["value"])
PY
