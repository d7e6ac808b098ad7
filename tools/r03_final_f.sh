#!/bin/bash
# end-of-round pass F: what the driver runs at round end, on the committed tree: smoke(), the GPU suite, bench.py with its defaults
set -e
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r03z_gpu_tests.log 2>&1 || { tail -30 gpurun_out/r03z_gpu_tests.log; exit 1; }
tail -1 gpurun_out/r03z_gpu_tests.log
python bench.py > gpurun_out/r03z_bench_default.json 2> gpurun_out/r03z_bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03z_bench_default.json").read().strip().splitlines()[-1])
print(d["metric"], d["value"], d["unit"], d["ms_per_step"], "frac", d["roofline"]["frac"], "of this build:", d["roofline"]["pmc_record_is_of_this_build"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
