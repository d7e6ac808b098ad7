#!/bin/bash
# end-of-round measurement pass, part A: rocprofv3 kernel trace + PMC passes of the three BASELINE scenes (tools/profile_bench.sh)
set -e
tag=r03z
bash tools/profile_bench.sh ${tag}_headline --steps 4 --warmup 1
echo "headline done"
bash tools/profile_bench.sh ${tag}_cornell --scene cornell --nx 600 --ny 600 --ns 1000 --steps 4 --warmup 1
echo "cornell done"
bash tools/profile_bench.sh ${tag}_final --scene final --nx 800 --ny 800 --ns 200 --steps 4 --warmup 1
echo "final done"
for s in headline cornell final; do
  python3 tools/summarize_profile.py gpurun_out/prof_${tag}_$s > gpurun_out/${tag}_${s}_summary.txt 2>&1 || true
done
python3 tools/pmc_to_json.py gpurun_out/prof_${tag}_headline random_scene_1200x800_500 > gpurun_out/pmc_random_scene_1200x800_500.json
python3 tools/pmc_to_json.py gpurun_out/prof_${tag}_cornell cornell_600x600_1000 > gpurun_out/pmc_cornell_600x600_1000.json
python3 tools/pmc_to_json.py gpurun_out/prof_${tag}_final final_800x800_200 > gpurun_out/pmc_final_800x800_200.json
# the raw rocprof directories are large: keep the csv tables only
find gpurun_out/prof_${tag}_* -type f ! -name "*.csv" ! -name "*.txt" ! -name "*.log" -delete 2>/dev/null || true
du -sh gpurun_out/prof_${tag}_* | tail -3
head -c 1500 gpurun_out/pmc_random_scene_1200x800_500.json
