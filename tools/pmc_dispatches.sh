#!/bin/bash
# Per-dispatch SQ counters of the render kernel for a list of configurations (tools/sweep.py, one round), in launch
# order: a frame with the cost-aware schedule is three dispatches (samples [0,8), [8,32), [32,ns)).
# Usage: tools/pmc_dispatches.sh <tag> <ns> cfg...        (cfg = comma-separated rt_set_option pairs, "" = defaults)
tag=$1; ns=$2; shift; shift
out=gpurun_out/pmcd_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $out/pmc -- python3 tools/sweep.py $SWEEP_ARGS --ns $ns --rounds 1 "$@" > $out/sweep.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True)[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "rt_render" not in r["Kernel_Name"]: continue
    rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
for d, c in rows.items():
    print(f"dispatch {d:4d} VALU {c['SQ_INSTS_VALU']:.4e} lanes {c['SQ_THREAD_CYCLES_VALU']/c['SQ_INSTS_VALU']:5.1f} SALU {c['SQ_INSTS_SALU']:.3e} LDS {c['SQ_INSTS_LDS']:.3e} "
          f"wavecyc {c['SQ_WAVE_CYCLES']:.3e} wait {c['SQ_WAIT_ANY']/c['SQ_WAVE_CYCLES']:.2f} waitinst {c['SQ_WAIT_INST_ANY']/c['SQ_WAVE_CYCLES']:.2f} active {c['SQ_ACTIVE_INST_ANY']/c['SQ_WAVE_CYCLES']:.2f}")
PY
grep -v amdgpu.ids $out/sweep.log
