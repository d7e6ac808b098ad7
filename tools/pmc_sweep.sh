#!/bin/bash
# VALU instruction counts per configuration: runs tools/sweep.py (1 round) under rocprofv3 --pmc and prints, per
# dispatch in launch order, instructions, active-lane average and duration.  Usage: tools/pmc_sweep.sh <tag> <ns> cfg...
tag=$1; ns=$2; shift; shift
out=gpurun_out/pmcsweep_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $out/pmc -- python3 tools/sweep.py --ns $ns --rounds 1 "$@" > $out/sweep.log 2>&1
python3 - "$out" "$@" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]; cfgs = sys.argv[2:]
f = glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True)[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "rt_render" not in r["Kernel_Name"]: continue
    rows.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
for (d, c), cfg in zip(rows.items(), cfgs):
    print(f"{cfg:55s} VALU {c['SQ_INSTS_VALU']:.3e} lanes {c['SQ_THREAD_CYCLES_VALU']/c['SQ_INSTS_VALU']:5.1f} SALU {c['SQ_INSTS_SALU']:.2e} LDS {c['SQ_INSTS_LDS']:.2e} "
          f"wait {c['SQ_WAIT_ANY']/c['SQ_WAVE_CYCLES']:.2f} waitinst {c['SQ_WAIT_INST_ANY']/c['SQ_WAVE_CYCLES']:.2f} active {c['SQ_ACTIVE_INST_ANY']/c['SQ_WAVE_CYCLES']:.2f}")
PY
cat $out/sweep.log | grep -v amdgpu.ids
