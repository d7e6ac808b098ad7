#!/bin/bash
# LDS / wait counters per configuration (second counter set for tools/pmc_sweep.sh)
tag=$1; ns=$2; shift; shift
out=gpurun_out/pmcsweep2_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $out/pmc -- python3 tools/sweep.py --ns $ns --rounds 1 "$@" > $out/sweep.log 2>&1
python3 - "$out" "$@" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]; cfgs = sys.argv[2:]
f = glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True)[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "rt_render" not in r["Kernel_Name"]: continue
    rows.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
for (d, c), cfg in zip(rows.items(), cfgs):
    print(f"{cfg:50s} LDS_IDX_ACTIVE {c['SQ_LDS_IDX_ACTIVE']:.3e} BANK_CONFLICT {c['SQ_LDS_BANK_CONFLICT']:.3e} INSTS_LDS {c['SQ_INSTS_LDS']:.3e} ACTIVE_INST_LDS {c['SQ_ACTIVE_INST_LDS']:.3e} WAIT_INST_LDS {c['SQ_WAIT_INST_LDS']:.3e} WAVE_CYCLES {c['SQ_WAVE_CYCLES']:.3e} BUSY {c['SQ_BUSY_CYCLES']:.3e} WAIT_ANY {c['SQ_WAIT_ANY']:.3e}")
PY
grep "min " $out/sweep.log
