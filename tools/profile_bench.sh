#!/bin/bash
# Profiles one bench.py configuration on the GPU box: kernel trace + stats, then PMC passes (each in its own run,
# never combined with tracing domains).  Usage: tools/profile_bench.sh <tag> [bench args...]
set -e
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
# which sources these counters belong to (bench.py compares it with the tree it runs from)
cat accelerated-ray-tracer_amd/csrc/* | sha1sum | cut -c1-16 > $out/csrc_sha1.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py "$@" --no-cpu-baseline > $out/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $out/pmc_sq1 -- python3 bench.py "$@" --no-cpu-baseline > $out/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM --output-format csv -d $out/pmc_sq2 -- python3 bench.py "$@" --no-cpu-baseline > $out/pmc_sq2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_IFETCH SQ_WAVE32_INSTS GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq3 -- python3 bench.py "$@" --no-cpu-baseline > $out/pmc_sq3.log 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py "$@" --no-cpu-baseline > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py "$@" --no-cpu-baseline > $out/pmc_write.log 2>&1
rocprofv3 -L > $out/counters_list.txt 2>&1 || true
