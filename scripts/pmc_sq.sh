#!/bin/bash
# SQ issue / wait breakdown of one conv micro-benchmark (two PMC passes, <= 8 SQ counters each).
# usage: bash scripts/pmc_sq.sh <tag> <bench_conv.py args...>   -> gpurun_out/pmc_<tag>/{a,b}_counter_collection.csv
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/pmc_$tag
mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS -d $out -o a --output-format csv -- python3 scripts/bench_conv.py "$@" --iters 3 > $out/a.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE -d $out -o b --output-format csv -- python3 scripts/bench_conv.py "$@" --iters 3 > $out/b.log 2>&1 || exit 1
