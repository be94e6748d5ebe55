#!/bin/bash
# Round profile: rocprofv3 kernel-trace summary of the default bench workload, then the PMC passes (each in a run of
# its own, counters only: HBM traffic FETCH_SIZE / WRITE_SIZE, MFMA busy).  Usage (on the GPU box, from the repo root):
#   bash scripts/profile_round.sh r2      -> gpurun_out/prof_r2/*
set -e
TAG=${1:-r3}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing"
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $OUT -o ${TAG}_trace --output-format csv -- $BENCH > $OUT/${TAG}_trace_bench.log 2>&1
BENCH1="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE -d $OUT -o ${TAG}_fetch --output-format csv -- $BENCH1 > $OUT/${TAG}_fetch.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE -d $OUT -o ${TAG}_write --output-format csv -- $BENCH1 > $OUT/${TAG}_write.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT -o ${TAG}_mfma --output-format csv -- $BENCH1 > $OUT/${TAG}_mfma.log 2>&1
# issue / wait / LDS breakdown per kernel (SQ block, 8 counters)
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU -d $OUT -o ${TAG}_sq --output-format csv -- $BENCH1 > $OUT/${TAG}_sq.log 2>&1
ls -la $OUT | head -30
