#!/bin/bash
# Collects the rocprofv3 evidence for the headline bench (run on the GPU box from the repo root):
#   1. kernel trace + stats of the default bench command
#   2. FETCH_SIZE and WRITE_SIZE in separate counter passes (eager launches, fewer steps)
# Usage: bash profiles/run_profile.sh <tag>
set -e
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py --steps 2048 --warmup 128 --no-cpu-baseline --no-rows-kept > $OUT/bench_trace.json 2> $OUT/trace.err
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-rows-kept > $OUT/bench_fetch.json 2> $OUT/fetch.err
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-rows-kept > $OUT/bench_write.json 2> $OUT/write.err
find $OUT -name "*.csv" | head -20
