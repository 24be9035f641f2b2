#!/bin/bash
# Collects the rocprofv3 evidence for the headline bench (run on the GPU box from the repo root):
#   1. kernel trace + stats of the default bench command (every step's rows kept: HBM regime) and of
#      the two-tensor variant (--obs-ring 2: rows stay in the last-level cache)
#   2. FETCH_SIZE and WRITE_SIZE in separate counter passes of both (eager launches, fewer steps)
# Usage: bash profiles/run_profile.sh <tag>
set -e
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for regime in kept llc; do
  ring=32; [ $regime = llc ] && ring=2
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$regime -o trace -- python3 bench.py --steps 2048 --warmup 128 --obs-ring $ring --no-cpu-baseline --no-extra-legs --spinup-ms 0 > $OUT/bench_trace_$regime.json 2> $OUT/trace_$regime.err
  timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$regime -o fetch -- python3 bench.py --steps 64 --warmup 8 --obs-ring $ring --no-cpu-baseline --no-extra-legs --spinup-ms 0 > $OUT/bench_fetch_$regime.json 2> $OUT/fetch_$regime.err
  timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$regime -o write -- python3 bench.py --steps 64 --warmup 8 --obs-ring $ring --no-cpu-baseline --no-extra-legs --spinup-ms 0 > $OUT/bench_write_$regime.json 2> $OUT/write_$regime.err
done
find $OUT -name "*.csv" | head -20
