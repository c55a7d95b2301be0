#!/bin/bash
# HIP API + kernel trace of the default bench step loop (no counters): where a step's host-side time goes
# usage (through gpurun): tools/profile_api.sh <tag> <bench args...>     outputs under gpurun_out/api_<tag>/
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/api_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-secondary "$@" > $OUT/trace.log 2>&1
echo "exit=$?"
grep -h '"metric"' $OUT/trace.log | tail -1 > $OUT/bench_line.json
find $OUT/trace -name "*hip_api_stats.csv" -exec head -25 {} \;
find $OUT/trace -name "*kernel_stats.csv" -exec head -12 {} \;
find $OUT/trace -name "*_trace.csv" -size +20M -delete
