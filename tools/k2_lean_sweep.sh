#!/bin/bash
# dev tool (library built with `make DEV=1`): k_scheme_lean variants.   usage (through gpurun): tools/k2_lean_sweep.sh <tag> "<VAR=val ...>" ...
TAG=$1; shift
export FMGPU_LIBRARY=${GRAFT_REPO_ROOT:-/root/repo}/fmindex-collection_amd/libfmgpu_dev.so   # the development build (make -C fmindex-collection_amd/csrc DEV=1): the shipped library reads no environment variable
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/leansweep_$TAG.log
: > $OUT
for cfg in "$@"; do
  echo "== $cfg" >> $OUT
  env $cfg python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --texts genome --only genome/k2/plain,genome/k2_151/plain 2>> $OUT > /dev/null || echo "FAILED $cfg" >> $OUT
done
grep -E "^==|bench.py: genome|FAILED" $OUT
