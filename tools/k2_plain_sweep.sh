#!/bin/bash
# dev tool (library built with `make DEV=1`): k = 2 on the plain index for several residencies / kernels.   usage (through gpurun): tools/k2_plain_sweep.sh <tag> [bench args]
TAG=$1; shift
export FMGPU_LIBRARY=${GRAFT_REPO_ROOT:-/root/repo}/fmindex-collection_amd/libfmgpu_dev.so   # the development build (make -C fmindex-collection_amd/csrc DEV=1): the shipped library reads no environment variable
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/sweep_$TAG.log
: > $OUT
for cfg in "bpc=7" "bpc=6" "bpc=5" "bpc=4" "old=1"; do
  unset FMGPU_DEV_BPC FMGPU_DEV_FLAGS
  case $cfg in
    bpc=*) export FMGPU_DEV_BPC=${cfg#bpc=};;
    old=1) export FMGPU_DEV_FLAGS=$((1<<30));;
  esac
  echo "== $cfg" >> $OUT
  python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --texts genome --only genome/k2/plain,genome/k2_151/plain "$@" 2>> $OUT > /dev/null || echo "FAILED $cfg" >> $OUT
done
grep -E "^==|bench.py: genome|FAILED" $OUT
