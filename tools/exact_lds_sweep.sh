#!/bin/bash
# dev tool (library built with `make DEV=1`): exact-search kernels for several residencies (unused dynamic LDS per block).  usage (through gpurun): tools/exact_lds_sweep.sh <tag> <record> [lds values]
TAG=$1; REC=$2; shift; shift
export FMGPU_LIBRARY=${GRAFT_REPO_ROOT:-/root/repo}/fmindex-collection_amd/libfmgpu_dev.so   # the development build (make -C fmindex-collection_amd/csrc DEV=1): the shipped library reads no environment variable
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ldssweep_$TAG.log
: > $OUT
for lds in ${@:-0 16384 24576 32768 40960}; do
  export FMGPU_DEV_EXACT_LDS=$lds
  echo "== lds=$lds" >> $OUT
  python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --only $REC 2>> $OUT > /dev/null || echo "FAILED" >> $OUT
done
grep -E "^==|bench.py: |FAILED" $OUT
