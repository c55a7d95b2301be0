#!/bin/bash
# dev tool (library built with `make DEV=1`): locate on the plain index for several residencies (unused dynamic LDS).  usage (through gpurun): tools/locate_sweep.sh <tag>
TAG=$1; shift
export FMGPU_LIBRARY=${GRAFT_REPO_ROOT:-/root/repo}/fmindex-collection_amd/libfmgpu_dev.so   # the development build (make -C fmindex-collection_amd/csrc DEV=1): the shipped library reads no environment variable
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/locsweep_$TAG.log
: > $OUT
for lds in 0 16384 24576 36864 49152; do
  export FMGPU_DEV_LOCATE_LDS=$lds
  echo "== lds=$lds" >> $OUT
  python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-protein --texts genome --only genome/locate/plain "$@" 2>> $OUT > /dev/null || echo "FAILED" >> $OUT
done
grep -E "^==|bench.py: genome|FAILED" $OUT
