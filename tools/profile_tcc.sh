#!/bin/bash
# two rocprofv3 counter passes (L2 requests, HBM write size) over a bench command (dev tool); usage: tools/profile_tcc.sh <tag> <bench args...>
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary $@"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pmc_tcc -- $B > $OUT/pmc_tcc.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.log 2>&1
echo "profile $TAG exit=$?"
