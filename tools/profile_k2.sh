#!/bin/bash
# rocprofv3 counter passes over the k=2 bench (dev tool)
set -o pipefail
TAG=${1:-r01_k2}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --workload k2 --steps 2 --warmup 1 --no-cpu-baseline $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1 &&
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc_sq2 -- $B > $OUT/pmc_sq2.log 2>&1 &&
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pmc_tcc -- $B > $OUT/pmc_tcc.log 2>&1 &&
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/pmc_tcp -- $B > $OUT/pmc_tcp.log 2>&1
echo "profile exit=$?"
