#!/bin/bash
# memory-pipeline counters for a bench workload (dev tool): tools/profile_mem.sh <tag> <bench args>
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline $@"
rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/pmc_tlb -- $B > $OUT/pmc_tlb.log 2>&1 &&
rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum --output-format csv -d $OUT/pmc_lat -- $B > $OUT/pmc_lat.log 2>&1 &&
rocprofv3 --pmc TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum --output-format csv -d $OUT/pmc_ta -- $B > $OUT/pmc_ta.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.log 2>&1 &&
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY GRBM_TA_BUSY GRBM_TC_BUSY GRBM_EA_BUSY --output-format csv -d $OUT/pmc_grbm -- $B > $OUT/pmc_grbm.log 2>&1
echo "profile exit=$?"
