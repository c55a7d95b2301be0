#!/bin/bash
# rocprofv3 evidence for ONE bench record: kernel trace + stats, then HBM counters in separate --pmc passes (never combined with a trace domain)
# usage (through gpurun): tools/profile_r04.sh <tag> <record id, e.g. genome/exact/plain> <kernel name, e.g. k_exact_a> [extra bench args]     outputs under gpurun_out/prof_<tag>/
set -o pipefail
TAG=$1; REC=$2; KERNEL=$3; shift; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --only $REC $@"
export FMGPU_BENCH_RECORDS=$OUT/bench_records.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1 && cp $OUT/bench_records.json $OUT/bench_records_trace.json &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1 &&
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -- $B > $OUT/pmc_tcc.log 2>&1 &&
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.log 2>&1 &&
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- $B > $OUT/pmc_sq2.log 2>&1 &&
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $OUT/pmc_tcp -- $B > $OUT/pmc_tcp.log 2>&1
echo "profile $TAG exit=$?"
grep -h '"metric"' $OUT/trace.log | tail -1 > $OUT/bench_line.json
# keep what is committed small: the per-kernel stats table and the counter rows of the search kernels
cd $R && python3 tools/summarize_round.py $OUT $KERNEL $OUT/summary.json > $OUT/summary.log 2>&1
find $OUT -name "*_kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT -type f \( -name "*.db" -o -name "*_kernel_trace.csv" -o -name "*counter_collection.csv" -o -name "*agent_info.csv" \) -delete
