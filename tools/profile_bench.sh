#!/bin/bash
# rocprofv3 passes over bench.py (dev tool; run through gpurun).  usage: tools/profile_bench.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $OUT/trace.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/pmc_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/pmc_write.log 2>&1 &&
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/pmc_tcc.log 2>&1 &&
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_tcp -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/pmc_tcp.log 2>&1
echo "profile exit=$?"
find $OUT -name "*.csv" | head -40
