#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/membench
mkdir -p $OUT
$R/tools/membench 3221225472 > $OUT/run_3g.log 2>&1 && $R/tools/membench 134217728 > $OUT/run_128m.log 2>&1 &&
cd /tmp && export TMPDIR=/tmp &&
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc -- $R/tools/membench 3221225472 > $OUT/pmc.log 2>&1
echo exit=$?
