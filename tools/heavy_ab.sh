#!/bin/bash
# A/B of the "offer only from heavy reads" threshold in the table-driven kernels (needs a DEV build: make DEV=1)
set -e
ONLY=genome/k2/tables,genome/k2_151/tables,uniform/k2/tables,genome/k2_edit/tables,uniform/k2_edit/tables
for rep in 1 2; do
  for f in 0 128; do
    echo "== FMGPU_DEV_FLAGS=$f rep $rep"
    FMGPU_DEV_FLAGS=$f python bench.py --steps 5 --warmup 1 --with-edit --only $ONLY 2>&1 >/dev/null | grep "ms/step"
  done
done
