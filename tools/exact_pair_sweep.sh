#!/bin/bash
# residency sweep of k_exact_p (needs a DEV build: make DEV=1): unused dynamic LDS per block sets the resident blocks per CU
for lds in ${SWEEP:-0 8192 14336 20480 26624 34816 47104}; do
  echo "== FMGPU_DEV_EXACT_LDS=$lds"
  FMGPU_DEV_EXACT_LDS=$lds python bench.py --steps 5 --warmup 1 --no-cpu-baseline --only genome/exact/plain 2>&1 >/dev/null | grep "ms/step"
done
