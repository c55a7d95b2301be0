#!/bin/bash
# every committed round-3 profile in one gpurun call (each record: kernel trace + the separate PMC passes)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for spec in "exact_plain genome/exact/plain k_exact_p" "exact_single genome/exact/single k_exact_a" "k2_plain genome/k2/plain k_scheme_lean" "k2_151_plain genome/k2_151/plain k_scheme_lean" "locate_plain genome/locate/plain k_locate_fused" \
            "exact_tables genome/exact/tables k_exact_kstep" "k2_tables genome/k2/tables k_scheme_fast" "protein_wavelet protein/exact/wavelet k_exact_s" "protein_tree protein/exact/tree k_exact_m" "protein_wide protein_wide/exact/wavelet k_exact_s"; do
  set -- $spec
  tools/profile_r03.sh $1 $2 $3 > gpurun_out/prof_$1.log 2>&1
  echo "$1: $(tail -n 1 gpurun_out/prof_$1.log)"
done
