#!/bin/bash
# every committed round-4 profile in one gpurun call (each record: kernel trace + the separate PMC passes); usage: tools/profile_all_r04.sh [tags ...] (default: all)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
ALL=("exact_plain genome/exact/plain k_exact_p" "exact_lut12 genome/exact/plain+lut12 k_exact_p" "exact_single genome/exact/single k_exact_a" "k2_plain genome/k2/plain k_scheme_lean"
     "k2_151_plain genome/k2_151/plain k_scheme_lean" "locate_plain genome/locate/plain k_locate_coop" "edit_plain genome/k2_edit/plain k_scheme_fast_edit --with-edit"
     "protein_wavelet protein/exact/wavelet k_exact_s" "protein_tree protein/exact/tree k_exact_m" "protein_wide protein_wide/exact/wavelet k_exact_s"
     "protein_xl protein_xl/exact/wavelet k_exact_s")
for spec in "${ALL[@]}"; do
  set -- $spec
  if [ -n "$WANT" ] && ! echo " $WANT " | grep -q " $1 "; then continue; fi
  tag=$1; rec=$2; ker=$3; shift; shift; shift
  tools/profile_r04.sh $tag $rec $ker "$@" > gpurun_out/prof_$tag.log 2>&1
  echo "$tag: $(tail -n 1 gpurun_out/prof_$tag.log)"
done
