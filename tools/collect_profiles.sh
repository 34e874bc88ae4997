#!/bin/bash
# Usage (build container, repo root): tools/collect_profiles.sh <tag>  -- copies what tools/refresh_profiles.sh <tag> left under
# gpurun_out/<tag>/ into profiles/<tag>_* (the tracked, judged copies)
tag=$1
o=gpurun_out/$tag
cp $o/bench_line.json profiles/${tag}_bench_line.json
for pair in "prof_bench:bench" "prof_dedup:dedup" "prof_train:train_step"; do
  src=${pair%%:*}; dst=${pair##*:}
  f=$(find $o/$src -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" profiles/${tag}_${dst}_kernel_stats.csv
done
cp $o/pmc_hbm.json profiles/${tag}_pmc_hbm.json
cp $o/pmc_mfma.json profiles/${tag}_pmc_mfma.json
cp $o/pmc_mfma_alt2.json profiles/${tag}_pmc_mfma_alt2.json
cp $o/pmc_train.json profiles/${tag}_train_hbm_pmc.json
cp $o/train_step_profile_bf16.txt profiles/${tag}_train_step_profile_bf16.txt
cp $o/mfma_issue_rate.txt profiles/${tag}_mfma_issue_rate.txt
ls -la profiles/${tag}_*
