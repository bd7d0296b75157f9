#!/bin/bash
# Round-6 evidence, first call: rocprofv3 kernel summaries of BASELINE configs 2-5 -> gpurun_out/ (scripts/install_profiles.py r06 copies them to profiles/)
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
bash scripts/round_profile.sh r06_base --no-secondary > gpurun_out/prof_r06_base.log 2>&1; echo base done
bash scripts/round_profile.sh r06_large --model large --no-secondary --no-h2d > gpurun_out/prof_r06_large.log 2>&1; echo large done
bash scripts/round_profile.sh r06_large_resadapt --model large --resadapt --records-per-gpu 2 --no-h2d --no-secondary > gpurun_out/prof_r06_large_resadapt.log 2>&1; echo resadapt done
bash scripts/prof_vcr.sh > gpurun_out/prof_r06_vcr_large_b4_summary.txt 2>&1; echo vcr done
