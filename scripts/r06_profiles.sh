#!/bin/bash
# Round-6 evidence in one gpurun call -> gpurun_out/ (copied to profiles/ afterwards): rocprofv3 kernel summaries of BASELINE configs
# 2-5, the PMC passes (HBM traffic, MFMA utilisation; counters only with --kernel-trace), mr_gemm beside the vendor library.
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
bash scripts/round_profile.sh r06_base --no-secondary > gpurun_out/prof_r06_base.log 2>&1; echo base done
bash scripts/round_profile.sh r06_large --model large --no-secondary --no-h2d > gpurun_out/prof_r06_large.log 2>&1; echo large done
bash scripts/round_profile.sh r06_large_resadapt --model large --resadapt --records-per-gpu 2 --no-h2d --no-secondary > gpurun_out/prof_r06_large_resadapt.log 2>&1; echo resadapt done
bash scripts/prof_vcr.sh > gpurun_out/prof_r06_vcr_large_b4_summary.txt 2>&1; echo vcr done
bash scripts/pmc_step.sh --no-secondary > gpurun_out/pmc_step_r06.log 2>&1; echo pmc traffic done
bash scripts/pmc_mfma.sh r06_base_b4 --no-secondary > gpurun_out/pmc_mfma_r06_base.log 2>&1; echo pmc mfma base done
bash scripts/pmc_mfma.sh r06_large_b4 --model large --no-secondary > gpurun_out/pmc_mfma_r06_large.log 2>&1; echo pmc mfma large done
python3 scripts/bench_vs_blas.py > gpurun_out/r06_gemm_vs_hipblaslt.txt 2>&1; echo blas done
