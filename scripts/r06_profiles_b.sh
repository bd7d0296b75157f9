#!/bin/bash
# Round-6 evidence, second call: the PMC passes (HBM traffic, MFMA utilisation; counters only with --kernel-trace) and mr_gemm beside the vendor library
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
bash scripts/pmc_step.sh --no-secondary > gpurun_out/pmc_step_r06.log 2>&1; echo pmc traffic done
bash scripts/pmc_mfma.sh r06_base_b4 --no-secondary > gpurun_out/pmc_mfma_r06_base.log 2>&1; echo pmc mfma base done
bash scripts/pmc_mfma.sh r06_large_b4 --model large --no-secondary > gpurun_out/pmc_mfma_r06_large.log 2>&1; echo pmc mfma large done
python3 scripts/bench_vs_blas.py > gpurun_out/r06_gemm_vs_hipblaslt.txt 2>&1; echo blas done
