#!/bin/bash
# A/B of CU budgets for the persistent GEMM grids of the main stream (option gemm_cus) and of the side stream's tower (MR_SIDE_CUS):
# the same bench line per combination on ONE box.  Usage (GPU box): bash scripts/cu_split_ab.sh > gpurun_out/r4/cu_split.log
set -e
run() {
    echo "== main=$1 side=$2"
    MR_SIDE_CUS=$2 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-h2d --no-secondary $( [ "$1" != 0 ] && echo --option gemm_cus=$1 ) 2>/dev/null | python -c "import sys, json; j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'])"
}
run 0 0
run 0 64
run 0 96
run 0 128
run 192 64
run 224 64
run 192 128
run 0 0
