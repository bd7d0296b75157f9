#!/bin/bash
# Same-box A/B of two builds of the library: scripts/ab_lib.sh OTHER.so -> attention microbench and the bench line's breakdown under MR_LIB=OTHER.so and
# under the shipped library, alternating (box-to-box differences of several percent make cross-call comparisons useless)
other=$1
summ='import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], "ms/step", round(d["ms_per_step"],2), "calib", d["config"]["calibration_tflops"], "breakdown", d["breakdown"]["ms_per_step"])'
mkdir -p gpurun_out/ab
for rep in 1 2; do
  MR_LIB=$other python scripts/attn_bench.py 2>&1 | grep -v amdgpu | sed "s/^/[other] /"
  python scripts/attn_bench.py 2>&1 | grep -v amdgpu | sed "s/^/[ours ] /"
done
for rep in 1 2; do
  MR_LIB=$other python bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-secondary --no-h2d > gpurun_out/ab/other$rep.json 2>/dev/null; python -c "$summ" gpurun_out/ab/other$rep.json "[other]"
  python bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-secondary --no-h2d > gpurun_out/ab/ours$rep.json 2>/dev/null; python -c "$summ" gpurun_out/ab/ours$rep.json "[ours ]"
done
