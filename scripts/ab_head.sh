# Same-box A/B of the attention kernels against the PREVIOUS commit's attention.hip: build merlot_reserve_amd/libdiag_head.so first
#   git show HEAD~1:merlot_reserve_amd/csrc/attention.hip > /tmp/a.hip && bash scripts/build_diag.sh head ... (see scripts/build_diag.sh); then gpurun -- bash scripts/ab_head.sh
mkdir -p gpurun_out/r5
python -m pytest tests/test_kernels_gpu.py -x -q -k attention > gpurun_out/r5/attn_tests.txt 2>&1 || { tail -30 gpurun_out/r5/attn_tests.txt; exit 1; }
tail -2 gpurun_out/r5/attn_tests.txt
for rep in 1 2; do
  MR_LIB=$PWD/merlot_reserve_amd/libdiag_head.so python scripts/attn_bench.py 2>&1 | grep -v amdgpu | sed "s/^/[head] /"
  python scripts/attn_bench.py 2>&1 | grep -v amdgpu | sed "s/^/[new ] /"
done
