# Stamps of the dQ / dK-dV kernels + A/B of a forward variant (libdiag_qfirst.so, libdiag_head.so, libdiag_attnstamps{2,3}.so built by scripts/build_diag.sh)
mkdir -p gpurun_out/r5
MR_LIB=$PWD/merlot_reserve_amd/libdiag_attnstamps2.so python scripts/attn_dq_stamps.py 2>&1 | grep -v amdgpu
MR_LIB=$PWD/merlot_reserve_amd/libdiag_attnstamps3.so python scripts/attn_dkv_stamps.py 2>&1 | grep -v amdgpu
for rep in 1 2; do
  MR_LIB=$PWD/merlot_reserve_amd/libdiag_head.so python scripts/attn_bench.py 2>&1 | grep -E "ViT|S=256|base joint" | sed "s/^/[head] /"
  MR_LIB=$PWD/merlot_reserve_amd/libdiag_qfirst.so python scripts/attn_bench.py 2>&1 | grep -E "ViT|S=256|base joint" | sed "s/^/[qfst] /"
  python scripts/attn_bench.py 2>&1 | grep -E "ViT|S=256|base joint" | sed "s/^/[new ] /"
done
