# A/B of non-temporal epilogue stores (variants nt / ntc2: scripts/build_g3_variants.sh nt:-DMR_G3_AUX_C=2,-DMR_G3_AUX_C2=2 ntc2:-DMR_G3_AUX_C2=2): no effect measured
mkdir -p gpurun_out/r5
for v in "" nt ntc2 "" nt; do
  if [ -z "$v" ]; then unset MR_LIB; else export MR_LIB=$PWD/merlot_reserve_amd/variants/$v/libmreserve_hip.so; fi
  echo "== variant '$v'"
  python scripts/bench_gemm_epi.py 2>&1 | grep -E "bn=256" | grep -E "gelu\+c2|aux|fwd qkv    bias\+rot"
done
