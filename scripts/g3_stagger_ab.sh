# A/B of the start-phase stagger experiment of the ping-pong GEMM (variants st1/st2/st4: scripts/build_g3_variants.sh st1:-DMR_G3_STAGGER=1 ...); DESIGN.md section 3
mkdir -p gpurun_out/r5
for v in "" st1 st2 st4 ""; do
  if [ -z "$v" ]; then unset MR_LIB; else export MR_LIB=$PWD/merlot_reserve_amd/variants/$v/libmreserve_hip.so; fi
  echo "== variant '$v'"
  python scripts/bench_gemm_epi.py 2>&1 | grep -E "bn=(256|192)" | grep -E "gelu\+c2|aux|bias\+rot|residual"
done
