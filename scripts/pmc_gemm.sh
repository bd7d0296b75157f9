#!/bin/bash
# usage: pmc_gemm.sh M N K ta tb   -> prints averaged SQ counters of the gemm256 kernel (two separate --pmc passes)
cd /tmp && export TMPDIR=/tmp
shape="$*"
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_WAVES"; do
  d=/tmp/pmc_out_$$_$RANDOM
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/scripts/bench_gemm_one.py $shape > /tmp/pmc_log.txt 2>&1
  python3 - "$d" "$shape" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")
if not f:
    print("no counter file", glob.glob(sys.argv[1] + "/*/*")); sys.exit(0)
rows = [r for r in csv.DictReader(open(f[0])) if "gemm" in r["Kernel_Name"]]
agg = {}
for r in rows:
    agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
print(sys.argv[2], {k: round(sum(v) / len(v)) for k, v in agg.items()})
PY
done
