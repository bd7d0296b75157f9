#!/bin/bash
# Every (partition, panel) variant of scripts/xcd_walk_ab.py: time, then bytes fetched out of L2 per launch (rocprofv3 --pmc FETCH_SIZE; counters only with
# --kernel-trace, the program itself after --).  Output: gpurun_out/r6/xcd_walk_ab.txt      usage: bash scripts/xcd_walk_ab.sh
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r6; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
: > $out/xcd_walk_ab.txt
for v in "0 0" "8 2" "8 3" "8 4" "8 6" "4 2" "4 3" "4 6" "2 3" "2 0"; do
  set -- $v
  python3 $root/scripts/xcd_walk_ab.py $1 $2 >> $out/xcd_walk_ab.txt 2>/dev/null
  rm -rf /tmp/xw_pmc
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/xw_pmc -- python3 $root/scripts/xcd_walk_ab.py $1 $2 1 > /tmp/xw_pmc.log 2>&1
  python3 - "$1" "$2" >> $out/xcd_walk_ab.txt <<'PY'
import csv, glob, re, sys
agg = {}
for f in glob.glob('/tmp/xw_pmc/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != 'FETCH_SIZE' or 'gemm' not in r['Kernel_Name']:
            continue
        k = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1; a[1] += float(r['Counter_Value'])
for k, (n, v) in sorted(agg.items()):
    print(f'xpx={sys.argv[1]} xpanel={sys.argv[2]}   FETCH {k:40s} {2 * 1024 * v / n / 1e6:7.1f} MB / launch ({n} launches; x2 per the gfx950 correction)')
PY
done
cat $out/xcd_walk_ab.txt
