#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_zs
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_zs -- python3 $root/scripts/bench_zeroshot.py base fp32 > /tmp/zs.log 2>&1
tail -1 /tmp/zs.log
f=$(ls /tmp/prof_zs/*/*kernel_stats.csv | head -1)
python3 $root/scripts/prof_summary.py $f 7 12
