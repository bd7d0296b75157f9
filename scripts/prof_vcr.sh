#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_vcr
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_vcr -- python3 $root/scripts/bench_vcr.py large 4 10 > /tmp/vcr.log 2>&1
tail -1 /tmp/vcr.log
f=$(ls /tmp/prof_vcr/*/*kernel_stats.csv | head -1)
python3 $root/scripts/prof_summary.py $f 14 22
