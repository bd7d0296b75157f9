#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_vcr
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_vcr -- python3 $root/scripts/bench_vcr.py large 4 10 > /tmp/vcr.log 2>&1
tail -1 /tmp/vcr.log
f=$(ls /tmp/prof_vcr/*/*kernel_stats.csv | head -1)
# bench_vcr.py large 4 10: one eager step (initialisation) + 3 warm-up + 10 timed graph replays launch kernels (the capture pass records, it does not run)
python3 $root/scripts/prof_summary.py $f --total-steps 14
