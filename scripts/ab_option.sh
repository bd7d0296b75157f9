#!/bin/bash
# Same-box A/B of a library option on the bench line (base + the three secondary workloads): bash scripts/ab_option.sh NAME=VALUE [NAME=VALUE ...]
# prints ms / step of the default run, of each option run, and of the default again.
run() {
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-h2d "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(j['ms_per_step'], 3), {k: round(v['ms_per_step'], 2) for k, v in j['config'].get('secondary', {}).items()})"
}
echo "default:"; run
for o in "$@"; do echo "--option $o:"; run --option $o; done
echo "default:"; run
