#!/bin/bash
# Same-box A/B of an environment switch of the host code on the bench line: bash scripts/ab_env.sh NAME=VALUE [...]
run() {
    env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-h2d --no-secondary 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(j['ms_per_step'], 3))"
}
echo "default:"; run A=1
for o in "$@"; do echo "$o:"; run $o; done
echo "default:"; run A=1
