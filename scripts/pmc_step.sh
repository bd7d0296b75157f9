#!/bin/bash
# HBM traffic of the step's kernels from the TCC counters: two separate --pmc passes (FETCH_SIZE, WRITE_SIZE cannot
# share a pass), eager launches (--no-graph) so every dispatch is attributed.  Output: gpurun_out/pmc_step.json
# usage (on the GPU box, from the repo root):  bash scripts/pmc_step.sh [bench args]
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python3 $root/bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-calibration --no-roofline --no-h2d "$@" > /tmp/pmc_$c.log 2>&1
  tail -2 /tmp/pmc_$c.log
done
python3 - <<'PY' > $out/pmc_step.json
import csv, glob, json, re
agg = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pmc_{c}/*/*counter_collection.csv")
    if not f:
        continue
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != c:
            continue
        k = re.sub(r"\((?!anonymous).*", "", r["Kernel_Name"]).replace("void ", "")
        a = agg.setdefault(k, {"FETCH_SIZE": [0, 0.0], "WRITE_SIZE": [0, 0.0]})
        a[c][0] += 1
        a[c][1] += float(r["Counter_Value"])
res = {}
for k, a in agg.items():
    n = max(a["FETCH_SIZE"][0], 1)
    # unit KiB; gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> x2 (MI355X_MICROARCH.md, HBM section)
    res[k] = {"launches": n, "fetch_bytes_per_launch": 2 * 1024 * a["FETCH_SIZE"][1] / n,
              "write_bytes_per_launch": 1024 * a["WRITE_SIZE"][1] / max(a["WRITE_SIZE"][0], 1)}
print(json.dumps(res, indent=1, sort_keys=True))
PY
echo done
