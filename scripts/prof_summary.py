"""Summarise a rocprofv3 kernel_stats.csv: per-kernel total / avg, normalised per step."""
import csv, sys
path, steps = sys.argv[1], float(sys.argv[2])
rows = list(csv.DictReader(open(path)))
tot = sum(float(r['TotalDurationNs']) for r in rows if 'cast_params' not in r['Name'])
print(f'total kernel ms/step (excl. one-off init): {tot / 1e6 / steps:.2f}')
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 20]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('_ZN12_GLOBAL__N_1', '')[:70]
    print(f"{n:70s} calls/step={float(r['Calls']) / steps:7.1f} ms/step={float(r['TotalDurationNs']) / 1e6 / steps:7.2f} avg_us={float(r['AverageNs']) / 1e3:8.1f}")
