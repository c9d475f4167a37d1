"""Summarises a rocprofv3 --kernel-trace CSV by (stream, queue): which HIP stream's kernels went through which HSA queue, when, and how busy."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
print("columns:", list(rows[0].keys()))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
agg = defaultdict(lambda: [0, 1 << 62, 0, 0])
for r in rows:
    k = (r.get("Stream_Id", "?"), r.get("Queue_Id", "?"))
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    a = agg[k]
    a[0] += 1; a[1] = min(a[1], s); a[2] = max(a[2], e); a[3] += e - s
print(f"{'stream':>8} {'queue':>6} {'launches':>9} {'first_ms':>10} {'last_ms':>10} {'busy_ms':>9}")
for k, a in sorted(agg.items(), key=lambda x: x[1][1]):
    print(f"{k[0]:>8} {k[1]:>6} {a[0]:>9} {a[1] / 1e6:>10.1f} {a[2] / 1e6:>10.1f} {a[3] / 1e6:>9.1f}")
