"""Per-kernel statistics of the STEADY-STATE part of a rocprofv3 kernel trace of bench.py: only dispatches inside the last `window_ms` of the
trace (the timed / sustained steps: weight generation, packing and graph capture lie before it), so that the percentages are those of the running
pipeline and not of the set-up.

    python scripts/profile_steady.py <kernel_trace.csv> <window_ms> [out.csv]

Output columns: Name, Calls, TotalDurationNs, AverageNs, Percentage (of the window's summed kernel time), CallsPerStep when --steps is given."""
import csv, sys, collections
trace, window_ms = sys.argv[1], float(sys.argv[2])
out = sys.argv[3] if len(sys.argv) > 3 and not sys.argv[3].startswith("--") else None
steps = None
for a in sys.argv[3:]:
    if a.startswith("--steps="):
        steps = int(a.split("=")[1])
rows = list(csv.DictReader(open(trace)))
t_end = max(int(r["End_Timestamp"]) for r in rows)
t0 = t_end - int(window_ms * 1e6)
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    if int(r["Start_Timestamp"]) >= t0:
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
lines = [["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"] + (["CallsPerStep"] if steps else [])]
for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append([k, n, d, round(d / n, 1), round(100.0 * d / tot, 3)] + ([round(n / steps, 2)] if steps else []))
w = csv.writer(open(out, "w", newline="") if out else sys.stdout)
w.writerows(lines)
print(f"window {window_ms} ms: {sum(v[0] for v in agg.values())} dispatches, kernel time {tot / 1e6:.2f} ms ({tot / 1e6 / window_ms:.2f} x the window: streams overlap)", file=sys.stderr)
