#!/bin/bash
# End-of-round evidence: (1) rocprofv3 kernel trace of the bench, full stats + steady-state window, (2) PMC traffic passes, (3) smoke()
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r05 -o bench -- python $R/bench.py --steps 60 --warmup 3 --no-cpu-baseline --no-gemm-profile --no-stage-times --sustained-steps 0 --no-other-workloads > $R/gpurun_out/prof_r05.bench.json 2> $R/gpurun_out/prof_r05.err; echo "prof exit $?"
cd $R
t=$(find gpurun_out/prof_r05 -name "*kernel_trace.csv" | head -1)
python scripts/profile_steady.py "$t" 400 gpurun_out/r05_steady_state_kernel_stats.csv --steps=40 2>&1 | tail -2
head -25 gpurun_out/r05_steady_state_kernel_stats.csv
f=$(find gpurun_out/prof_r05 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r05_rocprofv3_kernel_stats.csv
find gpurun_out/prof_r05 -name "*kernel_trace.csv" -delete
cat gpurun_out/prof_r05.bench.json | cut -c1-400
bash scripts/gpu_pmc.sh r05 > gpurun_out/pmc_r05.log 2>&1; tail -3 gpurun_out/pmc_r05.log
python scripts/pmc_to_traffic.py r05 gpurun_out/r05_traffic.json | head -30
for d in fetch write; do f=$(find gpurun_out/pmc_${d}_r05 -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python - "$f" gpurun_out/r05_pmc_${d}_summary.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]; acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
w = csv.writer(open(sys.argv[2], "w", newline="")); w.writerow(["Kernel", "Launches", "CounterSum", "CounterPerLaunch"])
for k, (v, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]): w.writerow([k, n, round(v, 1), round(v / n, 3)])
PY
done
rm -rf gpurun_out/pmc_fetch_r05 gpurun_out/pmc_write_r05
