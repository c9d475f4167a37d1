#!/bin/bash
# round 4, GPU call 1: right-sized ping-pong tiles -- bitwise check, per-shape / per-layer timing, row-pitch test, bench A/B, kernel trace
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 300 python scripts/exp/r04_rightsize.py > $O/r04_rightsize.log 2>&1; echo "rightsize exit $?"
PSAM_GEMM_PP=4 timeout 300 python scripts/exp/r04_rightsize.py > $O/r04_rightsize_pp4.log 2>&1; echo "rightsize pp4 exit $?"
timeout 400 python scripts/gemm_p_bench.py 21,51,55,57,62,63,64 > $O/r04_gemm_p.log 2>&1; echo "gemm_p exit $?"
PAD=64 timeout 300 python scripts/gemm_p_bench.py 21,51,62 > $O/r04_gemm_p_pad64.log 2>&1; echo "gemm_p pad exit $?"
for rep in 1 2; do for m in 0 4; do
  PSAM_GEMM_PP=$m timeout 300 python bench.py --no-cpu-baseline --sustained-steps 100 > $O/r04_bench_pp${m}_$rep.json 2> $O/r04_bench_pp${m}_$rep.err; echo "bench pp$m rep$rep exit $?"
done; done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/r04_trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --sustained-steps 0 --steps 10 --no-stage-times --no-gemm-profile > /tmp/r04_trace.log 2>&1; echo "trace exit $?"
cd $GRAFT_REPO_ROOT
f=$(find /tmp/r04_trace -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, gzip
rows = list(csv.DictReader(open(sys.argv[1])))
# keep the columns needed for a timeline: name (shortened), start, end, stream/queue
out = open("gpurun_out/r04_trace_compact.csv", "w")
out.write("name,start,end,queue,grid,wg,lds,vgpr\n")
for r in rows:
    n = r.get("Kernel_Name", "")[:60].replace(",", ";")
    out.write(f'{n},{r.get("Start_Timestamp")},{r.get("End_Timestamp")},{r.get("Queue_Id")},{r.get("Grid_Size")},{r.get("Workgroup_Size")},{r.get("LDS_Block_Size")},{r.get("VGPR_Count")}\n')
out.close()
print("trace rows", len(rows))
PY
gzip -f gpurun_out/r04_trace_compact.csv 2>/dev/null
tail -30 $O/r04_rightsize.log; tail -12 $O/r04_rightsize_pp4.log; grep -v "^check" $O/r04_gemm_p.log | tail -22; grep -v "^check" $O/r04_gemm_p_pad64.log | tail -12
for f in $O/r04_bench_pp*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], "sustained", d.get("sustained",{}).get("value"), "frac", d["roofline"]["frac"], "stage", d.get("stage_ms"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
