#!/bin/bash
# kernel traces of the single-cloud workloads (cfg5: ViT-g click session, cfg3: N=131072) + a full default bench line of the same box
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 400 python bench.py > $O/r04_bench14.json 2> /dev/null; echo "bench exit $?"
python - $O/r04_bench14.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], "sustained", d["sustained"]["value"], "frac", d["roofline"]["frac"], d.get("stage_ms"), d["parity"]["max_abs_err_mask_logits"])
PY
for wl in cfg5 cfg3; do
BENCH_EXTRA="--workload $wl" bash scripts/gpu_profile.sh r04_$wl 10 > $O/r04_prof_$wl.log 2>&1
f=$(find gpurun_out/prof_r04_$wl -name "*kernel_stats.csv" | head -1)
echo "== $wl"; head -30 "$f" | cut -c1-200
find gpurun_out/prof_r04_$wl -name "*kernel_trace.csv" -size +1M -delete
done
