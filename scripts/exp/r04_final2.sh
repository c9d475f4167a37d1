#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/r04_gpu_tests_final2.log 2>&1; echo "pytest exit $?"; tail -3 $O/r04_gpu_tests_final2.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 500 python bench.py > $O/r04_bench_final2.json 2> /dev/null; echo "bench exit $?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_bench_final2.json").read().strip().splitlines()[-1]); s=d.get("stage_ms") or {}
print(d["value"], d["ms_per_step"], "sustained", d["sustained"]["value"], "frac", d["roofline"]["frac"], "parity", d["parity"]["max_abs_err_mask_logits"], d["parity"]["ok"], "cpu", d["cpu_baseline"]["value"])
PY
