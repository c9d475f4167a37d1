#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "click_session or variants_through or row_ln_512 or register_epilogue or graph_pipeline" > $O/r04_pytest_gpu_c.log 2>&1; echo "pytest exit $?"
tail -25 $O/r04_pytest_gpu_c.log
timeout 900 python bench.py --workload cfg3 --sustained-steps 100 > $O/r04_bench_cfg3.json 2> $O/r04_bench_cfg3.err; echo "bench cfg3 exit $?"; tail -3 $O/r04_bench_cfg3.err
timeout 900 python bench.py --workload cfg5 --sustained-steps 100 > $O/r04_bench_cfg5.json 2> $O/r04_bench_cfg5.err; echo "bench cfg5 exit $?"; tail -3 $O/r04_bench_cfg5.err
timeout 600 python bench.py --data ply --no-cpu-baseline --sustained-steps 100 > $O/r04_bench_ply.json 2> $O/r04_bench_ply.err; echo "bench ply exit $?"; tail -3 $O/r04_bench_ply.err
timeout 600 python bench.py --no-cpu-baseline --sustained-steps 100 > $O/r04_bench_uniform.json 2> $O/r04_bench_uniform.err; echo "bench uniform exit $?"
for f in $O/r04_bench_cfg3.json $O/r04_bench_cfg5.json $O/r04_bench_ply.json $O/r04_bench_uniform.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["value"], d["ms_per_step"], "sustained", d.get("sustained",{}).get("value"), "frac", (d.get("roofline") or {}).get("frac"), "parity", d.get("parity"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "click", d.get("ms_per_additional_click"))
    print("   stage", d.get("stage_ms")); print("   tok", {k:(v.get("ms"), v.get("us_per_iteration")) for k,v in (d.get("tokenizer") or {}).items()})
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
