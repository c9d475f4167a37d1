#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "flash_attention or key_split or gelu_block or cfg5 or giant or pipeline or session or c_caller or interp3_bitwise" > $O/r04_pytest_keysplit2.log 2>&1; echo "pytest exit $?"; tail -4 $O/r04_pytest_keysplit2.log
for rep in 1 2; do
timeout 300 python bench.py --workload cfg5 --no-cpu-baseline --sustained-steps 100 > $O/r04_bench18_cfg5_$rep.json 2> /dev/null; echo "bench exit $?"
python - $O/r04_bench18_cfg5_$rep.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d.get("stage_ms") or {}
print("cfg5", d["value"], d["ms_per_step"], "sustained", d["sustained"]["value"], "frac", d["roofline"]["frac"], {k:s.get(k) for k in ("vit_blocks","encode_total","two_way_decoder","ms_per_additional_click")})
PY
done
timeout 300 python bench.py --workload cfg5 --streams 1 --no-cpu-baseline --sustained-steps 100 > $O/r04_bench18_cfg5_s1.json 2> /dev/null; echo "bench (1 dense stream) exit $?"
python - $O/r04_bench18_cfg5_s1.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d.get("stage_ms") or {}
print("cfg5 --streams 1", d["value"], d["ms_per_step"], "sustained", d["sustained"]["value"])
PY
