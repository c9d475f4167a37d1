#!/bin/bash
# key-split attention: tests, then cfg5 bench A/B (PSAM_ATTN_KEYSPLIT) on one box
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -s -k "flash_attention or key_split or gelu_block or cfg5 or giant_slim" > $O/r04_pytest_keysplit.log 2>&1; echo "pytest exit $?"; grep -E "key split|passed|failed|Error" $O/r04_pytest_keysplit.log | tail -12
for ksp in 0 1 0 1; do
PSAM_ATTN_KEYSPLIT=$ksp timeout 300 python bench.py --workload cfg5 --no-cpu-baseline --sustained-steps 100 > $O/r04_bench17_cfg5_$ksp.json 2> /dev/null; echo "bench exit $?"
python - $O/r04_bench17_cfg5_$ksp.json $ksp <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d.get("stage_ms") or {}
print("cfg5 keysplit", sys.argv[2], d["value"], d["ms_per_step"], "sustained", d["sustained"]["value"], "frac", d["roofline"]["frac"], {k:s.get(k) for k in ("vit_blocks","encode_total","two_way_decoder","ms_per_additional_click")})
PY
done
