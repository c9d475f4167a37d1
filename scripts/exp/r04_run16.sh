#!/bin/bash
# split-K with the in-kernel fix-up: tests, then cfg5 / cfg3 bench lines with and without it on the same box
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "splitk or split_k or gelu_block or cfg5 or giant or cfg3 or c_caller or twoway or decoder" > $O/r04_pytest_splitk.log 2>&1; echo "pytest exit $?"; tail -5 $O/r04_pytest_splitk.log
for wl in cfg5 cfg3; do
for fx in 0 1 0 1; do
PSAM_GEMM_SPLITK_FIXUP=$fx timeout 300 python bench.py --workload $wl --no-cpu-baseline --sustained-steps 100 > $O/r04_bench16_${wl}_$fx.json 2> /dev/null; echo "bench exit $?"
python - $O/r04_bench16_${wl}_$fx.json $wl $fx <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d.get("stage_ms") or {}
print(sys.argv[2], "fixup", sys.argv[3], d["value"], d["ms_per_step"], "sustained", d["sustained"]["value"], "frac", d["roofline"]["frac"], {k:s.get(k) for k in ("vit_blocks","encode_total","two_way_decoder","ms_per_additional_click")})
PY
done; done
