#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "twoway or decoder or pipeline or session or c_caller or predictor or demo_server or forward_eval or against_oracle" > $O/r04_pytest_fork.log 2>&1; echo "pytest exit $?"; tail -4 $O/r04_pytest_fork.log
for wl in cfg2 cfg5; do for fk in 0 1 0 1; do
PSAM_TWOWAY_FORK=$fk timeout 300 python bench.py --workload $wl --no-cpu-baseline --sustained-steps 100 > $O/r04_bench19_${wl}_$fk.json 2> /dev/null; echo "bench exit $?"
python - $O/r04_bench19_${wl}_$fk.json $wl $fk <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d.get("stage_ms") or {}
print(sys.argv[2], "fork", sys.argv[3], d["value"], d["ms_per_step"], "sustained", d["sustained"]["value"], {k:s.get(k) for k in ("two_way_decoder","decode_total","ms_per_additional_click")})
PY
done; done
