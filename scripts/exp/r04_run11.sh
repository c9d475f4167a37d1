#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
ALT=$PWD/point_sam_amd/csrc/libpointsam_hip_gelu_as.so
for rep in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --sustained-steps 100 > $O/r04_bench11_erf_$rep.json 2> /dev/null; echo "bench erf exit $?"
  PSAM_LIB_PATH=$ALT timeout 300 python bench.py --no-cpu-baseline --sustained-steps 100 > $O/r04_bench11_as_$rep.json 2> /dev/null; echo "bench as exit $?"
done
PSAM_LIB_PATH=$ALT timeout 600 python bench.py --sustained-steps 0 --no-stage-times --no-gemm-profile > $O/r04_bench11_as_parity.json 2> /dev/null; echo "bench as parity exit $?"
for f in $O/r04_bench11_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], "sustained", (d.get("sustained") or {}).get("value"), "parity", (d.get("parity") or {}).get("max_abs_err_mask_logits"), "stage", d.get("stage_ms"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
