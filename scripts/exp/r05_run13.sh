#!/bin/bash
# pruned cooperative FPS: parity tests, then stage times with pruning off / on (cfg2 has B = 8: single-workgroup kernel, unchanged; cfg3 and cfg5 are cooperative)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "fps" > gpurun_out/pytest_fps.log 2>&1; echo "pytest exit $?"
tail -15 gpurun_out/pytest_fps.log
for pr in 0 1; do
  echo "== PSAM_FPS_PRUNE=$pr"
  PSAM_FPS_PRUNE=$pr STAGE_CFGS=cfg3,cfg5 timeout 300 python scripts/stage_times.py 2>&1 | grep -E "^cfg" | python -c "
import sys, json
for l in sys.stdin:
    tag, js = l.split(' ', 1); d = json.loads(js)
    print(tag, {k: d[k] for k in ('fps', 'knn', 'three_nn', 'encode_total', 'decode_total', 'wall_ms_per_pass') if k in d})
"
done
