#!/bin/bash
# kNN band kernel: parity tests, then stage times with the four-pass kernel and with the band kernel (cfg2 + cfg3)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_kernels.log 2>&1; echo "pytest exit $?"
tail -15 gpurun_out/pytest_kernels.log
for band in 0 1; do
  echo "== PSAM_KNN_BAND=$band"
  PSAM_KNN_BAND=$band STAGE_CFGS=cfg2,cfg3 timeout 300 python scripts/stage_times.py 2>&1 | grep -E "^cfg" | python -c "
import sys, json
for l in sys.stdin:
    tag, js = l.split(' ', 1); d = json.loads(js)
    print(tag, {k: d[k] for k in ('fps', 'knn', 'three_nn', 'encode_total', 'wall_ms_per_pass')})
"
  cp gpurun_out/stage_times.json gpurun_out/stage_times_band$band.json
done
