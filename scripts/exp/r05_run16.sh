#!/bin/bash
# (1) randomized pruned-FPS test; (2) attention <4,2> with a standing priority for the odd hardware wave slot (libpsam_attnprio.so) vs the production library:
# the packed kernel alone (r04_attn.py, variant 1 column) and the bench, interleaved
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "fps_pruned" 2>&1 | tail -4
for rep in 1 2; do
  for lib in prod prio; do
    [ $lib = prio ] && export PSAM_LIB_PATH=$PWD/scripts/exp/libpsam_attnprio.so || unset PSAM_LIB_PATH
    echo "== $lib (rep $rep)"
    timeout 200 python scripts/exp/r04_attn.py 2>&1 | grep -E "^B=" | head -2
    timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-gemm-profile --no-stage-times --sustained-steps 0 --no-other-workloads 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['parity']['max_abs_err_mask_logits'])"
  done
done
