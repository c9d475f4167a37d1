#!/bin/bash
# bench.py under different tile-choice parameters (GPU box)
cd "${GRAFT_REPO_ROOT:-.}"
for cfgs in "8 1.8" "8 1.6" "0 1.8" "0 1.6" "8 2.2"; do
  set -- $cfgs
  v=$(PSAM_GEMM_RESERVE_CUS=$1 PSAM_GEMM_SHARE=$2 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])")
  echo "reserve=$1 share=$2 -> $v"
done
