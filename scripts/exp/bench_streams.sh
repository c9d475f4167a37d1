#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for cfgs in "2 3" "3 4" "2 4" "1 2" "3 3"; do
  set -- $cfgs
  v=$(python bench.py --no-cpu-baseline --no-stage-times --no-gemm-profile --steps 30 --streams $1 --slots $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['step_ms'])")
  echo "streams=$1 slots=$2 -> $v"
done
