#!/bin/bash
# socket power / clocks while the default bench runs (evidence for the power-wall argument)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
rocm-smi --showmaxpower --showpower 2>&1 | grep -v "^=\|^$" | head -8
(timeout 300 python bench.py --no-cpu-baseline --no-stage-times --no-gemm-profile --steps 4000 --sustained-steps 0 > gpurun_out/r04_power_bench.json 2>/dev/null) &
BP=$!
sleep 15
for i in $(seq 1 14); do rocm-smi --showpower --showclocks 2>&1 | grep -E "Socket|sclk" | sed -E 's/=+//g; s/\s+/ /g' | tr '\n' ' '; echo; sleep 2.5; done
wait $BP
python -c "
import json; d=json.loads(open('gpurun_out/r04_power_bench.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'])"
sleep 3; echo idle:; rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk" | tr '\n' ' '; echo
