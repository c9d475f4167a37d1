#!/bin/bash
# steady-state kernel statistics of the three workloads (rocprofv3 --kernel-trace), round 6 start
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for w in cfg2 cfg3 cfg5; do
  rm -rf /tmp/kt_$w
  rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$w -o kt -- python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline --no-stage-times --no-gemm-profile --no-mfma-probe --sustained-steps 0 --no-other-workloads > gpurun_out/r06_prof_$w.json 2>/dev/null
  f=$(find /tmp/kt_$w -name '*kernel_trace.csv' | head -1)
  ms=$(python -c "import json; d=json.loads(open('gpurun_out/r06_prof_$w.json').read().strip().splitlines()[-1]); print(d['ms_per_step']*30)")
  python scripts/profile_steady.py $f $ms gpurun_out/r06_steady_$w.csv --steps=30 2> gpurun_out/r06_steady_$w.txt
  cat gpurun_out/r06_steady_$w.txt; head -25 gpurun_out/r06_steady_$w.csv | cut -c1-200
done
