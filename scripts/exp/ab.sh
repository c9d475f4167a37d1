#!/bin/bash
# A/B two builds of the library on the same box: bench + per-kernel rocprof stats
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; mkdir -p gpurun_out/ab; export TMPDIR=/tmp
for rep in 1 2; do
for v in new old; do
  if [ $v = old ]; then export PSAM_HIP_LIB=$R/scripts/exp/ab/libpointsam_hip_old.so; else unset PSAM_HIP_LIB; fi
  python bench.py --no-cpu-baseline --no-gemm-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v bf16x6', d['value'], d['ms_per_step'])"
  python bench.py --no-cpu-baseline --no-gemm-profile --precision f32 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v f32', d['value'], d['ms_per_step'])"
done; done
for v in new old; do
  if [ $v = old ]; then export PSAM_HIP_LIB=$R/scripts/exp/ab/libpointsam_hip_old.so; else unset PSAM_HIP_LIB; fi
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab/$v -o p -- python $R/bench.py --no-cpu-baseline --no-gemm-profile --no-pipeline --steps 5 > /dev/null 2>&1)
  f=$(find gpurun_out/ab/$v -name '*kernel_stats.csv' | head -1); echo "== $v"; head -14 $f | cut -c1-150
done
