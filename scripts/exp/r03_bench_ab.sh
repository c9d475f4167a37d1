#!/bin/bash
# GPU box: bench.py A/B over the GEMM pick mode (PSAM_GEMM_PP) and the number of dense streams.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r03_bench_ab.log; : > $OUT
for rep in 1 2; do
for mode in ${MODES:-0 1 2 3}; do for st in ${STREAMS:-2 3}; do
  echo "== PSAM_GEMM_PP=$mode streams=$st rep=$rep" >> $OUT
  PSAM_GEMM_PP=$mode timeout 600 python bench.py --streams $st --slots 3 --steps 30 --sustained-steps 100 --no-cpu-baseline --no-stage-times 2>> gpurun_out/r03_bench_ab.err | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print(r['value'], r['ms_per_step'], 'sustained', r['sustained']['value'], 'frac', r['roofline']['frac'] if r['roofline'] else None)" >> $OUT
done; done; done
cat $OUT
