#!/bin/bash
# single-cloud GEMM configuration: cfg 9 (8 waves, one per CU, LDS epilogue) vs cfg 21 (4 waves, two per CU, register epilogue, mid-slab release)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_small_m_cfg.txt
: > $O
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-stage-times --no-mfma-probe --sustained-steps 0 --no-other-workloads"
for w in cfg3 cfg5; do
 for pen in 1.15 1.16 10; do
  for sk in "" 3 2; do
   if [ "$w" = "cfg3" ] && [ -n "$sk" ]; then continue; fi
   echo "== $w PEN9=$pen SPLITK=${sk:-auto}" >> $O
   PSAM_GEMM_PEN9=$pen ${sk:+PSAM_GEMM_SPLITK=$sk} $B --workload $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'avg_launch_ms', d['roofline']['avg_launch_ms'])" >> $O
  done
 done
done
cat $O
