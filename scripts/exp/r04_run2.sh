#!/bin/bash
# round 4, GPU call 2: register-only epilogue (transposed accumulators): bitwise check, timing, bench A/B; lgkmcnt ordering probe
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 400 python scripts/exp/r04_tr.py > $O/r04_tr.log 2>&1; echo "tr exit $?"
PSAM_GEMM_TR=1 timeout 400 python scripts/gemm_p_bench.py 21,51,55,57 > $O/r04_gemm_p_tr1.log 2>&1; echo "gemm_p tr1 exit $?"
PSAM_GEMM_TR=0 timeout 400 python scripts/gemm_p_bench.py 21,51,55,57 > $O/r04_gemm_p_tr0.log 2>&1; echo "gemm_p tr0 exit $?"
for rep in 1 2; do for m in 0 1; do
  PSAM_GEMM_TR=$m timeout 300 python bench.py --no-cpu-baseline --sustained-steps 100 > $O/r04_bench_tr${m}_$rep.json 2> $O/r04_bench_tr${m}_$rep.err; echo "bench tr$m rep$rep exit $?"
done; done
PSAM_GEMM_TR=1 timeout 600 python bench.py --sustained-steps 100 > $O/r04_bench_tr1_parity.json 2> $O/r04_bench_tr1_parity.err; echo "bench tr1 parity exit $?"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/lgkm scripts/exp/r04_lgkm_order.hip && timeout 300 /tmp/lgkm > $O/r04_lgkm_order.txt 2>&1; echo "lgkm exit $?"
cat $O/r04_lgkm_order.txt
cat $O/r04_tr.log | grep -v amdgpu.ids
for t in 1 0; do echo "== TR=$t"; grep -v "^check" $O/r04_gemm_p_tr$t.log | tail -14; done
for f in $O/r04_bench_tr*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], "sustained", d.get("sustained",{}).get("value"), "frac", d["roofline"]["frac"], "parity", d.get("parity",{}).get("max_abs_err_mask_logits"), "stage", d.get("stage_ms"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
