#!/bin/bash
# Round 5, GPU call 4: the persistent continuous-stream GEMM (csrc/gemm_f16x3c.hip): bitwise tests, cycle budget, sweep against cfg 21, bench A/B;
# and why the cfg3 / cfg5 legs of the default run are slower than standalone runs.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r05_run4; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s --tb=short -p no:cacheprovider -x -k "continuous or register_epilogue_bitwise" > $O/pytest.log 2>&1; echo "pytest exit $?" | tee -a $O/pytest.log
grep -E "^\[|passed|failed|Error|assert" $O/pytest.log | tail -8
PSAM_HIP_LIB=scripts/exp/libpointsam_abl.so timeout 600 python scripts/exp/r05_continuous_timing.py > $O/continuous_timing.txt 2>&1; echo "timing exit $?"
cat $O/continuous_timing.txt | tail -30
timeout 600 python scripts/gemm_p_bench.py 21,94,95 > $O/gemm_sweep.txt 2>&1; echo "sweep exit $?"
grep -v "^check.*ok$" $O/gemm_sweep.txt | tail -16
for c in 0 1; do
  PSAM_GEMM_CONTINUOUS=$c timeout 300 python bench.py --no-cpu-baseline --no-other-workloads --sustained-steps 100 --no-stage-times > $O/bench_c$c.json 2> $O/bench_c$c.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_c$c.json").read().strip().splitlines()[-1])
    print("PSAM_GEMM_CONTINUOUS=$c", d["value"], "sustained", d["sustained"]["value"], "frac", d["roofline"]["frac"], "launch ms", d["roofline"]["avg_launch_ms"])
except Exception as e:
    print("PSAM_GEMM_CONTINUOUS=$c failed", e); print(open("$O/bench_c$c.err").read()[-1500:])
PY
done
# the cfg5 leg, now a fresh process
PSAM_GEMM_CONTINUOUS=0 timeout 300 python bench.py --no-cpu-baseline --other-workloads cfg5 --sustained-steps 0 --no-stage-times --no-gemm-profile --steps 5 > $O/bench_leg5.json 2> $O/bench_leg5.err
python - <<PY
import json
try:
    d = json.loads(open("$O/bench_leg5.json").read().strip().splitlines()[-1])
    l = d["other_workloads"]["cfg5"]
    print("cfg5 leg right after a 5-step main run, no oracle:", l["value"], l["ms_per_step"])
except Exception as e:
    print("leg5 failed", e); print(open("$O/bench_leg5.err").read()[-1500:])
PY
