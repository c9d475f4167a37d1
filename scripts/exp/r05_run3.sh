#!/bin/bash
# Round 5, GPU call 3: why the persistent stream-K GEMM lost in the bench (658 vs 802 clouds/s): per-shape times of its modes, cycle budget per wave,
# bench with the CU reservation; standalone cfg5 / cfg3 bench lines beside the legs of the default run.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r05_run3; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s --tb=short -p no:cacheprovider -x -k "streamk" > $O/pytest.log 2>&1; echo "pytest exit $?" | tee -a $O/pytest.log
grep -E "^\[|passed|failed|Error|assert" $O/pytest.log | tail -8
PSAM_HIP_LIB=scripts/exp/libpointsam_abl.so timeout 600 python scripts/exp/r05_streamk_timing.py > $O/streamk_timing.txt 2>&1; echo "timing exit $?"
cat $O/streamk_timing.txt | tail -40
timeout 600 python scripts/gemm_p_bench.py 21,90,91,93 > $O/gemm_sweep.txt 2>&1; echo "sweep exit $?"
grep -v "^check.*ok$" $O/gemm_sweep.txt | tail -16
for sk in 0 1; do
  PSAM_GEMM_STREAMK=$sk timeout 300 python bench.py --no-cpu-baseline --no-other-workloads --sustained-steps 100 --no-stage-times > $O/bench_sk$sk.json 2> $O/bench_sk$sk.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_sk$sk.json").read().strip().splitlines()[-1])
    print("PSAM_GEMM_STREAMK=$sk", d["value"], "sustained", d["sustained"]["value"], "frac", d["roofline"]["frac"], "launch ms", d["roofline"]["avg_launch_ms"])
except Exception as e:
    print("PSAM_GEMM_STREAMK=$sk failed", e); print(open("$O/bench_sk$sk.err").read()[-1500:])
PY
done
for wl in cfg5 cfg3; do
  PSAM_GEMM_STREAMK=0 timeout 300 python bench.py --workload $wl --no-cpu-baseline --sustained-steps 0 > $O/bench_$wl.json 2> $O/bench_$wl.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$wl.json").read().strip().splitlines()[-1])
    print("$wl standalone", d["value"], d["ms_per_step"], d["step_ms"], d["config"]["batches_in_flight"])
except Exception as e:
    print("$wl failed", e); print(open("$O/bench_$wl.err").read()[-1500:])
PY
done
