#!/bin/bash
# Round 5, GPU call 2: the persistent stream-K GEMM (csrc/gemm_f16x3s.hip) -- parity tests, kernel sweep against cfg 21, bench A/B -- and the
# GraphPipeline slot/stream binding fix (cfg5 leg NaN of call 1).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r05_run2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -q -s --tb=short -p no:cacheprovider -x \
  -k "streamk or register_epilogue_bitwise or in_kernel_fixups or pipelines_bitwise_equal or graph_pipeline_matches or gemm_f16x3_split_k or c_eva_block" > $O/pytest.log 2>&1; echo "pytest exit $?" | tee -a $O/pytest.log
grep -E "^\[|passed|failed|Error|assert" $O/pytest.log | tail -20
timeout 600 python scripts/gemm_p_bench.py 21,90 > $O/gemm_sweep.txt 2>&1; echo "sweep exit $?"
grep -v "^check.*ok$" $O/gemm_sweep.txt | tail -20
for sk in 0 1; do
  PSAM_GEMM_STREAMK=$sk timeout 300 python bench.py --no-cpu-baseline --no-other-workloads --sustained-steps 100 --no-stage-times > $O/bench_sk$sk.json 2> $O/bench_sk$sk.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_sk$sk.json").read().strip().splitlines()[-1])
    print("PSAM_GEMM_STREAMK=$sk", d["value"], "sustained", d["sustained"]["value"], "frac", d["roofline"]["frac"], "launch ms", d["roofline"]["avg_launch_ms"], "slots", d["config"]["batches_in_flight"])
except Exception as e:
    print("PSAM_GEMM_STREAMK=$sk failed", e); print(open("$O/bench_sk$sk.err").read()[-1500:])
PY
done
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench exit $?"
python - <<PY
import json
try:
    d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
    print("default", d["value"], d["roofline"]["frac"], d["parity"]["max_abs_err_mask_logits"], d["parity"]["ok"])
    for w, l in d["other_workloads"].items():
        print(w, l["value"], l["ms_per_step"], l.get("roofline", {}).get("frac"), l["parity"]["max_abs_err_mask_logits"], l["parity"]["ok"])
except Exception as e:
    print("default bench failed", e); print(open("$O/bench_default.err").read()[-2000:])
PY
