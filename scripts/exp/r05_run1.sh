#!/bin/bash
# Round 5, GPU call 1: the new parity tests, the default bench (with the cfg3 / cfg5 legs), where the production GEMM's cycles go (timing build),
# and the sweep of the ping-pong configurations incl. the new two-workgroups-per-CU ones; then the bench with those configurations switched in.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r05_run1; mkdir -p $O
(rocminfo | grep -E "Marketing Name|gfx9" | head -2; nproc) > $O/box.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -s --tb=short -p no:cacheprovider -k "free_running or cfg3_gap or click_session" > $O/pytest_new.log 2>&1; echo "pytest exit $?" | tee -a $O/pytest_new.log
grep -E "^\[|passed|failed|Error" $O/pytest_new.log | tail -20
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench exit $?"
tail -c 3000 $O/bench_default.json; tail -3 $O/bench_default.err
PSAM_HIP_LIB=scripts/exp/libpointsam_abl.so timeout 600 python scripts/exp/r05_gemm_timing.py > $O/gemm_timing.txt 2>&1; echo "timing exit $?"
cat $O/gemm_timing.txt | tail -60
timeout 900 python scripts/gemm_p_bench.py 21,51,55,57,58,60,65,66,67 > $O/gemm_sweep.txt 2>&1; echo "sweep exit $?"
grep -v "^check" $O/gemm_sweep.txt | tail -40
for pp in 0 5 6 7 8; do
  PSAM_GEMM_PP=$pp timeout 300 python bench.py --no-cpu-baseline --no-other-workloads --sustained-steps 100 --no-stage-times > $O/bench_pp$pp.json 2> $O/bench_pp$pp.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_pp$pp.json").read().strip().splitlines()[-1])
    print("PSAM_GEMM_PP=$pp", d["value"], "sustained", d["sustained"]["value"], "frac", d["roofline"]["frac"])
except Exception as e:
    print("PSAM_GEMM_PP=$pp failed", e)
PY
done
