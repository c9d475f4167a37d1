#!/bin/bash
# end-of-round evidence: full GPU suite, the three bench workloads with cpu_baseline + parity, kernel stats of the default command
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/r04_gpu_tests_final.log 2>&1; echo "pytest exit $?"; tail -3 $O/r04_gpu_tests_final.log
timeout 500 python bench.py > $O/r04_bench_final.json 2> $O/r04_bench_final.err; echo "bench exit $?"
timeout 500 python bench.py --workload cfg3 > $O/r04_bench_cfg3_final.json 2> /dev/null; echo "bench cfg3 exit $?"
timeout 500 python bench.py --workload cfg5 > $O/r04_bench_cfg5_final.json 2> /dev/null; echo "bench cfg5 exit $?"
python - <<'PY'
import json
for f in ("r04_bench_final","r04_bench_cfg3_final","r04_bench_cfg5_final"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1]); s=d.get("stage_ms") or {}
        print(f, d["value"], d["ms_per_step"], "sustained", d["sustained"]["value"], "frac", d["roofline"]["frac"], "parity", d["parity"]["max_abs_err_mask_logits"], d["parity"]["ok"], "cpu", d["cpu_baseline"]["value"], {k:s.get(k) for k in ("vit_blocks","encode_total","two_way_decoder","decode_total","ms_per_additional_click")})
    except Exception as e: print(f, "ERR", e)
PY
bash scripts/gpu_profile.sh r04_end 3 > $O/r04_prof_end.log 2>&1
f=$(find gpurun_out/prof_r04_end -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-160
find gpurun_out/prof_r04_end -name "*kernel_trace.csv" -size +1M -delete
