#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r06_gpu_tests_v1.txt
cat gpurun_out/r06_gpu_tests_v1.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_v2.json 2> gpurun_out/r06_bench_v2.err
tail -c 800 gpurun_out/r06_bench_v2.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_v2.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'parity', d['parity']['max_abs_err_mask_logits'], d['parity']['ok'])
print('stage', d['stage_ms'])
for w,l in d['other_workloads'].items():
    print(w, {k:l.get(k) for k in ('value','ms_per_step','error')}, (l.get('roofline') or {}).get('frac'), l.get('parity',{}).get('ok'), l.get('parity',{}).get('max_abs_err_mask_logits'))
PY
