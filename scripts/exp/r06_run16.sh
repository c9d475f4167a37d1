#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_streams.py -x -q -m gpu -s 2>&1 | tail -25 > gpurun_out/r06_test_streams.txt
cat gpurun_out/r06_test_streams.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_v1.json 2> gpurun_out/r06_bench_v1.err
tail -c 1500 gpurun_out/r06_bench_v1.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_v1.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'])
print('power', d['roofline'].get('power'))
print('ceiling', d['roofline'].get('measured_mfma_ceiling_tflops'), d['roofline'].get('frac_of_measured_ceiling'), d['roofline'].get('measured_mfma_ceiling',{}).get('power'))
print('streams', d['config']['streams'])
for w,l in d['other_workloads'].items():
    print(w, {k:l.get(k) for k in ('value','ms_per_step','in_process','error')}, l.get('roofline'), l.get('parity',{}).get('ok'), l.get('sustained'))
print('tok', {k:{a:v.get(a) for a in ('ms','hbm_model_gbs','hbm_model_frac','pmc_gbs')} for k,v in d['tokenizer'].items()})
PY
