#!/bin/bash
# The evidence of a round, one gpurun call: GPU test suite, the default bench line, rocprofv3 kernel statistics of the same command (whole run and
# steady-state window), FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, kernel-trace only -- never with sys/hip traces).  TAG = file prefix, e.g. r06.
TAG=${1:-r06}
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -q -m gpu 2>&1 | tail -15 > gpurun_out/${TAG}_gpu_suite.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
S="--steps 40 --warmup 5 --no-cpu-baseline --no-stage-times --no-gemm-profile --no-mfma-probe --sustained-steps 0 --no-other-workloads"
rm -rf /tmp/kt_$TAG; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$TAG -o kt -- python bench.py $S > gpurun_out/${TAG}_bench_under_rocprof.json 2>/dev/null
cp $(find /tmp/kt_$TAG -name '*kernel_stats.csv' | head -1) gpurun_out/${TAG}_rocprofv3_kernel_stats.csv
ms=$(python -c "import json; d=json.loads(open('gpurun_out/${TAG}_bench_under_rocprof.json').read().strip().splitlines()[-1]); print(d['ms_per_step']*30)")
python scripts/profile_steady.py $(find /tmp/kt_$TAG -name '*kernel_trace.csv' | head -1) $ms gpurun_out/${TAG}_steady_state_kernel_stats.csv --steps=30 2> gpurun_out/${TAG}_steady_state.txt
for c in FETCH_SIZE WRITE_SIZE; do
  d=gpurun_out/pmc_$(echo $c | cut -d_ -f1 | tr A-Z a-z)_$TAG; rm -rf $d
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o pmc -- python bench.py --steps 2 --warmup 1 --no-graphs --no-cpu-baseline --no-stage-times --no-gemm-profile --no-mfma-probe --sustained-steps 0 --no-other-workloads > /dev/null 2>&1
done
python scripts/pmc_to_traffic.py $TAG gpurun_out/${TAG}_traffic.json > /dev/null 2>&1
rm -rf gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG
tail -3 gpurun_out/${TAG}_gpu_suite.log; python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'of ceiling', d['roofline'].get('frac_of_measured_ceiling'), 'parity', d['parity']['max_abs_err_mask_logits'], d['parity']['ok'])
for w,l in d['other_workloads'].items(): print(w, l.get('value'), l.get('ms_per_step'), (l.get('roofline') or {}).get('frac'), l.get('parity',{}).get('ok'), l.get('error'))
PY
head -8 gpurun_out/${TAG}_steady_state_kernel_stats.csv | cut -c1-160; python -c "
import json; t=json.load(open('gpurun_out/${TAG}_traffic.json')); print({k:v['hbm_bytes_per_launch'] for k,v in t.items() if isinstance(v,dict)})"
