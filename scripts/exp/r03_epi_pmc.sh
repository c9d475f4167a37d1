#!/bin/bash
# GPU box: SQ counters of the ping-pong GEMM with and without its epilogue (ablation build): what the epilogue's cycles are made of.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
CFGS=${GEMM_CFGS:-200,201,232}
cd /tmp
rocprofv3 -L > $R/gpurun_out/rocprof_counters.txt 2>&1
for pass in 1 2; do
  if [ $pass = 1 ]; then PMC="SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE";
  else PMC="SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_IFETCH SQ_LDS_IDX_ACTIVE"; fi
  GEMM_CFGS=$CFGS PSAM_HIP_LIB=$R/scripts/exp/libpointsam_abl.so rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $R/gpurun_out/epi_pmc$pass -o g -- python $R/scripts/exp/gemm_clock.py > $R/gpurun_out/epi_pmc$pass.err 2>&1
done
cd $R
GEMM_CFGS=$CFGS python - <<'PY'
import csv, glob, collections, os
cfgs = os.environ["GEMM_CFGS"].split(",")
out = []
for pas in (1, 2):
    cc = glob.glob(f"gpurun_out/epi_pmc{pas}/**/*counter_collection.csv", recursive=True)
    kt = glob.glob(f"gpurun_out/epi_pmc{pas}/**/*kernel_trace.csv", recursive=True)
    if not cc or not kt:
        out.append(f"pass {pas}: no counter output"); continue
    dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(kt[0]))}
    cnt = collections.defaultdict(dict)
    for r in csv.DictReader(open(cc[0])):
        cnt[r["Dispatch_Id"]][r["Counter_Name"]] = cnt[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    rows = sorted((int(d), dur[d], c) for d, c in cnt.items() if d in dur and "gemm_f16x3p" in dur[d][1])
    # launches come in groups of 30 per (fill, cfg): random fill first; keep the random groups
    groups = collections.defaultdict(list)
    for i, (d, (ns, name), c) in enumerate(rows):
        groups[i // 30].append((ns, c))
    names = sorted({k for _, _, c in rows for k in c})
    out.append(f"pass {pas}: per-launch means over the last 20 of 30 launches; qkv 4096x3072x1024, random operands")
    out.append(f"{'cfg':>6} {'us':>8} " + " ".join(f"{n[:22]:>22}" for n in names))
    for gi in range(len(cfgs)):
        lst = groups[gi][10:]
        if not lst: continue
        n = len(lst)
        out.append(f"{cfgs[gi]:>6} {sum(x[0] for x in lst) / n / 1e3:8.1f} " + " ".join(f"{sum(x[1].get(k, 0.0) for x in lst) / n:22.0f}" for k in names))
open("gpurun_out/r03_epi_pmc.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
grep -c "" gpurun_out/rocprof_counters.txt; tail -3 gpurun_out/epi_pmc1.err
