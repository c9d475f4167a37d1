#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; mkdir -p gpurun_out/pmcg; export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmcg/a -o g -- python $R/scripts/exp/pmc_gemm.py > /dev/null 2>$R/gpurun_out/pmcg/a.err
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/pmcg/b -o g -- python $R/scripts/exp/pmc_gemm.py > /dev/null 2>$R/gpurun_out/pmcg/b.err
cd $R
python - <<'PY'
import csv, glob, collections
for tag in ("a", "b"):
    fs = glob.glob(f"gpurun_out/pmcg/{tag}/**/*counter_collection.csv", recursive=True)
    if not fs: print(tag, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if "pipe" not in r["Kernel_Name"]: continue
        acc[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for g, d in acc.items():
        print(tag, "grid", g, {k: round(sum(v) / len(v)) for k, v in d.items()})
PY
tail -2 gpurun_out/pmcg/a.err gpurun_out/pmcg/b.err
