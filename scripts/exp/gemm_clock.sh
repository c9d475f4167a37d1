#!/bin/bash
# GPU box: counters of scripts/exp/gemm_clock.py -> gpurun_out/gemm_clock.txt (per shape and operand fill: duration, cycles, clock, MFMA busy)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/gemm_clock -o g -- python $R/scripts/exp/gemm_clock.py > $R/gpurun_out/gemm_clock.err 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
cc = glob.glob("gpurun_out/gemm_clock/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob("gpurun_out/gemm_clock/**/*kernel_trace.csv", recursive=True)[0]
dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"])) for r in csv.DictReader(open(kt))}
cnt = collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    cnt[r["Dispatch_Id"]][r["Counter_Name"]] = cnt[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
rows = [(int(d), dur[d], c) for d, c in cnt.items() if d in dur and ("gemm_f16x3p_kernel" in dur[d][1] or "gemm_f16x3c_kernel" in dur[d][1])]
rows.sort()
out = ["f16x3p GEMM (128x128 tile, cfg 21), 30 back-to-back launches per (shape, operand fill); rocprofv3 --pmc, one pass; mean of the last 20 launches of each group.",
       "clock = (GRBM_GUI_ACTIVE / 8 XCDs) / kernel duration; matrix-pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles).", "",
       f"{'grid':>8} {'group':>6} {'dur us':>8} {'cycles':>9} {'clock MHz':>10} {'mfma util':>10} {'issue':>7} {'stall':>7} {'parked':>7}"]
groups = collections.defaultdict(list)
for i, (d, (ns, name, grid), c) in enumerate(rows):
    groups[(grid, i // 30)].append((ns, c))
import os
ncfg = len(os.environ.get("GEMM_CFGS", "21").split(","))
for (grid, gi), lst in sorted(groups.items(), key=lambda kv: kv[0][1]):
    lst = lst[10:]
    n = len(lst)
    ns = sum(x[0] for x in lst) / n
    g = lambda k: sum(x[1].get(k, 0.0) for x in lst) / n
    cyc = g("GRBM_GUI_ACTIVE") / 8
    wc = g("SQ_WAVE_CYCLES")
    out.append(f"{grid:8d} {('random' if (gi // ncfg) % 2 == 0 else 'zeros') + ('' if ncfg == 1 else '/' + os.environ['GEMM_CFGS'].split(',')[gi % ncfg]):>6} {ns / 1e3:8.1f} {cyc:9.0f} {cyc / ns * 1e3:10.0f} {g('SQ_VALU_MFMA_BUSY_CYCLES') / (1024 * cyc):10.3f} "
               f"{g('SQ_ACTIVE_INST_ANY') / wc:7.2f} {g('SQ_WAIT_INST_ANY') / wc:7.2f} {g('SQ_WAIT_ANY') / wc:7.2f}")
open("gpurun_out/gemm_clock.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
