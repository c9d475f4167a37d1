#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python scripts/gemm_bench.py 10 ${1:-0,1,2} 2>&1 | tee gpurun_out/gemm_bench.txt
if [ -n "$2" ]; then
  cd /tmp
  rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/counters.txt 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc1 -o g -- python $GRAFT_REPO_ROOT/scripts/gemm_bench.py 2 0 > /dev/null 2>$GRAFT_REPO_ROOT/gpurun_out/pmc1.err
  rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc2 -o g -- python $GRAFT_REPO_ROOT/scripts/gemm_bench.py 2 0 > /dev/null 2>$GRAFT_REPO_ROOT/gpurun_out/pmc2.err
  cd $GRAFT_REPO_ROOT; ls -R gpurun_out/pmc1 gpurun_out/pmc2 | head -20; tail -3 gpurun_out/pmc1.err gpurun_out/pmc2.err
fi
