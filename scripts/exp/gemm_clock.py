"""Clock / matrix-pipe evidence for the f16x3p GEMM: run under
   rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace
the same launches (qkv / fc1 / fc2 shapes, the shipped 128x128 tile) on random and on all-zero operands, 30 launches each back to back
(so that the chip is in its steady power state).  scripts/exp/gemm_clock.sh parses the counters: cycles / duration = sustained clock."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_sam_amd import ops
from gemm_p_bench import pack_g8, run_p
CFGS = [int(c) for c in os.environ.get("GEMM_CFGS", "21").split(",")]      # with the ablation build: 100 + 32 * which + ablation bits (gemm_abl.py)
SHAPES = (("qkv", 4096, 3072, 1024), ("fc1", 4096, 5504, 1024), ("fc2", 4096, 1024, 2752)) if len(CFGS) == 1 else (("qkv", 4096, 3072, 1024),)
for name, M, N, K in SHAPES:
    y = torch.empty(M, N, device="cuda")
    for fill in (None, 0.0):
        x = torch.randn(M, K, device="cuda") if fill is None else torch.full((M, K), fill, device="cuda")
        W = torch.randn(N, K, device="cuda") / 32 if fill is None else torch.full((N, K), fill, device="cuda")
        sa, sw = ops.row_scale_f16(x), ops.row_scale_f16(W)
        xp, wp = pack_g8(x, sa), pack_g8(W, sw)
        for cfg in CFGS:
            torch.cuda.synchronize()
            for _ in range(30):
                run_p(cfg, xp, sa, wp, sw, y, M, N, K)
            torch.cuda.synchronize()
