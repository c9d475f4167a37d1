"""PatchEncoder conv2.0-shaped GEMM ([262144, 128] x [512, 128]^T + a bias row per 64 rows -> fp32) under the LDS epilogue (0) and the register epilogue (1)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from point_sam_amd import ops
L = ops._lib.load()
torch.manual_seed(0)
M, N, K, grp = 262144, 512, 128, 64
x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / 11
rb = torch.randn(M // grp, N, device="cuda")
fw = ops.F16Weight(W)
out = torch.empty(M, N, device="cuda")
with ops.gemm_mode("f16x3"):
    xp, sx = ops.scale_pack_rows_g8(x)
    def run(ep):
        L.psam_gemm_f16x3p_force_epilogue(ep)
        ops.linear(xp, fw, None, rowbias=rb, rowgroup=grp, x_scale=sx, x_packed=True, out=out)
    res = {0: [], 1: []}
    for rnd in range(5):
        for ep in (0, 1):
            for _ in range(3): run(ep)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); s.record()
            for _ in range(10): run(ep)
            e.record(); torch.cuda.synchronize()
            res[ep].append(s.elapsed_time(e) * 100)
    for ep in (0, 1):
        print(f"epilogue {ep}: min {min(res[ep]):7.1f} us median {statistics.median(res[ep]):7.1f} us  ({(M * N * 4 + M * K * 4) / min(res[ep]) / 1e6:.2f} TB/s of output + input)")
L.psam_gemm_f16x3p_force_epilogue(-1)
