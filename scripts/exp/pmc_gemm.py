import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from point_sam_amd import ops
L = ops._lib.load()
st = lambda: torch.cuda.current_stream().cuda_stream
for name, M, N, K in [("fc2", 4096, 1024, 2752), ("qkv", 4096, 3072, 1024), ("fc1", 4096, 5504, 1024)]:
    x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5; y = torch.empty(M, N, device="cuda")
    sa, sw = ops.row_scale_f16(x), ops.row_scale_f16(W)
    xp, wp = ops.pack_rows_f16x2(x, sa), ops.pack_rows_f16x2(W, sw)
    for _ in range(4):
        L.psam_gemm_f16x3_ex(xp.data_ptr(), K, sa.data_ptr(), 1, wp.data_ptr(), K, sw.data_ptr(), 1, y.data_ptr(), N, 0, 0, 0, 0, 0, 0, M, N, K, 1.0, 0, st())
    torch.cuda.synchronize()
