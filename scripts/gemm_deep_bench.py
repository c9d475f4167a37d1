"""f16x3 pipelined GEMM: operand register sets (prefetch distance) 2 / 3 / 4, packed operands (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_sam_amd import ops
L = ops._lib.load()
st = lambda: torch.cuda.current_stream().cuda_stream
SHAPES = [("qkv", 4096, 3072, 1024), ("proj", 4096, 1024, 1024), ("fc1", 4096, 5504, 1024), ("fc2", 4096, 1024, 2752), ("pe_conv2.3", 262144, 512, 512), ("upscale", 262144, 256, 256)]
for name, M, N, K in SHAPES:
    x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5; y = torch.empty(M, N, device="cuda")
    sa, sw = ops.row_scale_f16(x), ops.row_scale_f16(W)
    xp, wp = ops.pack_rows_f16x2(x, sa), ops.pack_rows_f16x2(W, sw)
    line = f"{name:11s} {M}x{N}x{K} |"
    ref = None
    for ap in (1, 0):
        for sets in (2, 3, 4):
            L.psam_gemm_f16x3_force_deep(sets)
            f = lambda: L.psam_gemm_f16x3_ex((xp if ap else x).data_ptr(), K, sa.data_ptr(), ap, wp.data_ptr(), K, sw.data_ptr(), 1, y.data_ptr(), N, 0, 0, 0, 0, 0, 0, M, N, K, 1.0, 0, st())
            f(); torch.cuda.synchronize()
            if ref is None: ref = y.clone()
            assert torch.equal(y, ref)
            for _ in range(3): f()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); s.record()
            for _ in range(10): f()
            e.record(); torch.cuda.synchronize(); us = s.elapsed_time(e) * 100
            line += f" A{'p' if ap else 'f'}:{sets}: {us:6.1f}us {2*M*N*K/us/1e6:4.0f}TF |"
    L.psam_gemm_f16x3_force_deep(-1)
    print(line, flush=True)
