"""Packed-weight bf16x6 GEMM vs symmetric bf16x6 vs f32 (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_sam_amd import ops
SHAPES = [("qkv", 4096, 3072, 1024), ("proj", 4096, 1024, 1024), ("fc1", 4096, 5504, 1024), ("fc2", 4096, 1024, 2752),
          ("pe_conv2.3", 262144, 512, 512), ("upscale", 262144, 256, 256)]
g = torch.Generator().manual_seed(0)
L = ops._lib.load()
for name, M, N, K in SHAPES:
    x = torch.randn(M, K, generator=g).cuda(); W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda(); b = torch.randn(N, generator=g).cuda()
    pw = ops.PackedWeight(W)
    y = torch.empty(M, N, device="cuda")
    rows = torch.randint(0, M, (128,), generator=g).cuda()
    ref = x[rows].double() @ W.double().T + b.double()
    line = f"{name:11s} {M:7d}x{N:5d}x{K:5d} |"
    def timeit(fn):
        for _ in range(2): fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / 10
    with ops.gemm_mode("bf16x6"):
        ms = timeit(lambda: ops.linear(x, W, b, out=y)); line += f" sym: {ms*1e3:7.1f}us {2*M*N*K/ms/1e9:6.1f}TF |"
        for cfg in (0, 1):
            L.psam_gemm_bf16x6_pw_force_config(cfg)
            ms = timeit(lambda: ops.linear(x, pw, b, out=y))
            err = ((y[rows].double() - ref).abs().max() / ref.abs().max()).item()
            line += f" pw{cfg}: {ms*1e3:7.1f}us {2*M*N*K/ms/1e9:6.1f}TF err {err:.1e} |"
        L.psam_gemm_bf16x6_pw_force_config(-1)
    print(line, flush=True)
