"""Accuracy (vs fp64) and speed of the bf16x6 GEMM against the f32-MFMA GEMM (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_sam_amd import ops
SHAPES = [("qkv", 4096, 3072, 1024), ("proj", 4096, 1024, 1024), ("fc1", 4096, 5504, 1024), ("fc2", 4096, 1024, 2752),
          ("pe_conv2.3", 262144, 512, 512), ("upscale", 262144, 256, 256), ("odd", 1000, 392, 516)]
g = torch.Generator().manual_seed(0)
for name, M, N, K in SHAPES:
    x = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda(); b = torch.randn(N, generator=g).cuda()
    rows = torch.randint(0, M, (256,), generator=g).cuda()
    ref = (x[rows].double() @ W.double().T + b.double())
    line = f"{name:11s} {M:7d}x{N:5d}x{K:5d} |"
    for mode in ("f32", "bf16x6:0", "bf16x6:1", "f16x3:0", "f16x3:1", "f16x3:2", "f16x3:3"):
        if ":" in mode:
            getattr(ops._lib.load(), f"psam_gemm_{mode.split(':')[0]}_force_config")(int(mode[-1]))
        ops.GEMM_MODE = mode.split(":")[0]
        y = ops.linear(x, W, b)
        err = ((y[rows].double() - ref).abs().max() / ref.abs().max()).item()
        for _ in range(2): ops.linear(x, W, b, out=y)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for _ in range(10): ops.linear(x, W, b, out=y)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 10
        line += f" {mode}: {ms*1e3:7.1f}us {2*M*N*K/ms/1e9:5.1f}TF err {err:.1e} |"
    ops.GEMM_MODE = "f32"
    print(line, flush=True)
