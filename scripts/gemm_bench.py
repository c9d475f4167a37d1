"""GEMM micro-benchmark over the shapes of the hot path (GPU box).  usage: python scripts/gemm_bench.py [iters] [cfgs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_sam_amd import ops

SHAPES = [("qkv", 4096, 3072, 1024), ("proj", 4096, 1024, 1024), ("fc1", 4096, 5504, 1024), ("fc2", 4096, 1024, 2752),
          ("pe_conv2.3", 262144, 512, 512), ("pe_conv2.0", 262144, 512, 128), ("pe_conv1.3", 262144, 128, 128), ("upscale", 262144, 256, 256)]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cfgs = [int(c) for c in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2]
L = ops._lib.load()
g = torch.Generator().manual_seed(0)
for name, M, N, K in SHAPES:
    x = torch.randn(M, K, generator=g).cuda(); W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda(); b = torch.randn(N, generator=g).cuda()
    y = torch.empty(M, N, device="cuda")
    line = f"{name:11s} {M:7d}x{N:5d}x{K:5d} {2*M*N*K/1e9:8.1f} GF |"
    for cfg in cfgs:
        L.psam_gemm_force_config(cfg)
        for _ in range(2): ops.linear(x, W, b, out=y)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for _ in range(iters): ops.linear(x, W, b, out=y)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / iters
        line += f" cfg{cfg}: {ms*1e3:8.1f} us {2*M*N*K/ms/1e9:6.1f} TF |"
    L.psam_gemm_force_config(-1)
    print(line, flush=True)
