"""Epilogue decomposition of the ping-pong GEMM (ablation build, PSAM_HIP_LIB=scripts/exp/libpointsam_abl.so): cfg 1000 + bits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_sam_amd import ops
from gemm_p_bench import pack_g8, run_p, timeit
NAMES = {1000: "full epilogue", 1001: "no pass loop (staging only)", 1002: "no staging writes", 1003: "neither (skeleton)", 1004: "passes compute, no finish/store", 1006: "no staging, no finish", 201: "no epilogue at all", 200: "cfg 50 full"}
for name, M, N, K in [("qkv", 4096, 3072, 1024), ("pe_conv2.3", 262144, 512, 512)]:
    x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5
    y = torch.empty(M, N, device="cuda"); bias = torch.randn(N, device="cuda")
    sa, sw = ops.row_scale_f16(x), ops.row_scale_f16(W)
    xp, wp = pack_g8(x, sa), pack_g8(W, sw)
    fns = {c: (lambda c=c: run_p(c, xp, sa, wp, sw, y, M, N, K, bias=bias)) for c in NAMES}
    r = timeit(fns, rounds=3, iters=10)
    print(f"{name} {M}x{N}x{K} pp 256x256:", flush=True)
    for c, (mn, md) in r.items():
        print(f"    {NAMES[c]:36s} {mn:7.1f} us", flush=True)
