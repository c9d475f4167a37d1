"""Ablation timing of the f16x3p GEMM (PSAM_HIP_LIB=scripts/exp/libpointsam_abl.so): which part of the tile time is what."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from point_sam_amd import ops
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gemm_p_bench import pack_g8, run_p, timeit
L = ops._lib.load()
ABL = {0: "full", 1: "no epilogue", 2: "no DMA", 4: "no MFMA", 16: "no frag reads", 3: "no epi, no DMA", 5: "no epi, no MFMA (DMA + frags)", 19: "MFMA only", 23: "barriers only"}
if int(os.environ.get("ABL_BASE", "100")) == 200:
    ABL.update({32: "no C stores", 34: "no C stores, no DMA"})
BASE = int(os.environ.get("ABL_BASE", "100"))      # 100: lock-step ring kernel; 200: ping-pong kernel (gemm_f16x3pp.hip)
WHICH = {0: "128x128 4w S2", 1: "256x128 8w S3", 2: "256x192 8w S2", 3: "256x256 8w S2"} if BASE == 100 else {0: "pp 256x256 S5 P1", 3: "pp 256x256 S5 P2", 1: "pp 256x128 S6 P2", 2: "pp 128x128 S8 P2"}
if BASE == 200:     # calibration: what plain streaming writes / copies of a qkv-sized output (50 MB) take on this box
    yy = torch.empty(4096, 3072, device="cuda"); y2 = torch.randn(4096, 3072, device="cuda")
    r = timeit({"zero_ 50 MB": lambda: yy.zero_(), "copy_ 50 MB -> 50 MB": lambda: yy.copy_(y2)}, rounds=3, iters=20)
    for k, (mn, md) in r.items():
        print(f"calibration {k}: {mn:.1f} us", flush=True)
for name, M, N, K, act in [("qkv", 4096, 3072, 1024, 0), ("fc1-shape", 4096, 5504, 1024, 0)] + ([] if BASE == 200 else [("qkv x2 batches", 8192, 3072, 1024, 0)]):
    x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5
    y = torch.empty(M, N, device="cuda"); bias = torch.randn(N, device="cuda")
    sa, sw = ops.row_scale_f16(x), ops.row_scale_f16(W)
    xp, wp = pack_g8(x, sa), pack_g8(W, sw)
    for which in WHICH:
        fns = {}
        for a in ABL:
            fns[a] = (lambda a=a: run_p(BASE + (64 if BASE == 200 else 32) * which + a, xp, sa, wp, sw, y, M, N, K, bias=bias))
        r = timeit(fns, rounds=3, iters=10)
        print(f"{name} {M}x{N}x{K} {WHICH[which]}:", flush=True)
        for a, (mn, md) in r.items():
            print(f"    {ABL[a]:34s} {mn:7.1f} us", flush=True)
