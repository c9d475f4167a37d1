"""Ablation timing of the f16x3 attention kernel at the bench shape (B=8, H=16, L=512, hd=64), PSAM_HIP_LIB=scripts/exp/libpointsam_attn_abl.so."""
import os, sys, statistics, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from point_sam_amd import ops
L = ops._lib.load()
try:
    L.psam_attention_set_ablation.argtypes = [ctypes.c_int32]
    HAVE = True
except AttributeError:
    HAVE = False
B, H, Lq, hd = 8, 16, 512, 64
D = H * hd
qkv = torch.randn(B * Lq, 3 * D, device="cuda")
o = torch.empty(B * Lq, D, device="cuda")
a_scale = torch.full((B * Lq,), 2.0 ** 12, device="cuda"); o_scale = torch.empty(B * Lq, device="cuda")


def run(pack):
    ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], o, B, H, Lq, Lq, hd, hd ** -0.5, pack=(a_scale, 1e3, 1.0, o_scale) if pack else None)


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    return statistics.median(ts), min(ts)


NAMES = {0: "full", 1: "no convert+store (tile 0 reused)", 3: "no convert+store, no loads", 4: "no exp / P split", 8: "no S MFMAs", 16: "no PV MFMAs", 24: "no MFMAs",
         28: "no MFMAs, no exp/split", 31: "loop skeleton (LDS frag reads + max/sum + barriers)", 63: "same, no barriers", 27: "softmax only (exp/split + frag reads)", 7: "MFMAs + frag reads only"}
with ops.gemm_mode("f16x3"):
    for a, name in NAMES.items():
        if a and not HAVE:
            continue
        if HAVE:
            L.psam_attention_set_ablation(a)
        md, mn = timed(lambda: run(False))
        print(f"abl {a:2d} {name:52s} median {md:6.1f} us  min {mn:6.1f} us", flush=True)
    if HAVE:
        L.psam_attention_set_ablation(0)
    md, mn = timed(lambda: run(True))
    print(f"packed output, full                                        median {md:6.1f} us  min {mn:6.1f} us")
    # ---- the packed-operand kernel
    sq = torch.full((B * Lq,), 2.0 ** 11, device="cuda")
    qkvp = ops.pack_rows_g8(qkv, sq)
    so = torch.empty(B * Lq, device="cuda")
    PN = {0: "full", 1: "no DMA after the prologue", 2: "no barriers", 4: "no exp / P split", 8: "no S MFMAs", 16: "no PV MFMAs", 24: "no MFMAs", 32: "no V reads", 64: "no K reads",
          96: "no fragment reads", 28: "no MFMAs, no exp/split", 124: "skeleton: DMA + barriers + max/sum", 127: "skeleton, no DMA, no barriers", 120: "softmax arithmetic only (+DMA, barriers)",
          100: "MFMAs only (+DMA, barriers)", 128: "no tile loop at all (prologue + epilogue)", 384: "prologue only", 256: "no epilogue"}
    for a, name in PN.items():
        if a and not HAVE:
            continue
        if HAVE:
            L.psam_attention_set_ablation(a)
        md, mn = timed(lambda: ops.attention_packed(qkvp, sq, o, so, B, H, Lq, hd, hd ** -0.5, 8.0))
        print(f"packed operands: abl {a:3d} {name:44s} median {md:6.1f} us  min {mn:6.1f} us", flush=True)
    if HAVE:
        L.psam_attention_set_ablation(0)
