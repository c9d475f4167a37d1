"""Attention at the encoder shape (B=8, H=16, L=512, hd=64): per-tile-converting kernel vs the packed-operand kernel (incl. what each needs from
the qkv GEMM: fp32 output vs packed output).  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_sam_amd import ops
from gemm_p_bench import timeit
B, H, L, hd = 8, 16, 512, 64
D, M = H * hd, B * L
qkv = torch.randn(M, 3 * D, device="cuda")
a_scale = torch.full((M,), 2.0 ** 12, device="cuda")
sq = torch.full((M,), 2.0 ** 11, device="cuda")
qkvp = ops.pack_rows_g8(qkv, sq)
o = torch.empty(M, D, device="cuda"); so = torch.empty(M, device="cuda")
h = torch.randn(M, D, device="cuda"); W = ops.F16Weight(torch.randn(3 * D, D, device="cuda") / 32); bq = torch.zeros(3 * D, device="cuda")
with ops.gemm_mode("f16x3"):
    hp, sh = ops.scale_pack_rows_g8(h)
    out32 = torch.empty(M, 3 * D, device="cuda"); outp = torch.empty(M, 3 * D, device="cuda"); s2 = torch.empty(M, device="cuda")
    fns = {
        "attention f16x3 (fp32 q/k/v, packed out)": lambda: ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], o, B, H, L, L, hd, 0.125, pack=(a_scale, 100.0, 1.0, so)),
        "attention packed operands": lambda: ops.attention_packed(qkvp, sq, o, so, B, H, L, hd, 0.125, 8.0),
        "qkv GEMM, fp32 output": lambda: ops.linear(hp, W, bq, x_scale=sh, x_packed=True, out=out32),
        "qkv GEMM, packed output": lambda: ops.linear(hp, W, bq, x_scale=sh, x_packed=True, out=outp, pack_out=(s2, 0.0, 300.0)),
    }
    r = timeit(fns, rounds=5, iters=20)
    for k, (mn, md) in r.items():
        print(f"{k:44s} {mn:7.1f} us (median {md:.1f})", flush=True)
