"""Packed-operand attention at the encoder's shape: variant 0 (256-row workgroups, 3-tile ring, one per CU) vs variant 1 (two 128-row workgroups per
CU on a 2-tile ring): bitwise equality and time, alone and beside a GEMM stream."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from point_sam_amd import ops
L_ = ops._lib.load()
torch.manual_seed(0)
for (B, H, L) in [(8, 16, 512), (1, 16, 2048), (2, 16, 512), (8, 16, 130)]:
    D, M = H * 64, B * L
    qkv = torch.randn(M, 3 * D, device="cuda")
    sq = torch.full((M,), 2.0 ** 11, device="cuda")
    qkvp = ops.pack_rows_g8(qkv, sq)
    outs = {}
    def run(v):
        L_.psam_attention_packed_force_variant(v)
        o = torch.empty(M, D, device="cuda"); so = torch.empty(M, device="cuda")
        ops.attention_packed(qkvp, sq, o, so, B, H, L, 64, 0.125, 8.0)
        return o
    a, b = run(0), run(1)
    torch.cuda.synchronize()
    same = torch.equal(a.view(torch.int32), b.view(torch.int32))
    res = {}
    for rnd in range(5):
        for v in (0, 1):
            for _ in range(3): run(v)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); s.record()
            for _ in range(20): run(v)
            e.record(); torch.cuda.synchronize()
            res.setdefault(v, []).append(s.elapsed_time(e) * 50)
    print(f"B={B} H={H} L={L}: bitwise equal {same}; variant 0 {min(res[0]):6.1f} us (median {statistics.median(res[0]):6.1f}) | variant 1 {min(res[1]):6.1f} us (median {statistics.median(res[1]):6.1f})", flush=True)
L_.psam_attention_packed_force_variant(-1)
