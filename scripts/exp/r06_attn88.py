"""Round 6: the fp32-input f16x3 attention at head dim 88 (giant encoder): 128-wide LDS layout with 96 ACTIVE channels (flash_attn_f16x3_kernel<128, 96>)
-- timing at the giant shapes, unsplit and key-split; correctness is covered by tests/test_gpu_kernels.py (vs fp64 SDPA; packed output = packing of fp32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from point_sam_amd import ops


def time_us(fn, n=20):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    c = ops.new_counters("cuda")
    with torch.cuda.stream(s), ops.use_counters(c):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s), ops.use_counters(c):
        for _ in range(n):
            fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


for B, H, L, hd in ((1, 16, 512, 88), (2, 16, 512, 88), (1, 16, 512, 128), (1, 16, 512, 96)):
    D = H * hd
    qkv = torch.randn(B * L, 3 * D, device="cuda")
    o = torch.empty(B * L, D, device="cuda")
    a_scale = torch.full((B * L,), 2.0 ** 12, device="cuda"); so = torch.empty(B * L, device="cuda")
    row = []
    with ops.gemm_mode("f16x3"):
        for ks in (1, 4):
            with ops.attention_keysplit(ks):
                t = time_us(lambda: ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], o, B, H, L, L, hd, hd ** -0.5, pack=(a_scale, 1e3, 1.0, so)))
            row.append(f"key split cap {ks}: {t:5.1f} us")
    print(f"B={B} H={H} L={L} hd={hd}: " + "  ".join(row), flush=True)
