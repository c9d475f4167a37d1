"""Flash attention: f32-MFMA kernel vs fp16-split kernel at the encoder's shape (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_sam_amd import ops
for (B, H, L, hd) in [(8, 16, 512, 64), (8, 12, 128, 64), (1, 16, 2048, 64), (8, 16, 512, 128)]:
    D = H * hd
    qkv = torch.randn(B * L, 3 * D, device="cuda"); out = torch.empty(B * L, D, device="cuda")
    line = f"B={B} H={H} L={L} hd={hd} |"
    for mode in ("f32", "f16x3"):
        with ops.gemm_mode(mode):
            f = lambda: ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, B, H, L, L, hd, hd ** -0.5)
            for _ in range(3): f()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); s.record()
            for _ in range(20): f()
            e.record(); torch.cuda.synchronize(); us = s.elapsed_time(e) * 50
        line += f" {mode}: {us:7.1f} us {4.0 * B * H * L * L * hd / us / 1e6:6.1f} TF |"
    print(line, flush=True)
