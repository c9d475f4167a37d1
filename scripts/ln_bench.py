"""LayerNorm kernel timing at the path's shapes (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_sam_amd import ops
for rows, cols, res in [(4096, 2752, False), (4096, 1024, False), (4096, 1024, True), (262144, 512, False), (262144, 128, False), (4096, 256, False)]:
    x = torch.randn(rows, cols, device="cuda"); w = torch.randn(cols, device="cuda"); b = torch.randn(cols, device="cuda")
    r = torch.randn(rows, cols, device="cuda") if res else None
    y = torch.empty_like(x); rs = torch.empty(rows, device="cuda")
    f = lambda: ops.layernorm(x, w, b, 1e-5, residual=r, out=y, scale_out=rs)
    for _ in range(3): f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(20): f()
    e.record(); torch.cuda.synchronize(); us = s.elapsed_time(e) * 50
    gb = rows * cols * 4 * (3 if res else 2) / 1e9
    print(f"LN {rows}x{cols} res={res}: {us:7.1f} us  {gb / us * 1e6 / 1e3:6.2f} TB/s", flush=True)
