"""Power/clock check: the same GEMM launches on random vs all-zero operands (identical instruction stream; only switching activity differs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_sam_amd import ops
from gemm_p_bench import pack_g8, run_p, timeit
M, N, K = 4096, 3072, 1024
y = torch.empty(M, N, device="cuda")
for cfg in (0, 23, 14):
    fns = {}
    for name, fill in (("random", None), ("zeros", 0.0), ("const", 1.0)):
        x = torch.randn(M, K, device="cuda") if fill is None else torch.full((M, K), fill, device="cuda")
        W = torch.randn(N, K, device="cuda") / 32 if fill is None else torch.full((N, K), fill, device="cuda")
        sa, sw = ops.row_scale_f16(x), ops.row_scale_f16(W)
        xp, wp = pack_g8(x, sa), pack_g8(W, sw)
        fns[name] = (lambda xp=xp, sa=sa, wp=wp, sw=sw: run_p(cfg, xp, sa, wp, sw, y, M, N, K))
    r = timeit(fns, rounds=4, iters=30)
    print(f"cfg{cfg} qkv 4096x3072x1024: " + " | ".join(f"{k} {v[0]:.1f} us" for k, v in r.items()), flush=True)
