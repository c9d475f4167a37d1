import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_sam_amd import ops
from gemm_p_bench import pack_g8, run_p
g = torch.Generator(device="cuda").manual_seed(1)
for (M, N, K, act) in [(300, 200, 160, 1), (129, 257, 128, 0), (512, 384, 1024, 0), (256, 256, 128, 3), (256, 256, 128, 0), (256, 256, 256, 3)]:
    x = torch.randn(M, K, device="cuda", generator=g) * torch.exp(2 * torch.randn(M, 1, device="cuda", generator=g))
    W = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    bias = torch.randn(N, device="cuda", generator=g)
    if (M, N, K) != (256, 256, 128) and (M, N, K) != (256, 256, 256):
        continue
    sa, sw = ops.row_scale_f16(x), ops.row_scale_f16(W)
    xp, wp = pack_g8(x, sa), pack_g8(W, sw)
    outs = {}
    for cfg in (0, 7, 14, 15, 16):
        y = torch.full((M, N // 2 if act == 3 else N), float("nan"), device="cuda")
        run_p(cfg, xp, sa, wp, sw, y, M, N, K, bias=bias, act=act)
        y2 = torch.full_like(y, float("nan"))
        run_p(cfg, xp, sa, wp, sw, y2, M, N, K, bias=bias, act=act)
        outs[cfg] = y
        print(f"{M}x{N}x{K} act{act} cfg{cfg}: deterministic {torch.equal(y, y2)} nan {int(torch.isnan(y).sum())}")
    ref = x.double() @ W.double().T + bias.double()
    if act == 3:
        r = ref.view(M, N // 64, 2, 32)
        ref = (torch.nn.functional.silu(r[:, :, 0]) * r[:, :, 1]).reshape(M, N // 2)
    for cfg in (0, 7, 14, 15, 16):
        d = (outs[cfg].double() - ref).abs()
        i = int(d.argmax())
        r_, c_ = i // d.shape[1], i % d.shape[1]
        rel = (d / ref.abs().clamp_min(1e-30))
        print(f"  cfg{cfg}: max abs err {d.max():.3e} at ({r_},{c_}) ref {ref[r_, c_]:.6e} got {outs[cfg][r_, c_]:.6e}; |ref|max {ref.abs().max():.3e}; rows with err>1e-6*max: {int((d.max(1).values > 1e-6 * ref.abs().max()).sum())}; "
              f"vs cfg0 max diff {(outs[cfg] - outs[0]).abs().max():.3e}")
