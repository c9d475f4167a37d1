"""split-K: in-kernel fix-up against the reduction pass, on the giant encoder's shapes.  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from point_sam_amd import ops
L = ops._lib.load()
for (M, N, K, act, inplace) in [(512, 1408, 6144, 0, True), (512, 1408, 1408, 0, True), (512, 4224, 1408, 0, False), (512, 1408, 1408, 1, False), (2048, 1024, 2784, 0, True), (300, 260, 1024, 2, False)]:
    g = torch.Generator().manual_seed(M + K)
    x = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b, res = torch.randn(N, generator=g).cuda(), torch.randn(M, N, generator=g).cuda()
    fw = ops.F16Weight(W)
    z = x.double() @ W.double().t() + b.double()
    want = (torch.nn.functional.gelu(z) if act == 1 else z.clamp_min(0) if act == 2 else z) + res.double()
    outs = {}
    with ops.gemm_mode("f16x3"):
        for fx in (0, 1):
            L.psam_gemm_f16x3p_force_splitk_fixup(fx)
            runs = []
            for rep in range(3):
                if inplace:
                    o = res.clone(); ops.linear(x, fw, b, act=act, residual=o, out=o)
                else:
                    o = ops.linear(x, fw, b, act=act, residual=res)
                runs.append(o.clone())
            torch.cuda.synchronize()
            outs[fx] = runs
    L.psam_gemm_f16x3p_force_splitk_fixup(-1)
    e0, e1 = (outs[0][0].double() - want).abs().max().item(), (outs[1][0].double() - want).abs().max().item()
    d = (outs[0][0] - outs[1][0]).abs()
    print(f"{M}x{N}x{K} act {act} inplace {inplace} ks {ops.splitk_factor(M, N, fw.Kp, act)}: err reduce {e0:.2e} fixup {e1:.2e}; fixup repeatable {all(torch.equal(outs[1][0], r) for r in outs[1])}; "
          f"fixup == reduce bitwise {torch.equal(outs[0][0], outs[1][0])}, max diff {d.max().item():.2e}, wrong rows {(d.amax(1) > 1e-3).sum().item()} cols {(d.amax(0) > 1e-3).sum().item()}", flush=True)
