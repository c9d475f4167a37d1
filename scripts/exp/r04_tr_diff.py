"""Which outputs differ between the LDS epilogue (0) and the register epilogue (1)?  Per fused GEMM of the EVA02 block, bench size."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from point_sam_amd import ops
L = ops._lib.load()
cu = lambda t: t.cuda().contiguous()
g = torch.Generator().manual_seed(0)
M, D, H = 4096, 1024, 2730
Hp = (H + 31) // 32 * 32
h = torch.randn(M, D, generator=g)
W1 = torch.randn(2 * Hp, D, generator=g) / 32
b1 = torch.randn(2 * Hp, generator=g) * 0.1
w2g = torch.randn(D, Hp, generator=g) / 52
ln_c, ln_d, res = cu(w2g.sum(1)), cu(torch.randn(D, generator=g) * 0.1), cu(torch.randn(M, D, generator=g))
k1, k2 = float(2.0 ** 15 * math.sqrt(D) * W1.double().norm(dim=1).max()), float(b1.abs().max())
fw1, fw2g, b1 = ops.F16Weight(cu(W1)), ops.F16Weight(cu(w2g)), cu(b1)
wq, bq = ops.F16Weight(cu(torch.randn(3 * D, D, generator=g) / 32)), cu(torch.randn(3 * D, generator=g) * 0.1)
wp = ops.F16Weight(cu(torch.randn(D, D, generator=g) / 32))
with ops.gemm_mode("f16x3"):
    hp, sh = ops.scale_pack_rows_g8(cu(h))
    def run():
        up = torch.empty(M, Hp, device="cuda"); su = torch.empty(M, device="cuda"); st = torch.empty(M, ops.stat_segs(2 * Hp), 2, device="cuda")
        ops.linear(hp, fw1, b1, act=ops.ACT_SWIGLU, x_scale=sh, x_packed=True, out=up, pack_out=(su, k1, k2), stats=(st, H))
        L.psam_gemm_f16x3p_force_epilogue(0)
        mean, rstd = ops.ln_stats_finalize(st, H, 1e-6)
        return up, su, st[:, :(H + 31) // 32].contiguous(), mean, rstd
    out = {}
    for ep in (0, 1):
        L.psam_gemm_f16x3p_force_epilogue(ep)
        up, su, st, mean, rstd = run()
        out[ep] = dict(up=up, su=su, st_mean=st[..., 0].contiguous(), st_m2=st[..., 1].contiguous())
        if ep == 0:
            up0, su0, mean0, rstd0 = up, su, mean, rstd
        L.psam_gemm_f16x3p_force_epilogue(ep)
        out[ep]["fc2"] = ops.linear(up0, fw2g, ln_d, residual=res, x_scale=su0, x_packed=True, ln_fold=(mean0, rstd0, ln_c))
        sq = torch.empty(M, device="cuda"); qo = torch.empty(M, 3 * D, device="cuda")
        ops.linear(hp, wq, bq, x_scale=sh, x_packed=True, out=qo, pack_out=(sq, 0.0, 50.0))
        out[ep]["qkv_packed"] = qo; out[ep]["qkv_scale"] = sq
        out[ep]["proj"] = ops.linear(hp, wp, bq[:D].contiguous(), residual=res, x_scale=sh, x_packed=True)
        out[ep]["plain_gelu"] = ops.linear(hp, wp, bq[:D].contiguous(), act=ops.ACT_GELU, x_scale=sh, x_packed=True)
    torch.cuda.synchronize()
    for k in out[0]:
        a, b = out[0][k], out[1][k]
        ne = (a.view(torch.int32) != b.view(torch.int32))
        n = int(ne.sum())
        msg = f"{k:12s}: {n} of {a.numel()} words differ"
        if n and a.dtype == torch.float32 and "packed" not in k and k != "up":
            d = (a - b).abs(); i = int(d.argmax()); msg += f"; max |diff| {d.max().item():.3e} at {i} ({a.flatten()[i].item():.6e} vs {b.flatten()[i].item():.6e}); rows differing {int(ne.view(a.shape[0], -1).any(1).sum())}"
        if n and (k == "up" or "packed" in k):
            idx = ne.nonzero()[:5].tolist(); msg += f"; first {idx}"
        print(msg)
L.psam_gemm_f16x3p_force_epilogue(-1)
