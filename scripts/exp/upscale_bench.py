"""Decoder upscaling chain (mask_decoder.py:53-59,164-176) at the bench shape: every kernel of the unfused and of the fused variant timed
alone (HIP events, median of 20).  GPU box:  python scripts/exp/upscale_bench.py"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from point_sam_amd import ops
L = ops._lib.load()


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    return statistics.median(ts)


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    Z, N, G, E, C = 8, 32768, 512, 256, 3
    M = Z * N
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    keys = r(Z, G, E)
    coords, centers = torch.rand(Z, N, 3, device="cuda", generator=g) * 2 - 1, torch.rand(Z, G, 3, device="cuda", generator=g) * 2 - 1
    idx, wgt = ops.three_nn(coords, centers)
    W0, b0, W3, b3 = r(E, E) / 16, r(E) * 0.1, r(E, E) / 16, r(E) * 0.1
    gam, bet = 1 + 0.2 * r(E), 0.1 * r(E)
    hyper = r(Z, C, E)
    fw0, fw3 = ops.F16Weight(W0), ops.F16Weight(W3)
    up, u1, u2 = (torch.empty(M, E, device="cuda") for _ in range(3))
    s_up, s1, rs = (torch.empty(M, device="cuda") for _ in range(3))
    masks = torch.empty(Z, C, N, device="cuda")
    bound = ops.row_ln_bound(gam, bet)
    rep = 1
    out = {}
    with ops.gemm_mode("f16x3"):
        out["interp3 fp32"] = timed(lambda: ops.interp3(keys, idx, wgt, up, rep))
        out["interp3 packed"] = timed(lambda: ops.interp3(keys, idx, wgt, up, rep, scale_out=s_up))
        ops.interp3(keys, idx, wgt, up, rep)
        out["up.0 (pack pass + GEMM)"] = timed(lambda: ops.linear(up, fw0, b0, out=u1))
        out["LN+GELU packed"] = timed(lambda: ops.layernorm(u1, gam, bet, 1e-5, act=ops.ACT_GELU, out=u2, scale_out=rs, pack=True))
        out["up.3 GEMM+GELU"] = timed(lambda: ops.linear(u2, fw3, b3, act=ops.ACT_GELU, out=up, x_scale=rs, x_packed=True))
        out["hyper batched GEMM"] = timed(lambda: ops.gemm_batched(hyper, up, masks, C, N, E, E, E, N, C * E, N * E, C * N, Z))
        ops.interp3(keys, idx, wgt, up, rep, scale_out=s_up)
        for cfg in (-1, 40, 14, 4, 21):
            L.psam_gemm_f16x3p_force_config(cfg)
            out[f"plain 256x256 GEMM packed in, cfg {cfg}"] = timed(lambda: ops.linear(up, fw0, b0, out=u1, x_scale=s_up, x_packed=True))
        L.psam_gemm_f16x3p_force_config(-1)
        out["fused up.0 (row LN + GELU + pack)"] = timed(lambda: ops.linear(up, fw0, b0, act=ops.ACT_GELU, x_scale=s_up, x_packed=True, out=u1, pack_out=(s1, 0.0, bound),
                                                                            row_ln=(gam, bet, 1e-5)))
        out["fused up.3 (GELU + hyper, no store)"] = timed(lambda: ops.linear(u1, fw3, b3, act=ops.ACT_GELU, x_scale=s1, x_packed=True, hyper=(hyper, masks, N), no_store=True))
        out["row LN + GELU, fp32 out"] = timed(lambda: ops.linear(up, fw0, b0, act=ops.ACT_GELU, x_scale=s_up, x_packed=True, out=u2, row_ln=(gam, bet, 1e-5)))
        out["row LN only, fp32 out"] = timed(lambda: ops.linear(up, fw0, b0, x_scale=s_up, x_packed=True, out=u2, row_ln=(gam, bet, 1e-5)))
        # round-2 default chain: Linear on the patch rows, interpolation + LN + GELU (packed), second Linear + GELU, hyper products
        k1 = ops.linear(keys.view(Z * G, E), fw0, b0)
        out["up.0 on the G patch rows (4096 x 256)"] = timed(lambda: ops.linear(keys.view(Z * G, E), fw0, b0))
        out["interp3 + LN + GELU, packed"] = timed(lambda: ops.interp3(k1.view(Z, G, E), idx, wgt, up, rep, scale_out=s1, ln=(gam, bet, 1e-5), act=ops.ACT_GELU))
        out["up.3 GEMM + GELU (fp32 out)"] = timed(lambda: ops.linear(up, fw3, b3, act=ops.ACT_GELU, out=u2, x_scale=s1, x_packed=True))
        out["hyper batched GEMM (again)"] = timed(lambda: ops.gemm_batched(hyper, u2, masks, C, N, E, E, E, N, C * E, N * E, C * N, Z))
        out["up.3 GEMM + GELU + hyper partial planes + sum (no store)"] = timed(lambda: ops.linear(up, fw3, b3, act=ops.ACT_GELU, x_scale=s1, x_packed=True,
                                                                                                hyper=(hyper, masks, N), no_store=True))
    for k, v in out.items():
        print(f"{k:45s} {v:8.1f} us")


if __name__ == "__main__":
    main()
