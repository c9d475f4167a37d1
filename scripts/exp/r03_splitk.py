"""Split-K of the packed-operand GEMM on the B = 1 shapes (cfg5: M = 512, D = 1408, mlp 6144; cfg3: M = 2048, D = 1024, mlp 2752 (SwiGLU: fc2 only)).
Times ops.linear per forced split factor and checks every result against ks = 1."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

SHAPES = [("g.qkv", 512, 4224, 1408), ("g.proj", 512, 1408, 1408), ("g.fc1", 512, 6144, 1408), ("g.fc2", 512, 1408, 6144),
          ("L.qkv", 2048, 3072, 1024), ("L.proj", 2048, 1024, 1024), ("L.fc2", 2048, 1024, 2752 // 32 * 32 + 32)]


def child(ks, cfg):
    import torch
    from point_sam_amd import ops, _lib
    L = _lib.load()
    if cfg >= 0:
        L.psam_gemm_f16x3p_force_config(cfg)
    torch.manual_seed(0)
    out = {}
    ops.GEMM_MODE = "f16x3"
    for name, M, N, K in SHAPES:
        x = torch.randn(M, K, device="cuda")
        W = torch.randn(N, K, device="cuda") / K ** 0.5
        b = torch.randn(N, device="cuda")
        r = torch.randn(M, N, device="cuda")
        fw = ops.F16Weight(W)
        xp, xs = ops.scale_pack_rows_g8(x, K)
        y = ops.linear(xp, fw, b, act=ops.ACT_GELU, residual=r, x_scale=xs, x_packed=True)
        ref = torch.nn.functional.gelu(x.double() @ W.double().t() + b.double()) + r.double()
        err = float((y.double() - ref).abs().max())
        for _ in range(5):
            ops.linear(xp, fw, b, act=ops.ACT_GELU, residual=r, x_scale=xs, x_packed=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ops.linear(xp, fw, b, act=ops.ACT_GELU, residual=r, x_scale=xs, x_packed=True)
        e1.record(); torch.cuda.synchronize()
        out[name] = (round(e0.elapsed_time(e1) * 20, 1), ops.splitk_factor(M, N, fw.Kp, ops.ACT_GELU), f"{err:.1e}")
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(int(sys.argv[1]), int(sys.argv[2]))
    else:
        for cfg in (-1, 9, 4, 28):
            for ks in (0, 1, 2, 3, 4, 5, 6, 8):
                env = dict(os.environ)
                if ks == 1: env["PSAM_GEMM_SPLITK"] = "0"
                elif ks > 1: env["PSAM_GEMM_SPLITK"] = str(ks)
                else: env.pop("PSAM_GEMM_SPLITK", None)
                r = subprocess.run([sys.executable, __file__, str(ks), str(cfg)], env=env, capture_output=True, text=True)
                print(f"cfg {cfg:3d} ks {'auto' if ks == 0 else ks}: {r.stdout.strip() or r.stderr[-300:].replace(chr(10), ' | ')}", flush=True)
