"""f16x3p GEMM (g8-packed operands, LDS-DMA ring): correctness vs fp64 and timing of every tile / ring configuration on the
encoder shapes, plus one encoder layer's four GEMMs on one and on two streams.  GPU box:  python scripts/gemm_p_bench.py [cfgs]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_sam_amd import ops
L = ops._lib.load()
st = lambda: torch.cuda.current_stream().cuda_stream
CFGS = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 4, 9, 12, 14, 21, 23, 28]
NAMES = {0: "128x128 4w S2", 1: "128x128 4w S3", 2: "128x128 4w S3 LA", 3: "128x128 4w S4 LA", 4: "256x128 8w S3", 5: "256x128 8w S3 LA",
         6: "128x256 8w S3 LA", 7: "256x128 8w S2", 8: "128x128 8w(32x64) S2", 9: "128x128 8w(32x64) S4 LA", 10: "128x64 4w S2", 11: "128x64 4w S3 LA",
         12: "256x192 8w S2", 13: "128x192 4w S2", 14: "256x256 8w(64x128) S2", 15: "256x256 8w(128x64) S2", 16: "256x128 4w(128x64) S2",
         20: "128x128 4w S2 PF1", 21: "128x128 4w S2 PF2", 22: "256x192 PF1", 23: "256x192 PF2", 24: "256x256 PF1", 25: "256x256 PF2", 26: "256x128 S3 PF1",
         27: "256x128 S2 PF2", 28: "128x128 8w S2 PF2", 29: "128x128 8w S4 LA PF1",
         # ping-pong kernel (gemm_f16x3pp.hip): one 8-wave workgroup per CU, two wave groups one barrier interval apart
         50: "pp 256x256 (128x64) S5 prio", 51: "pp 256x256 no prio", 52: "pp 256x256 static prio", 53: "pp 256x256 (64x128)", 59: "pp 256x256 S4", 60: "pp 256x256 S5 P2", 61: "pp 256x256 S4 P2",
         80: "q 128x256 4w(64x128) S3", 81: "q 128x128 S3", 82: "q 128x128 S4", 83: "q 128x256 4w(32x256)", 84: "q 128x192 4w(64x96)",
         62: "pp 256x224 (32x224)", 63: "pp 256x192 (32x192)", 64: "pp 256x256 (32x256)",
         55: "pp 256x128 S6 P2", 56: "pp 256x128 S6 P1", 57: "pp 128x128 S8 P2", 58: "pp 128x128 S4 P2",
         90: "stream-K 128x128 persistent", 94: "continuous 128x128 persistent (queue)", 95: "continuous, one per CU",
         65: "pp 256x128 S3 P1 2/CU", 66: "pp 256x128 S3 P1 2/CU prio", 67: "pp 128x256 S3 P1 2/CU"}
ODD_TN = (12, 23, 30, 62, 84)
# per-shape configuration maps for the two-stream layer loop (qkv, proj, fc1, fc2)
COMBOS = {"all c0": (0, 0, 0, 0), "all c4": (4, 4, 4, 4), "all c14": (14, 14, 14, 14), "c12 c4 c14 c4": (12, 4, 14, 4), "c12 c9 c14 c9": (12, 9, 14, 9),
          "c14 c4 c14 c4": (14, 4, 14, 4), "c12 c4 c4 c4": (12, 4, 4, 4), "c13 c0 c13 c0": (13, 0, 13, 0), "c12 c16 c14 c16": (12, 16, 14, 16),
          "c14 c14 c14 c14": (14, 14, 14, 14), "c15 c4 c15 c4": (15, 4, 15, 4), "c12 c9 c4 c9": (12, 9, 4, 9), "c12 c0 c14 c0": (12, 0, 14, 0),
          "c23 c9 c25 c9": (23, 9, 25, 9), "c23 c28 c25 c28": (23, 28, 25, 28), "c23 c29 c27 c29": (23, 29, 27, 29), "c22 c29 c24 c29": (22, 29, 24, 29),
          "c23 c9 c27 c9": (23, 9, 27, 9), "c23 c9 c4 c9": (23, 9, 4, 9), "all c21": (21, 21, 21, 21),
          "c50 c57 c50 c57": (50, 57, 50, 57), "c50 c55 c50 c55": (50, 55, 50, 55), "c55 c57 c55 c57": (55, 57, 55, 57), "c50 c58 c50 c58": (50, 58, 50, 58),
          "c50 c21 c50 c21": (50, 21, 50, 21), "c55 c57 c50 c57": (55, 57, 50, 57), "all c55": (55, 55, 55, 55), "all c50": (50, 50, 50, 50),
          "c21 c57 c21 c57": (21, 57, 21, 57), "all c60": (60, 60, 60, 60), "c60 c57 c60 c57": (60, 57, 60, 57), "c60 c57 c55 c57": (60, 57, 55, 57), "c60 c21 c60 c21": (60, 21, 60, 21), "c21 c57 c55 c57": (21, 57, 55, 57),
          "c62 c21 c63 c21": (62, 21, 63, 21), "c62 c57 c63 c57": (62, 57, 63, 57), "c62 c21 c21 c21": (62, 21, 21, 21), "c21 c21 c63 c21": (21, 21, 63, 21),
          "c64 c21 c64 c21": (64, 21, 64, 21), "all c80": (80, 80, 80, 80), "c80 c21 c80 c21": (80, 21, 80, 21), "c84 c21 c80 c21": (84, 21, 80, 21),
          "all c94": (94, 94, 94, 94), "c94 c21 c94 c21": (94, 21, 94, 21), "c94 c21 c94 c94": (94, 21, 94, 94), "c94 c94 c94 c21": (94, 94, 94, 21),
          "all c90": (90, 90, 90, 90), "c90 c21 c90 c21": (90, 21, 90, 21), "c90 c21 c90 c90": (90, 21, 90, 90),
          "all c65": (65, 65, 65, 65), "c65 c21 c65 c21": (65, 21, 65, 21), "c65 c58 c65 c58": (65, 58, 65, 58), "c66 c21 c66 c21": (66, 21, 66, 21),
          "c67 c21 c67 c21": (67, 21, 67, 21), "all c58": (58, 58, 58, 58), "c21 c58 c21 c58": (21, 58, 21, 58), "c55 c21 c55 c21": (55, 21, 55, 21),
          "c51 c21 c51 c21": (51, 21, 51, 21),
          "all c82": (82, 82, 82, 82), "all c83": (83, 83, 83, 83), "c80 c82 c80 c82": (80, 82, 80, 82)}


def half_chip_streams():
    """Two streams confined to complementary halves of every XCD (CU-mask bit i = XCC i % 8, SE (i / 8) % 4, CU i / 32)."""
    import ctypes
    lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "exp", "libcumask.so")
    if not os.path.exists(lib):
        return []
    C = ctypes.CDLL(lib)
    out = []
    for m in ((1 << 128) - 1, ((1 << 128) - 1) << 128):
        words = (ctypes.c_uint32 * 8)(*[(m >> (32 * w)) & 0xffffffff for w in range(8)])
        sp = ctypes.c_void_p()
        assert C.cumask_stream_create(ctypes.byref(sp), words, 8) == 0
        out.append(torch.cuda.ExternalStream(sp.value))
    return out


HALF = []
BEST = {}


PAD = int(os.environ.get("PAD", "0"))      # extra 32-bit containers per packed row (row pitch no longer a multiple of 4 KiB: L2 channel test)


def pack_g8(x, s):
    M, K = x.shape
    Kp = (K + 31) // 32 * 32
    out = torch.empty(M, Kp + PAD, device="cuda")[:, :Kp]
    ops.check(L.psam_pack_rows_f16x2_g8(x.data_ptr(), x.stride(0), s.data_ptr(), M, K, out.data_ptr(), out.stride(0), st()), "pack g8")
    return out


def run_p(cfg, xp, sa, wp, sw, y, M, N, K, bias=None, res=None, act=0, rowbias=None, rowgroup=0):
    L.psam_gemm_f16x3p_force_config(cfg)
    ops.check(L.psam_gemm_f16x3p(xp.data_ptr(), xp.stride(0), sa.data_ptr(), wp.data_ptr(), wp.stride(0), sw.data_ptr(), y.data_ptr(), y.stride(0),
                                 ops._p(bias), ops._p(res), 0 if res is None else res.stride(0), ops._p(rowbias), 0 if rowbias is None else rowbias.stride(0),
                                 rowgroup, M, N, K, 1.0, act, st()), "gemm p")


def correctness():
    g = torch.Generator(device="cuda").manual_seed(1)
    bad = 0
    for (M, N, K, act) in [(300, 200, 160, 1), (129, 257, 128, 0), (512, 384, 1024, 0), (256, 256, 128, 3), (1000, 640, 96 * 3, 3), (4096, 1024, 2752, 0)]:
        x = torch.randn(M, K, device="cuda", generator=g) * torch.exp(2 * torch.randn(M, 1, device="cuda", generator=g))
        W = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
        bias = torch.randn(N, device="cuda", generator=g)
        res = None if act == 3 else torch.randn(M, N, device="cuda", generator=g)
        sa, sw = ops.row_scale_f16(x), ops.row_scale_f16(W)
        xp, wp = pack_g8(x, sa), pack_g8(W, sw)
        Kp = xp.shape[1]
        ref = x.double() @ W.double().T + bias.double()
        if act == 3:   # weight rows alternate 32-row blocks of gate / value
            r = ref.view(M, N // 64, 2, 32)
            ref = (torch.nn.functional.silu(r[:, :, 0]) * r[:, :, 1]).reshape(M, N // 2)
        else:
            if act == 1:
                ref = torch.nn.functional.gelu(ref)
            ref = ref + res.double()
        for cfg in CFGS:
            if act == 3 and cfg in ODD_TN:  # odd TN: no SwiGLU pairing
                continue
            y = torch.full((M, N // 2 if act == 3 else N), float("nan"), device="cuda")
            run_p(cfg, xp, sa, wp, sw, y, M, N, Kp, bias=bias, res=res, act=act)
            err = ((y.double() - ref).abs().max() / ref.abs().max()).item()
            ok = err < 3e-6
            bad += not ok
            print(f"check {M}x{N}x{K} act{act} cfg{cfg:2d}: rel err {err:.2e} {'ok' if ok else 'FAIL'}", flush=True)
    return bad


def timeit(fns, rounds=5, iters=20):
    """fns: dict name -> callable; interleaved rounds; returns name -> (min us, median us) per call"""
    for f in fns.values():
        for _ in range(3):
            f()
    res = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); s.record()
            for _ in range(iters):
                f()
            e.record(); torch.cuda.synchronize()
            res[k].append(s.elapsed_time(e) * 1000 / iters)
    return {k: (min(v), statistics.median(v)) for k, v in res.items()}


SHAPES = [("qkv", 4096, 3072, 1024, 0), ("proj", 4096, 1024, 1024, 0), ("fc1", 4096, 5504, 1024, 3), ("fc2", 4096, 1024, 2752, 0),
          ("pe_conv2.3", 262144, 512, 512, 0), ("up.3", 262144, 256, 256, 1)]


def main():
    global HALF
    HALF = half_chip_streams() if os.environ.get("HALF_CHIP") else []
    bad = correctness()
    print("correctness failures:", bad, flush=True)
    if os.environ.get("ONLY_CHECK"):
        return
    data = {}
    for name, M, N, K, act in SHAPES:
        x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5
        y = torch.empty(M, N // 2 if act == 3 else N, device="cuda")
        bias = torch.randn(N, device="cuda")
        res = torch.randn(M, N, device="cuda") if name in ("proj", "fc2") else None
        sa, sw = ops.row_scale_f16(x), ops.row_scale_f16(W)
        xp, wp = pack_g8(x, sa), pack_g8(W, sw)
        data[name] = (xp, sa, wp, sw, y, M, N, K, bias, res, act)
        fns = {}
        for cfg in CFGS:
            if act == 3 and cfg in ODD_TN:  # odd TN: no SwiGLU pairing
                continue
            fns[f"c{cfg}"] = (lambda cfg=cfg: run_p(cfg, xp, sa, wp, sw, y, M, N, K, bias=bias, res=res, act=act))
        r = timeit(fns, rounds=4, iters=10 if M > 100000 else 20)
        gf = 2.0 * M * N * K
        line = f"{name:10s} {M}x{N}x{K}:"
        for k, (mn, md) in r.items():
            line += f" {k} {mn:6.1f}us {gf / mn / 1e6:4.0f}TF |"
        print(line, flush=True)
        BEST[name] = int(min(r, key=lambda k: r[k][0])[1:])
    # one encoder layer's four GEMMs back to back, on one stream and on two streams at once (the bench keeps two batches in flight)
    layer = ["qkv", "proj", "fc1", "fc2"]
    s2 = torch.cuda.Stream()
    s3 = torch.cuda.Stream()
    def layer_run(cfgmap):
        for nm in layer:
            xp, sa, wp, sw, y, M, N, K, bias, res, act = data[nm]
            c = cfgmap[nm]
            if c == "old":
                raise RuntimeError
            run_p(c, xp, sa, wp, sw, y, M, N, K, bias=bias, res=res, act=act)
    print("layer (qkv+proj+fc1+fc2 = 103.7 GF) x 24, us per layer:", flush=True)
    COMBOS["best single"] = tuple(BEST[nm] for nm in layer)
    for cname, cm4 in COMBOS.items():
        if any(c not in CFGS for c in cm4):
            continue
        cm = dict(zip(layer, cm4))
        def one():
            for _ in range(24):
                layer_run(cm)
        def two():
            s2.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s2):
                for _ in range(24):
                    layer_run(cm)
            for _ in range(24):
                layer_run(cm)
            torch.cuda.current_stream().wait_stream(s2)
        def halves():   # two streams, each confined to 16 CUs of every XCD (hipExtStreamCreateWithCUMask)
            cur = torch.cuda.current_stream()
            for hs in HALF:
                hs.wait_stream(cur)
                with torch.cuda.stream(hs):
                    for _ in range(24):
                        layer_run(cm)
            for hs in HALF:
                cur.wait_stream(hs)
        def three():
            cur = torch.cuda.current_stream()
            for sx in (s2, s3):
                sx.wait_stream(cur)
                with torch.cuda.stream(sx):
                    for _ in range(24):
                        layer_run(cm)
            for _ in range(24):
                layer_run(cm)
            cur.wait_stream(s2); cur.wait_stream(s3)
        fns = {"one": one, "two": two, "three": three}
        if HALF:
            fns["halves"] = halves
        r = timeit(fns, rounds=3, iters=2)
        line = (f"  {cname:18s}: 1 stream {r['one'][0] / 24:7.1f} us/layer ({103.7e3 / (r['one'][0] / 24):4.0f} TF) | 2 streams {r['two'][0] / 48:7.1f} us/layer "
                f"({103.7e3 / (r['two'][0] / 48):4.0f} TF)")
        line += f" | 3 streams {r['three'][0] / 72:7.1f} us/layer ({103.7e3 / (r['three'][0] / 72):4.0f} TF)"
        if HALF:
            line += f" | 2 half-chip streams {r['halves'][0] / 48:7.1f} us/layer ({103.7e3 / (r['halves'][0] / 48):4.0f} TF)"
        print(line, flush=True)
    L.psam_gemm_f16x3p_force_config(-1)


if __name__ == "__main__":
    main()
