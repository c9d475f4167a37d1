"""Round 6, VERDICT r05 item 3: the packed-operand GEMM at single-cloud row counts, per shape: tile configuration x split-K factor, us per launch
(20 launches replayed as one graph, HIP events, best of 5), alone on the chip.  Shapes: the giant encoder's four GEMMs at M = 512 (cfg #5) and ViT-L's at
M = 2048 (cfg #3).  `pick` = what the library chooses by itself."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_sam_amd import ops, _lib
from gemm_p_bench import pack_g8
L = ops._lib.load()
st = lambda: torch.cuda.current_stream().cuda_stream
counters = ops.new_counters("cuda")


def mk(M, N, K):
    x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5
    sa, sw = ops.row_scale_f16(x), ops.row_scale_f16(W)
    return pack_g8(x, sa), sa, pack_g8(W, sw), sw, torch.empty(M, N, device="cuda"), torch.randn(N, device="cuda"), torch.randn(M, N, device="cuda")


def launch(cfg, ks, xp, sa, wp, sw, y, bias, res, M, N, K, ws):
    L.psam_gemm_f16x3p_force_config(cfg)
    fuse = None
    if ks > 1:
        fuse = _lib.GemmFuse()
        fuse.splitk_ws, fuse.splitk_plane, fuse.splitk, fuse.counters = ws.data_ptr(), M * N, ks, counters.data_ptr()
    rc = L.psam_gemm_f16x3p_ex(xp.data_ptr(), xp.stride(0), sa.data_ptr(), wp.data_ptr(), wp.stride(0), sw.data_ptr(), y.data_ptr(), y.stride(0), bias.data_ptr(),
                               ops._p(res), 0 if res is None else res.stride(0), None, 0, 0, M, N, K, 1.0, 0, ctypes.byref(fuse) if fuse else None, st())
    return rc


def time_us(fn, n=20):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(n):
            fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


SHAPES = [("giant qkv", 512, 4224, 1408, False), ("giant proj", 512, 1408, 1408, True), ("giant fc1", 512, 6144, 1408, False), ("giant fc2", 512, 1408, 6144, True),
          ("large qkv", 2048, 3072, 1024, False), ("large proj", 2048, 1024, 1024, True), ("large fc1", 2048, 5504, 1024, False), ("large fc2", 2048, 1024, 2752, True)]
CFGS = [int(c) for c in os.environ.get("CFGS", "9,21,28,4,0").split(",")]
for name, M, N, K, has_res in SHAPES:
    xp, sa, wp, sw, y, bias, res = mk(M, N, K)
    Kp = xp.shape[1]
    ws = torch.empty(6, M * N, device="cuda")
    res = res if has_res else None
    ks_pick = int(L.psam_gemm_f16x3p_splitk(M, N, Kp, 0))
    t_pick = time_us(lambda: launch(-1, ks_pick, xp, sa, wp, sw, y, bias, res, M, N, Kp, ws))
    flops = 2.0 * M * N * K
    print(f"== {name} {M}x{N}x{K}: pick (ks={ks_pick}) {t_pick:6.1f} us = {flops / t_pick / 1e6:6.1f} TFLOP/s", flush=True)
    for cfg in CFGS:
        row = []
        for ks in (1, 2, 3, 4, 5, 6):
            if ks > Kp // 128:
                continue
            try:
                if launch(cfg, ks, xp, sa, wp, sw, y, bias, res, M, N, Kp, ws) != 0:
                    row.append(f"ks{ks}   n/a"); continue
                t = time_us(lambda: launch(cfg, ks, xp, sa, wp, sw, y, bias, res, M, N, Kp, ws))
                row.append(f"ks{ks} {t:5.1f}")
            except Exception as e:
                row.append(f"ks{ks} err")
        print(f"   cfg {cfg:2d}: " + "  ".join(row), flush=True)
L.psam_gemm_f16x3p_force_config(-1)
