"""Where the persistent continuous-stream GEMM's cycles go (csrc/experiments/gemm_f16x3c.hip, TIMING instance of the measurement build
PSAM_HIP_LIB=scripts/exp/libpointsam_abl.so): per wave the cycles in the K loops (incl. drawing tiles) and in the epilogues, per encoder shape, next to
the plain kernels' times (HIP events): cfg 21 = one workgroup per tile, 94 = continuous (two resident workgroups per CU), 95 = continuous, one per CU."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_sam_amd import ops
from gemm_p_bench import pack_g8, run_p, timeit
L = ops._lib.load()
set_buf = L.psam_gemm_f16x3p_set_timing_buffer
set_buf.restype, set_buf.argtypes = None, [ctypes.c_void_p]
SHAPES = [("qkv", 4096, 3072, 1024, 0), ("fc1", 4096, 5504, 1024, 3), ("proj", 4096, 1024, 1024, 0), ("fc2", 4096, 1024, 2752, 0), ("qkv 2 batches", 8192, 3072, 1024, 0)]


def mk(M, N, K, act, res):
    x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5
    y = torch.empty(M, N // 2 if act == 3 else N, device="cuda"); bias = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda") if res else None
    sa, sw = ops.row_scale_f16(x), ops.row_scale_f16(W)
    return pack_g8(x, sa), sa, pack_g8(W, sw), sw, y, bias, r


def report(tag, d, nslabs):
    d = d.cpu().to(torch.int64)
    t0 = d[:, 10] + (d[:, 11] << 32); t1 = d[:, 12] + (d[:, 13] << 32)
    d, t0, t1 = d[t1 > 0], t0[t1 > 0], t1[t1 > 0]
    if d.numel() == 0:
        print(f"-- {tag}: no stamped waves"); return
    base = int(t0.min())
    start, end = (t0 - base).double(), (t1 - base).double()
    tiles = d[:, 2].double()
    busy = tiles > 0
    print(f"-- {tag}: {d.shape[0]} waves, {int(busy.sum())} with work; kernel span {float(end.max()):.0f} clk; wave lifetime mean {float((end - start)[busy].mean()):.0f} "
          f"min {float((end - start)[busy].min()):.0f} max {float((end - start)[busy].max()):.0f}; tiles per wave mean {float(tiles[busy].mean()):.2f} max {float(tiles.max()):.0f}; "
          f"first slab after {float(d[busy, 3].double().mean()):.0f} clk", flush=True)
    print(f"   per tile: K loop {float(d[busy, 0].double().sum() / tiles[busy].sum()):.0f} clk = {float(d[busy, 0].double().sum() / tiles[busy].sum()) / nslabs:.0f} per slab (MFMA issue floor 768), "
          f"epilogue {float(d[busy, 1].double().sum() / tiles[busy].sum()):.0f} clk", flush=True)


def main():
    for name, M, N, K, act in SHAPES:
        xp, sa, wp, sw, y, bias, r = mk(M, N, K, act, name in ("proj", "fc2"))
        f = lambda cfg: run_p(cfg, xp, sa, wp, sw, y, M, N, K, bias=bias, res=r, act=act)
        set_buf(None)
        tm = timeit({"c21": lambda: f(21), "c94 continuous": lambda: f(94), "c95 continuous, 1 per CU": lambda: f(95)}, rounds=3, iters=10)
        print(f"== {name} {M}x{N}x{K}: " + " | ".join(f"{k} {v[0]:.1f} us" for k, v in tm.items()), flush=True)
        for cfg, tag in ((94, "continuous"), (95, "continuous, one per CU")):
            buf = torch.zeros(512 * 4 * 16, dtype=torch.int32, device="cuda")
            set_buf(buf.data_ptr())
            for _ in range(2):
                f(cfg)
            torch.cuda.synchronize(); buf.zero_(); f(cfg); torch.cuda.synchronize()
            set_buf(None)
            report(f"{name} cfg {cfg} ({tag})", buf.view(-1, 16).clone(), K // 32)
    L.psam_gemm_f16x3p_force_config(-1)


if __name__ == "__main__":
    main()
