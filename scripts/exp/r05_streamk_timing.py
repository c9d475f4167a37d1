"""Where the persistent stream-K GEMM's cycles go (csrc/experiments/gemm_f16x3s.hip, TIMING instance of the measurement build PSAM_HIP_LIB=scripts/exp/libpointsam_abl.so):
per wave the cycles in the K loops, in issuing the next piece's first slabs, parking a part, counting in, combining, epilogues; per shape and mode
(90 = even shares, 91 = whole tiles round-robin, 92 = even shares on one workgroup per CU), next to the plain kernels' times (HIP events)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_sam_amd import ops
from gemm_p_bench import pack_g8, run_p, timeit
L = ops._lib.load()
set_buf = L.psam_gemm_f16x3p_set_timing_buffer
set_buf.restype, set_buf.argtypes = None, [ctypes.c_void_p]
SHAPES = [("qkv", 4096, 3072, 1024, 0), ("fc1", 4096, 5504, 1024, 3), ("proj", 4096, 1024, 1024, 0), ("fc2", 4096, 1024, 2752, 0)]
NAMES = ["loop", "issue", "park", "count", "combine", "epilogue", "prologue"]


def mk(M, N, K, act, res):
    x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5
    y = torch.empty(M, N // 2 if act == 3 else N, device="cuda"); bias = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda") if res else None
    sa, sw = ops.row_scale_f16(x), ops.row_scale_f16(W)
    return pack_g8(x, sa), sa, pack_g8(W, sw), sw, y, bias, r


def report(tag, d):
    d = d.cpu().to(torch.int64)
    d = d[(d[:, 7] > 0)]
    if d.numel() == 0:
        print(f"-- {tag}: no stamped waves (the launch did not take the stream-K kernel)"); return
    t0 = d[:, 10] + (d[:, 11] << 32); t1 = d[:, 12] + (d[:, 13] << 32)
    base = int(t0.min())
    start, end = (t0 - base).double(), (t1 - base).double()
    cu = (d[:, 15] & 0xf) * 4096 + ((d[:, 14] >> 8) & 0xff)
    per_cu = torch.unique(cu, return_counts=True)[1].double() / 4
    row = " ".join(f"{NAMES[i]} {float(d[:, i].double().mean()):7.0f}" for i in range(7))
    print(f"-- {tag}: {d.shape[0]} waves on {int(torch.unique(cu).numel())} CU ids ({float(per_cu.mean()):.2f} workgroups per CU, max {float(per_cu.max()):.0f}); kernel span {float(end.max()):.0f} clk; "
          f"wave lifetime mean {float((end - start).mean()):.0f} min {float((end - start).min()):.0f} max {float((end - start).max()):.0f}; start spread {float(start.max()):.0f}", flush=True)
    print(f"   mean cycles per wave: {row} | pieces {float(d[:, 7].double().mean()):.2f} slabs {float(d[:, 8].double().mean()):.1f} tiles finished {float(d[:, 9].double().mean()):.2f} "
          f"| loop cycles per slab {float(d[:, 0].double().sum() / d[:, 8].double().sum()):.0f}", flush=True)


def main():
    for name, M, N, K, act in SHAPES:
        xp, sa, wp, sw, y, bias, r = mk(M, N, K, act, name in ("proj", "fc2"))
        f = lambda cfg: run_p(cfg, xp, sa, wp, sw, y, M, N, K, bias=bias, res=r, act=act)
        set_buf(None)
        tm = timeit({"c21": lambda: f(21), "c90 shares": lambda: f(90), "c91 whole tiles": lambda: f(91), "c92 shares, 1 per CU": lambda: f(92)}, rounds=3, iters=10)
        print(f"== {name} {M}x{N}x{K}: " + " | ".join(f"{k} {v[0]:.1f} us" for k, v in tm.items()), flush=True)
        for cfg, tag in ((90, "even shares"), (91, "whole tiles"), (92, "even shares, one per CU")):
            buf = torch.zeros(512 * 4 * 16, dtype=torch.int32, device="cuda")
            set_buf(buf.data_ptr())
            for _ in range(2):
                f(cfg)
            torch.cuda.synchronize(); buf.zero_(); f(cfg); torch.cuda.synchronize()
            set_buf(None)
            report(f"{name} cfg {cfg} ({tag})", buf.view(-1, 16).clone())
    L.psam_gemm_f16x3p_force_config(-1)


if __name__ == "__main__":
    main()
