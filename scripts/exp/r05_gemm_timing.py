"""Where a wave's cycles go in the production f16x3p GEMM (128x128 tile, four waves, mid-slab release, register epilogue).
Needs the measurement build: PSAM_HIP_LIB=scripts/exp/libpointsam_abl.so (scripts/exp/gemm_abl.sh).  cfg 3000 + 64 = the timing instance: six
s_memtime stamps per 32-k slab per wave, accumulated into
    vm  wait for the wave's own DMA pieces of the slab        b1  slab barrier            f0  first-step fragment reads until they are in registers
    m0  step-0 MFMAs + step-1 fragment reads issued           b2  mid-slab barrier        m1  rest of step 0, step 1, DMA issues
plus prologue (kernel entry -> first slab), epilogue, absolute start / end stamps and HW_ID per wave.
Prints the mean cycles per slab of each bucket (all waves; waves of workgroups that started in the first 2 us = first round; the rest),
the perturbation of the instrumentation (timing instance vs plain, HIP events), and the same with a second stream running plain GEMMs beside it."""
import ctypes, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_sam_amd import ops
from gemm_p_bench import pack_g8, run_p, timeit
L = ops._lib.load()
set_buf = L.psam_gemm_f16x3p_set_timing_buffer
set_buf.restype, set_buf.argtypes = None, [ctypes.c_void_p]
NAMES = ["vm", "b1", "f0", "m0", "b2", "m1"]
SHAPES = [("qkv", 4096, 3072, 1024, 0), ("fc1", 4096, 5504, 1024, 3), ("proj", 4096, 1024, 1024, 0), ("fc2", 4096, 1024, 2752, 0)]


def mk(M, N, K, act):
    x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5
    y = torch.empty(M, N // 2 if act == 3 else N, device="cuda"); bias = torch.randn(N, device="cuda")
    sa, sw = ops.row_scale_f16(x), ops.row_scale_f16(W)
    return pack_g8(x, sa), sa, pack_g8(W, sw), sw, y, bias


def report(tag, d, nslabs):
    d = d.cpu().to(torch.int64)
    W = d.shape[0]
    t0 = d[:, 8] + (d[:, 9] << 32); t1 = d[:, 10] + (d[:, 11] << 32)
    base = int(t0.min())
    start = (t0 - base).double(); end = (t1 - base).double()
    loop = d[:, :6].sum(1).double()
    first = start < 4000      # cycles after the first workgroup's start: the first round of workgroups
    cu = (d[:, 13] & 0xf) * 4096 + ((d[:, 12] >> 8) & 0xff)       # (XCC, SE / SH / CU bits of HW_ID)
    print(f"-- {tag}: {W} waves, kernel span {float(end.max()):.0f} clk ({float(end.max()) / 1e3:.1f} k), distinct CU ids {int(torch.unique(cu).numel())}, "
          f"first-round waves {int(first.sum())}", flush=True)
    for sel, nm in ((torch.ones_like(first), "all"), (first, "first round"), (~first, "later rounds")):
        if int(sel.sum()) == 0:
            continue
        row = " ".join(f"{NAMES[i]} {float(d[sel, i].double().mean()) / nslabs:6.0f}" for i in range(6))
        print(f"   {nm:12s} per slab: {row} | sum {float(loop[sel].mean()) / nslabs:6.0f} (MFMA issue floor 768) | prologue {float(d[sel, 6].double().mean()):6.0f} "
              f"epilogue {float(d[sel, 7].double().mean()):6.0f} | wave lifetime {float((end - start)[sel].mean()):7.0f} start {float(start[sel].mean()):7.0f}", flush=True)
    # how many workgroups share a CU over time: overlap of lifetimes on the same CU id
    wg_start, wg_end, wg_cu = start.view(-1, 4).min(1).values, end.view(-1, 4).max(1).values, cu.view(-1, 4)[:, 0]
    alone = 0.0
    for c in torch.unique(wg_cu).tolist():
        m = wg_cu == c
        ev = sorted([(float(s), 1) for s in wg_start[m]] + [(float(e), -1) for e in wg_end[m]])
        n, last, t_one = 0, 0.0, 0.0
        for t, k in ev:
            if n == 1:
                t_one += t - last
            n += k; last = t
        alone += t_one
    print(f"   CU time with exactly ONE resident workgroup of this kernel: {alone / max(1, int(torch.unique(wg_cu).numel())):.0f} clk per CU", flush=True)


def main():
    s2 = torch.cuda.Stream()
    for name, M, N, K, act in SHAPES:
        xp, sa, wp, sw, y, bias = mk(M, N, K, act)
        grid = ((M + 127) // 128) * ((N + 127) // 128)
        buf = torch.zeros(grid * 4 * 16, dtype=torch.int32, device="cuda")
        set_buf(buf.data_ptr())
        f = lambda cfg: run_p(cfg, xp, sa, wp, sw, y, M, N, K, bias=bias, act=act)
        r = timeit({"plain": lambda: f(3000), "timing": lambda: f(3064)}, rounds=3, iters=10)
        print(f"== {name} {M}x{N}x{K}: plain {r['plain'][0]:.1f} us, timing instance {r['timing'][0]:.1f} us", flush=True)
        if act == 0:
            ra = timeit({k: (lambda c=c: f(c)) for k, c in (("no epilogue", 3001), ("no DMA", 3002), ("no MFMA", 3004), ("no epi no MFMA", 3005), ("MFMA only", 3019))}, rounds=3, iters=10)
            print("   ablations of the production instance: " + " | ".join(f"{k} {v[0]:.1f}" for k, v in ra.items()), flush=True)
        torch.cuda.synchronize(); buf.zero_(); f(3064); torch.cuda.synchronize()
        report(f"{name} alone", buf.view(-1, 16).clone(), K // 32)
        # beside a second stream that runs plain GEMMs of the fc1 shape back to back
        xp2, sa2, wp2, sw2, y2, b2 = mk(4096, 5504, 1024, 3)
        s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s2):
            for _ in range(6):
                run_p(3000, xp2, sa2, wp2, sw2, y2, 4096, 5504, 1024, bias=b2, act=3)
        for _ in range(2):
            f(3000)
        buf.zero_(); f(3064)
        torch.cuda.current_stream().wait_stream(s2); torch.cuda.synchronize()
        report(f"{name} beside another stream's fc1 GEMMs", buf.view(-1, 16).clone(), K // 32)
    set_buf(None)
    L.psam_gemm_f16x3p_force_config(-1)


if __name__ == "__main__":
    main()
