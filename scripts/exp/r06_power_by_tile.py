"""Round 6, VERDICT r05 item 6: WATTS, not microseconds, of the qkv GEMM (4096 x 3072 x 1024, f16x3p, random operands) per tile configuration, each held
~5 s with package power and shader clock sampled from sysfs (point_sam_amd.profiling.PowerSampler).  CFGS / PSAM_GEMM_PANEL from the environment."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_sam_amd import ops
from point_sam_amd.profiling import PowerSampler
from gemm_p_bench import pack_g8, run_p, NAMES
M, N, K = 4096, 3072, 1024
x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / 32
sa, sw = ops.row_scale_f16(x), ops.row_scale_f16(W)
xp, wp = pack_g8(x, sa), pack_g8(W, sw)
y = torch.empty(M, N, device="cuda")
cfgs = [int(c) for c in os.environ.get("CFGS", "21,4,14,23,55,60,21").split(",")]
print(f"qkv GEMM {M}x{N}x{K}, PSAM_GEMM_PANEL={os.environ.get('PSAM_GEMM_PANEL')}", flush=True)
for cfg in cfgs:
    try:
        run_p(cfg, xp, sa, wp, sw, y, M, N, K)
        torch.cuda.synchronize()
    except Exception as e:
        print(f"cfg {cfg}: not available ({e})"); continue
    ps = PowerSampler(0).start()
    t0 = time.time(); n = 0
    while time.time() - t0 < 5.0:
        for _ in range(200):
            run_p(cfg, xp, sa, wp, sw, y, M, N, K)
        torch.cuda.synchronize(); n += 200
    t1 = time.time()
    pw = ps.stop(skip_s=1.5)
    us = (t1 - t0) / n * 1e6
    tf = 2.0 * M * N * K / us / 1e6
    print(f"cfg {cfg:2d} {NAMES.get(cfg, ''):28s}: {us:6.1f} us = {tf:4.0f} TFLOP/s fp32-eq ({3 * tf:5.0f} executed); package {pw['mean_w']:6.0f} W (max {pw['max_w']:.0f}), sclk {pw['sclk_mhz']} MHz, "
          f"{3 * tf / max(pw['mean_w'] - 255.0, 1.0):.3f} executed TFLOP/s per W above idle ({pw['samples']} samples)", flush=True)
    time.sleep(2.0)
