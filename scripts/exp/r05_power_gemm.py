"""Package power, shader clock and GEMM rate of the qkv GEMM (4096 x 3072 x 1024, f16x3p) held for ~6 s per condition: which of {schedule, operand data}
moves the clock.  rocm-smi is sampled once a second from a thread while the GPU loops the launch.
  cfg 21 = one workgroup per tile, two resident per CU (production); cfg 95 = the persistent kernel with ONE workgroup per CU (fewer MFMAs per cycle);
  operands: random | lo planes zeroed (two of the three partial products multiply by zero: a third of the matrix pipe's switching) | all zero."""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_sam_amd import ops
from gemm_p_bench import pack_g8, run_p
M, N, K = 4096, 3072, 1024
samples, stop = [], threading.Event()


def sampler():
    while not stop.is_set():
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
        p = re.search(r"Socket Graphics Package Power \(W\): ([\d.]+)", out)
        c = re.search(r"sclk clock level: \w+: \((\d+)Mhz\)", out)
        if p and c:
            samples.append((time.time(), float(p.group(1)), int(c.group(1))))
        time.sleep(0.7)


def operands(kind):
    x = torch.randn(M, K, device="cuda") if kind != "zeros" else torch.zeros(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") / 32 if kind != "zeros" else torch.zeros(N, K, device="cuda")
    sa, sw = ops.row_scale_f16(x), ops.row_scale_f16(W)
    xp, wp = pack_g8(x, sa), pack_g8(W, sw)
    if kind == "lo planes zeroed":
        for P in (xp, wp):
            P.view(torch.int32).view(P.shape[0], -1, 8)[:, :, 4:] = 0
    if kind.startswith("lo low "):      # "lo low <n> bits zeroed": the n least significant mantissa bits of every lo value cleared (how much of the power is operand entropy)
        nb = int(kind.split()[2])
        mask = (0xFFFF << nb) & 0xFFFF
        m32 = ((mask << 16) | mask) - (1 << 32)      # (the mask's top bit is set: as a signed 32-bit value)
        for P in (xp, wp):
            lo = P.view(torch.int32).view(P.shape[0], -1, 8)[:, :, 4:]
            lo &= m32
    return xp, sa, wp, sw


def main():
    th = threading.Thread(target=sampler, daemon=True); th.start()
    y = torch.empty(M, N, device="cuda")
    print(f"qkv GEMM {M}x{N}x{K}; package power cap: " + (re.search(r"Max Graphics Package Power \(W\): ([\d.]+)", subprocess.run(["rocm-smi", "--showmaxpower"], capture_output=True, text=True).stdout) or [0, "?"])[1] + " W", flush=True)
    conds = ((21, "random"), (95, "random"), (21, "lo planes zeroed"), (21, "zeros"), (21, "random"))
    if os.environ.get("POWER_SWEEP") == "entropy":
        conds = ((21, "random"), (21, "lo low 4 bits zeroed"), (21, "lo low 7 bits zeroed"), (21, "lo low 10 bits zeroed"), (21, "lo planes zeroed"), (21, "random"))
    for cfg, kind in conds:
        xp, sa, wp, sw = operands(kind)
        torch.cuda.synchronize()
        t0 = time.time(); n = 0
        while time.time() - t0 < 6.0:
            for _ in range(200):
                run_p(cfg, xp, sa, wp, sw, y, M, N, K)
            torch.cuda.synchronize(); n += 200
        t1 = time.time()
        sel = [(p, c) for t, p, c in samples if t0 + 1.5 < t < t1 - 0.3]
        us = (t1 - t0) / n * 1e6
        pw = sum(p for p, _ in sel) / max(1, len(sel)); ck = sum(c for _, c in sel) / max(1, len(sel))
        print(f"cfg {cfg:2d}, {kind:22s}: {us:6.1f} us per launch = {2.0 * M * N * K / us / 1e6:4.0f} TFLOP/s fp32-equivalent ({3 * 2.0 * M * N * K / us / 1e6 / 2500:.3f} of the fp16 peak executed); "
              f"package {pw:6.0f} W, sclk {ck:5.0f} MHz ({len(sel)} samples)", flush=True)
        time.sleep(2.0)
    stop.set()


if __name__ == "__main__":
    main()
