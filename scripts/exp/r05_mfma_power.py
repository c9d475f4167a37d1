"""Products per second of the matrix pipe under the package power cap, by operand type and operand data (scripts/exp/r05_mfma_power.hip): each condition is
held ~5 s, rocm-smi sampled every 0.7 s.  Operands: 12 fragments per lane, used in rotating pairs (the inputs toggle between consecutive MFMAs).
  f16 / bf16 32x32x16 (16384 MAC per instruction), i8 32x32x32 (32768), fp8 e4m3 32x32x16 (16384)."""
import ctypes, os, re, subprocess, sys, threading, time
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(HERE, "libr05_mfma_power.so"))
L.run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
samples, stop = [], threading.Event()


def sampler():
    while not stop.is_set():
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
        p = re.search(r"Socket Graphics Package Power \(W\): ([\d.]+)", out)
        c = re.search(r"sclk clock level: \w+: \((\d+)Mhz\)", out)
        if p and c:
            samples.append((time.time(), float(p.group(1)), int(c.group(1))))
        time.sleep(0.7)


def frags(kind):
    n = 12 * 64 * 4      # 12 fragments x 64 lanes x 4 dwords
    g = torch.Generator(device="cuda").manual_seed(1)
    if kind == "f16 random":
        return torch.randn(n * 2, device="cuda", generator=g).half().view(torch.int32)
    if kind == "f16 hi/lo mix":      # what the f16x3 GEMM multiplies: hi = fp16(x), lo = fp16(x - hi) of row-scaled values
        x = torch.randn(n * 2, device="cuda", generator=g) * 8192
        hi = x.half(); lo = (x - hi.float()).half()
        t = torch.where((torch.arange(n * 2, device="cuda") // (64 * 8)) % 2 == 0, hi, lo)      # alternate fragments: hi, lo, hi, lo ...
        return t.view(torch.int32)
    if kind == "bf16 random":
        return torch.randn(n * 2, device="cuda", generator=g).bfloat16().view(torch.int32)
    if kind == "i8 random":
        return torch.randint(-127, 128, (n * 4,), device="cuda", generator=g, dtype=torch.int32).to(torch.int8).view(torch.int32)
    if kind == "fp8 random":
        return torch.randn(n * 4, device="cuda", generator=g).to(torch.float8_e4m3fn).view(torch.int32)
    if kind == "zeros":
        return torch.zeros(n, device="cuda", dtype=torch.int32)
    raise ValueError(kind)


def main():
    th = threading.Thread(target=sampler, daemon=True); th.start()
    out = torch.empty(1024 * 256, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    grid, iters = 1024, 4000      # 4 workgroups of 4 waves per CU
    for typ, name, macs, kind in ((0, "f16 32x32x16", 16384, "f16 random"), (0, "f16 32x32x16", 16384, "f16 hi/lo mix"), (1, "bf16 32x32x16", 16384, "bf16 random"),
                                  (2, "i8 32x32x32", 32768, "i8 random"), (3, "fp8 32x32x16", 16384, "fp8 random"), (0, "f16 32x32x16", 16384, "zeros"), (2, "i8 32x32x32", 32768, "zeros")):
        f = frags(kind)
        torch.cuda.synchronize()
        t0 = time.time(); n = 0
        while time.time() - t0 < 5.0:
            for _ in range(20):
                L.run(typ, f.data_ptr(), out.data_ptr(), grid, iters, st)
            torch.cuda.synchronize(); n += 20
        t1 = time.time()
        sel = [(p, c) for t, p, c in samples if t0 + 1.5 < t < t1 - 0.3]
        rate = n * grid * 4 * iters * 24 * macs * 2 / (t1 - t0)      # ops per second (2 per MAC)
        pw = sum(p for p, _ in sel) / max(1, len(sel)); ck = sum(c for _, c in sel) / max(1, len(sel))
        print(f"{name:14s} {kind:14s}: {rate / 1e12:7.0f} T(FL)OP/s, package {pw:6.0f} W, sclk {ck:5.0f} MHz -> {rate / 1e12 / max(pw - 255.0, 1.0):6.2f} T(FL)OP/s per W above idle ({len(sel)} samples)", flush=True)
        time.sleep(2.0)
    stop.set()


if __name__ == "__main__":
    main()
