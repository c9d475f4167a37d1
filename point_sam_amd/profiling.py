"""Measurement helpers shared by bench.py and scripts/stage_times.py: per-stage GPU times (HIP events) of the path, and the
tokenizer stages' work rates against the VALU peak.  Not on the product path."""
import torch

from . import ops



class StageTimer:
    """Wraps the model's stage methods; while `on`, every call is bracketed by two HIP events on the current stream."""

    STAGES = (("fps", "fps"), ("knn", "knn"), ("three_nn", "three_nn"))

    def __init__(self, model):
        self.model, self.ev, self.on, self._saved = model, {}, False, []
        for name, label in self.STAGES:
            self._wrap(ops, name, label)
        for name, label in (("_patch_encoder", "patch_encoder"), ("_block", "vit_blocks"), ("_two_way", "two_way_decoder"), ("_encode", "encode_total"),
                            ("_decode", "decode_total")):
            self._wrap(model, name, label)

    def _wrap(self, obj, name, label):
        fn = getattr(obj, name)
        self._saved.append((obj, name, fn))

        def w(*a, **k):
            if not self.on:
                return fn(*a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = fn(*a, **k)
            e.record()
            self.ev.setdefault(label, []).append((s, e))
            return out
        setattr(obj, name, w)

    def restore(self):
        for obj, name, fn in self._saved:
            setattr(obj, name, fn)

    def totals(self, passes):
        torch.cuda.synchronize()
        return {k: round(sum(s.elapsed_time(e) for s, e in v) / passes, 3) for k, v in self.ev.items()}


@torch.no_grad()
def stage_times(model, coords, features, prompt_coords, prompt_labels, passes: int = 3, warmup: int = 1):
    """ms per pass of every stage, one batch at a time on the current stream (no pipelining: the stages do not overlap each other)."""
    t = StageTimer(model)
    try:
        def one():
            tok = model.tokenize(coords)
            st = model.encode(coords, features, tok)
            return model.decode(st, prompt_coords, prompt_labels, None, True)
        for _ in range(warmup):
            one()
        torch.cuda.synchronize()
        t.on = True
        for _ in range(passes):
            one()
        out = t.totals(passes)
    finally:
        t.on = False
        t.restore()
    out["decoder_other"] = round(out.get("decode_total", 0.0) - out.get("two_way_decoder", 0.0), 3)
    out["encoder_other"] = round(out.get("encode_total", 0.0) - out.get("patch_encoder", 0.0) - out.get("vit_blocks", 0.0), 3)
    return out


# VALU peak for the distance work of the tokenizer kernels: 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz = 78.6e12 lane-operations per second
# (MI355X_MICROARCH.md: fp32 vector peak 157.3 TFLOP/s = that many FMAs).  One squared distance with the oracle's operation order costs
# 8 individually rounded operations (3 subtractions, 3 multiplications, 2 additions; no FMA: bit-exactness), so the chip can evaluate at
# most 9.83e12 distances per second and one CU 3.84e10.
CHIP_CUS = 256
VALU_LANE_OPS_PER_CU = 4 * 32 * 2.4e9
DIST_OPS = 8


HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def tokenizer_metrics(stage_ms: dict, B: int, N: int, G: int, K: int, traffic: dict = None) -> dict:
    """What bounds the tokenizer kernels, per stage: distance evaluations per second against the VALU peak of the CUs the kernel occupies
    (FPS: one workgroup = one CU per cloud for N <= 32768, N / 4096 per cloud above; kNN and 3-NN: the whole chip), and for FPS the time
    per dependent iteration (the kernel is a chain of G iterations, each ending in a workgroup-wide arg-max: latency-bound).
    These kernels keep their working set on chip, so HBM bytes are not their measure (the round-1/2 streaming-model figures were not
    evidence: a kernel that does not move those bytes can "exceed" the HBM peak)."""
    out = {}
    peak_cu = VALU_LANE_OPS_PER_CU / DIST_OPS
    def entry(ms, evals, cus, extra=None, model_bytes=None, pmc_key=None):
        rate = evals / (ms * 1e-3)
        e = {"ms": ms, "distance_evals": evals, "distance_evals_per_s": round(rate, 1), "cus_occupied": cus,
             "valu_peak_evals_per_s": round(peak_cu * cus, 1), "frac_of_valu_peak": round(rate / (peak_cu * cus), 4)}
        if model_bytes:
            # SURVEY.md 8(d) "report both": the streaming model (what a kernel that re-reads its inputs from HBM on every pass would move) over the stage time,
            # against the HBM peak -- beside the latency / VALU view above.  These kernels keep their working set on chip, so the bytes are NOT moved
            # (`pmc_bytes_per_launch` is what rocprofv3 counted at the fabric): a model fraction near or above 1 means "served on chip", not "HBM saturated".
            gbs = model_bytes / (ms * 1e-3) / 1e9
            e.update({"hbm_model_bytes": float(model_bytes), "hbm_model_gbs": round(gbs, 1), "hbm_peak_gbs": HBM_PEAK_GBS, "hbm_model_frac": round(gbs / HBM_PEAK_GBS, 4)})
            t = (traffic or {}).get(pmc_key) if pmc_key else None
            if t:
                e["pmc_bytes_per_launch"] = t["hbm_bytes_per_launch"]
                e["pmc_gbs"] = round(t["hbm_bytes_per_launch"] / (ms * 1e-3) / 1e9, 1)
                e["pmc_note"] = "rocprofv3 FETCH_SIZE(x2)+WRITE_SIZE per launch at the cfg #2 shapes (committed PMC pass under profiles/)"
        if extra:
            e.update(extra)
        return e
    if stage_ms.get("fps"):
        cus = B * max(1, N // 32768 * 8 if N > 32768 else 1)
        out["fps"] = entry(stage_ms["fps"], float(B) * G * N, min(cus, CHIP_CUS), model_bytes=float(B) * G * N * 20, pmc_key="void fps_kernel<8>" if N == 32768 and B == 8 else None, extra=
                           {"us_per_iteration": round(stage_ms["fps"] * 1e3 / G, 3), "iterations": G,
                            "bound": "dependent iterations: distance update (VALU) + workgroup arg-max (DPP / LDS / barrier latency) per iteration; the multi-workgroup kernel (N > 32768, or "
                                     "one or two clouds) adds one store + one polled load across the fabric (~1.0 us) and prunes the scan exactly (distance_evals counts the un-pruned "
                                     "algorithm's evaluations): its iteration is a latency chain, profiles/r05/r05_fps_pruned.txt"})
    if stage_ms.get("knn"):
        out["knn"] = entry(stage_ms["knn"], float(B) * G * N * 2, CHIP_CUS, model_bytes=float(B) * (G * N * 12 + G * K * 8), pmc_key="knn_band_kernel" if N == 32768 and B == 8 else None, extra=
                           {"bound": "VALU + L2 stream: one histogram sweep + one collection sweep over the cloud per center (2 distance evaluations per point-center pair; "
                                     "the two lower radix passes run on the selected bin's points only -- round 5, was 4 evaluations)"})
    if stage_ms.get("three_nn"):
        out["three_nn"] = entry(stage_ms["three_nn"], float(B) * N * G, CHIP_CUS, model_bytes=float(B) * N * G * 12, extra={"bound": "VALU: one distance + top-3 insertion per point-center pair"})
    return out


def _card_dir(device_index: int = 0):
    """sysfs directory of the GPU torch sees as `device_index` (matched by PCI bus id; a box may expose other cards in sysfs that this process cannot
    use), or None."""
    import glob
    import os
    try:
        p = torch.cuda.get_device_properties(device_index)
        want = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}"
    except Exception:
        return None
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        if want in os.path.realpath(d).lower():
            return d
    return None


def read_sclk_mhz(device_index: int = 0):
    """Current shader clock (sysfs pp_dpm_sclk, the starred level) of the GPU in use -- of every card sysfs exposes if that one cannot be identified --,
    best effort: [] if unreadable."""
    import glob
    import re
    out = []
    d = _card_dir(device_index) if torch.cuda.is_available() else None
    for f in ([d + "/pp_dpm_sclk"] if d else sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))):
        try:
            for line in open(f):
                if "*" in line:
                    m = re.search(r"(\d+)\s*[Mm][Hh]z", line)
                    if m:
                        out.append(int(m.group(1)))
        except OSError:
            pass
    return out


class PowerSampler:
    """Samples the package power and shader clock of the GPU in use from sysfs hwmon (power1_input in uW, freq1_input in Hz; power1_cap = the cap) every
    `period` seconds on a thread, between start() and stop().  Falls back to `rocm-smi` when the card's sysfs entry cannot be found."""

    def __init__(self, device_index: int = 0, period: float = 0.1):
        import glob
        self.period, self.samples, self._stop, self._th = period, [], None, None
        d = _card_dir(device_index) if torch.cuda.is_available() else None
        hw = sorted(glob.glob(d + "/hwmon/hwmon*")) if d else []
        self.hw = hw[0] if hw else None
        self.source = "sysfs hwmon power1_input / freq1_input" if self.hw else "rocm-smi --showpower --showclocks"

    def _read(self):
        import re
        import subprocess
        if self.hw:
            try:
                return float(open(self.hw + "/power1_input").read()) * 1e-6, float(open(self.hw + "/freq1_input").read()) * 1e-6
            except (OSError, ValueError):
                return None
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
        except Exception:
            return None
        p = re.search(r"Package Power \(W\): ([\d.]+)", out)
        c = re.search(r"sclk clock level: \w+: \((\d+)Mhz\)", out)
        return (float(p.group(1)), float(c.group(1))) if p and c else None

    def cap_w(self):
        import re
        import subprocess
        if self.hw:
            try:
                return float(open(self.hw + "/power1_cap").read()) * 1e-6
            except (OSError, ValueError):
                pass
        try:
            m = re.search(r"Max Graphics Package Power \(W\): ([\d.]+)", subprocess.run(["rocm-smi", "--showmaxpower"], capture_output=True, text=True, timeout=5).stdout)
            return float(m.group(1)) if m else None
        except Exception:
            return None

    def start(self):
        import threading
        import time
        self.samples, self._stop = [], threading.Event()

        def loop():
            while not self._stop.is_set():
                r = self._read()
                if r:
                    self.samples.append((time.perf_counter(),) + r)
                self._stop.wait(self.period if self.hw else 0.5)
        self._th = threading.Thread(target=loop, daemon=True)
        self._th.start()
        return self

    def stop(self, skip_s: float = 0.5):
        """{cap_w, mean_w, max_w, sclk_mhz, samples}: samples of the first `skip_s` seconds are dropped (the power manager's ramp)."""
        self._stop.set()
        self._th.join()
        t0 = self.samples[0][0] if self.samples else 0.0
        sel = [s for s in self.samples if s[0] - t0 >= skip_s] or self.samples
        if not sel:
            return None
        return {"cap_w": self.cap_w(), "mean_w": round(sum(s[1] for s in sel) / len(sel), 1), "max_w": round(max(s[1] for s in sel), 1),
                "sclk_mhz": round(sum(s[2] for s in sel) / len(sel)), "samples": len(sel), "source": self.source}


def mfma_ceiling(seconds: float = 2.0, device_index: int = 0):
    """The matrix pipe's sustained fp16 rate on THIS box, now: csrc/probe/mfma_probe.hip (register-resident operands of the f16x3 GEMM's value
    distribution -- alternating hi / lo fragments of row-scaled fp32 data --, no memory traffic) held for `seconds` with power and clock sampled.
    Returns {tflops, power: {...}} or None when the probe library is missing."""
    import ctypes
    import os
    import time
    lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "probe", "libpsam_probe.so")
    if not os.path.exists(lib):
        return None
    L = ctypes.CDLL(lib)
    L.psam_probe_mfma_f16.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.psam_probe_mfma_f16.restype = ctypes.c_int
    dev = torch.device("cuda", device_index)
    n = 12 * 64 * 4
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(n * 2, device=dev, generator=g) * 8192
    hi = x.half()
    lo = (x - hi.float()).half()
    frags = torch.where((torch.arange(n * 2, device=dev) // (64 * 8)) % 2 == 0, hi, lo).contiguous()
    grid, iters = 1024, 4000      # 4 workgroups of 4 waves per CU
    out = torch.empty(grid * 256, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    flops = grid * 4 * iters * 24 * 2.0 * 32 * 32 * 16
    for _ in range(3):
        L.psam_probe_mfma_f16(frags.data_ptr(), out.data_ptr(), grid, iters, st)
    torch.cuda.synchronize(dev)
    ps = PowerSampler(device_index).start()
    t0, launches = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            if L.psam_probe_mfma_f16(frags.data_ptr(), out.data_ptr(), grid, iters, st) != 0:
                ps.stop()
                return None
        torch.cuda.synchronize(dev)
        launches += 10
    dt = time.perf_counter() - t0
    return {"tflops": round(launches * flops / dt / 1e12, 1), "seconds": round(dt, 2), "power": ps.stop(),
            "what": "back-to-back v_mfma_f32_32x32x16_f16 on register-resident hi/lo fp16 operands (random values), 16 waves per CU, no memory traffic"}
