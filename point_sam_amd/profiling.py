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


def tokenizer_metrics(stage_ms: dict, B: int, N: int, G: int, K: int) -> dict:
    """What bounds the tokenizer kernels, per stage: distance evaluations per second against the VALU peak of the CUs the kernel occupies
    (FPS: one workgroup = one CU per cloud for N <= 32768, N / 4096 per cloud above; kNN and 3-NN: the whole chip), and for FPS the time
    per dependent iteration (the kernel is a chain of G iterations, each ending in a workgroup-wide arg-max: latency-bound).
    These kernels keep their working set on chip, so HBM bytes are not their measure (the round-1/2 streaming-model figures were not
    evidence: a kernel that does not move those bytes can "exceed" the HBM peak)."""
    out = {}
    peak_cu = VALU_LANE_OPS_PER_CU / DIST_OPS
    def entry(ms, evals, cus, extra=None):
        rate = evals / (ms * 1e-3)
        e = {"ms": ms, "distance_evals": evals, "distance_evals_per_s": round(rate, 1), "cus_occupied": cus,
             "valu_peak_evals_per_s": round(peak_cu * cus, 1), "frac_of_valu_peak": round(rate / (peak_cu * cus), 4)}
        if extra:
            e.update(extra)
        return e
    if stage_ms.get("fps"):
        cus = B * max(1, N // 32768 * 8 if N > 32768 else 1)
        out["fps"] = entry(stage_ms["fps"], float(B) * G * N, min(cus, CHIP_CUS),
                           {"us_per_iteration": round(stage_ms["fps"] * 1e3 / G, 3), "iterations": G,
                            "bound": "dependent iterations: distance update (VALU) + workgroup arg-max (DPP / LDS / barrier latency) per iteration; the multi-workgroup kernel (N > 32768, or "
                                     "one or two clouds) adds one store + one polled load across the fabric (~1.0 us) and prunes the scan exactly (distance_evals counts the un-pruned "
                                     "algorithm's evaluations): its iteration is a latency chain, profiles/r05_fps_pruned.txt"})
    if stage_ms.get("knn"):
        out["knn"] = entry(stage_ms["knn"], float(B) * G * N * 2, CHIP_CUS,
                           {"bound": "VALU + L2 stream: one histogram sweep + one collection sweep over the cloud per center (2 distance evaluations per point-center pair; "
                                     "the two lower radix passes run on the selected bin's points only -- round 5, was 4 evaluations)"})
    if stage_ms.get("three_nn"):
        out["three_nn"] = entry(stage_ms["three_nn"], float(B) * N * G, CHIP_CUS, {"bound": "VALU: one distance + top-3 insertion per point-center pair"})
    return out


def read_sclk_mhz():
    """Current shader clock of every GPU the kernel driver exposes (sysfs pp_dpm_sclk, the starred level), best effort: [] if unreadable."""
    import glob
    import re
    out = []
    for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
        try:
            for line in open(f):
                if "*" in line:
                    m = re.search(r"(\d+)\s*[Mm][Hh]z", line)
                    if m:
                        out.append(int(m.group(1)))
        except OSError:
            pass
    return out
