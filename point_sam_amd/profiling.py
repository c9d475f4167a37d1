"""Measurement helpers shared by bench.py and scripts/stage_times.py: per-stage GPU times (HIP events) of the path, and the
streaming-model byte counts of the tokenizer stages.  Not on the product path."""
import torch

from . import ops

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


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


def tokenizer_roofline(stage_ms: dict, B: int, N: int, G: int, K: int) -> dict:
    """Streaming-model bytes of SURVEY.md 8(d) per batch / measured stage time, as a fraction of the HBM peak.  (The FPS kernel keeps
    the cloud on chip -- registers + LDS -- so it does NOT move these bytes; the figure is the rate a streaming implementation would
    need to match it.  kNN and 3-NN do stream the coordinates, through L2.)"""
    model_bytes = {"fps": float(G) * N * 20, "knn": float(G) * N * 12 + float(G) * K * 8, "three_nn": float(N) * G * 12}
    out = {}
    for k, per_cloud in model_bytes.items():
        ms = stage_ms.get(k)
        if ms:
            gbs = per_cloud * B / (ms * 1e-3) / 1e9
            out[k] = {"ms": ms, "streaming_model_GBps": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}
    return out
