"""Per-stage GPU times (HIP events) of the path for BASELINE configs 2, 3 and 5; writes gpurun_out/stage_times.json.
Stages are timed by wrapping the model's own methods; each configuration runs 2 warm-ups + 5 timed passes, no pipelining."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_sam_amd import ops
from point_sam_amd.config import get_config
from point_sam_amd.model import PointCloudSAM
from point_sam_amd.weights import random_state_dict
from point_sam_amd.synthetic import synthetic_batch


class Timer:
    def __init__(self): self.ev, self.on = {}, False
    def wrap(self, obj, name, label):
        fn = getattr(obj, name)
        def w(*a, **k):
            if not self.on: return fn(*a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); out = fn(*a, **k); e.record()
            self.ev.setdefault(label, []).append((s, e)); return out
        setattr(obj, name, w)
    def totals(self, passes):
        torch.cuda.synchronize()
        return {k: round(sum(s.elapsed_time(e) for s, e in v) / passes, 3) for k, v in self.ev.items()}


def run(tag, cfg_name, G, K, B, N, clicks, precision="f16x3"):
    cfg = get_config(cfg_name, G, K)
    model = PointCloudSAM(cfg, random_state_dict(cfg, 42), "cuda", precision=precision)
    T = Timer()
    for name, label in (("fps", "fps"), ("knn", "knn"), ("three_nn", "three_nn")):
        T.wrap(ops, name, label)
    T.wrap(model, "_patch_encoder", "patch_encoder(+mask encoder)")
    T.wrap(model, "_block", "vit_blocks")
    T.wrap(model, "_two_way", "two_way_decoder")
    T.wrap(model, "_encode", "encode_total")
    T.wrap(model, "_decode", "decode_total")
    xyz, rgb, prompt, labels = synthetic_batch(B, N, seed=42)
    xyz, rgb, prompt, labels = xyz.cuda(), rgb.cuda(), prompt.cuda(), labels.cuda()
    g = torch.Generator().manual_seed(1)
    extra = xyz[:, torch.randint(0, N, (max(clicks - 1, 0),), generator=g)]
    def one():
        tok = model.tokenize(xyz)
        st = model.encode(xyz, rgb, tok)
        masks, iou = model.decode(st, prompt, labels, None, True)
        for c in range(1, clicks):     # encoder cached, decoder-only loop (pc_sam.py:139-194)
            best = torch.gather(masks, 1, iou.argmax(1).view(-1, 1, 1).expand(-1, 1, N))[:, 0]
            pc = torch.cat([prompt, extra[:, :c]], 1); pl = torch.ones(B, c + 1, dtype=labels.dtype, device="cuda")
            masks, iou = model.decode(st, pc, pl, best, False)
        return masks
    for _ in range(2): one()
    torch.cuda.synchronize(); T.on = True
    t0 = time.perf_counter(); P = 5
    for _ in range(P): one()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / P * 1e3
    res = T.totals(P); res["wall_ms_per_pass"] = round(wall, 3); res["config"] = f"{cfg_name} B={B} N={N} G={G} K={K} clicks={clicks} {precision}"
    if clicks > 1:
        res["ms_per_additional_click"] = round((res["decode_total"] - res["decode_total"] / clicks) / (clicks - 1), 3)   # rough: decode total split evenly
    print(tag, json.dumps(res), flush=True)
    return res


ALL = {"cfg2": ("large", 512, 64, 8, 32768, 1), "cfg3": ("large", 2048, 256, 1, 131072, 1), "cfg5": ("giant", 512, 64, 1, 32768, 5)}
out = {k: run(k, *v) for k, v in ALL.items() if k in os.environ.get("STAGE_CFGS", "cfg2,cfg3,cfg5").split(",")}
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/stage_times.json", "w"), indent=1)
