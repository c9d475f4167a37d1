"""Determinism / uninitialised-read probe (debugging aid): results of the tiny model before and after the allocator's free blocks are
filled with NaN or with large finite garbage; then the graph pipeline against the eager path under the same conditions."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from point_sam_amd.config import get_config
from point_sam_amd.weights import random_state_dict
from point_sam_amd.model import PointCloudSAM, GraphPipeline
from point_sam_amd.synthetic import synthetic_batch


def dirty(value):
    blocks = [torch.full((256 << 20,), value, device="cuda") for _ in range(24)]      # 24 GiB
    small = [torch.full((n,), value, device="cuda") for n in (1 << 10, 1 << 14, 1 << 18, 1 << 20, 1 << 22) for _ in range(64)]
    torch.cuda.synchronize()
    del blocks, small


cfg = get_config("tiny", 64, 16)
model = PointCloudSAM(cfg, random_state_dict(cfg, 4), device="cuda", precision="f16x3")
batches = [tuple(t.cuda() for t in synthetic_batch(2, 3000, seed=40 + i)) for i in range(7)]
clean = [model.predict_masks(*b) for b in batches]
st0 = [model.encode(b[0], b[1]).pc_embeddings.clone() for b in batches]
for value in (float("nan"), 3.0e4, -7.0e8):
    dirty(value)
    got = [model.predict_masks(*b) for b in batches]
    st1 = [model.encode(b[0], b[1]).pc_embeddings.clone() for b in batches]
    print(f"garbage {value}: masks max diff", [float((a[0] - b[0]).abs().nan_to_num(9e9).max()) for a, b in zip(clean, got)],
          "embeddings", [float((a - b).abs().nan_to_num(9e9).max()) for a, b in zip(st0, st1)], flush=True)
for rep in range(3):
    dirty(float("nan") if rep % 2 == 0 else 1.0e5)
    pipe = GraphPipeline(model, *batches[0], None, True, slots=3, dense_streams=2)
    got = []
    for k in range(pipe.depth):
        pipe.submit(*batches[k])
    for k in range(len(batches)):
        m, i = pipe.next()
        got.append((m.clone(), i.clone()))
        if k + pipe.depth < len(batches):
            pipe.submit(*batches[k + pipe.depth])
    torch.cuda.synchronize()
    print("graph vs clean eager:", [float((a[0] - b[0]).abs().nan_to_num(9e9).max()) for a, b in zip(clean, got)], flush=True)
