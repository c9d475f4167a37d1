"""Which part of the path makes the pipelined (multi-stream) execution differ from the eager one?  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from point_sam_amd import get_config, ops
from point_sam_amd.model import PointCloudSAM, GraphPipeline, BatchPipeline
from point_sam_amd.weights import random_state_dict
from point_sam_amd.synthetic import synthetic_batch
cfg = get_config("large", 512, 64)
sd = random_state_dict(cfg, seed=42)
model = PointCloudSAM(cfg, sd, "cuda", precision="f16x3")
B, N = 8, 32768
dev = [t.cuda() for t in synthetic_batch(B, N, seed=42)]
d = lambda a, b: float((a.float() - b.float()).abs().max())
def run_pipe(pipe, n=6):
    outs = []
    for k in range(min(pipe.depth, n)):
        pipe.submit(*dev, None, True)
    for k in range(n):
        m, i = pipe.next()
        outs.append(m.clone())
        if k + pipe.depth < n:
            pipe.submit(*dev, None, True)
    torch.cuda.synchronize()
    return outs
def report(tag):
    m8, _ = model.predict_masks(*dev)
    torch.cuda.synchronize()
    e2 = run_pipe(BatchPipeline(model, dense_streams=2))
    e1 = run_pipe(BatchPipeline(model, dense_streams=1))
    print(f"{tag:28s} eager 2 dense streams vs serial: {max(d(o, m8) for o in e2):.2e} | tokenizer stream only: {max(d(o, m8) for o in e1):.2e}", flush=True)
report("all fusions")
for flag in ("fuse_mlp", "fuse_attn_pack", "fuse_patch", "fuse_hyper", "upscale_linear_first", "fuse_upscale"):
    setattr(model, flag, False)
    report(flag + "=False")
    setattr(model, flag, True)
for c in (21, 0, 9, 4):
    ops._lib.load().psam_gemm_f16x3p_force_config(c)
    report(f"forced cfg {c}")
ops._lib.load().psam_gemm_f16x3p_force_config(-1)
