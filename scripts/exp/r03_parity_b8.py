"""Where does the bench configuration (B=8, graphs, 3 slots, 2 streams) differ from the oracle?  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import pointsam_oracle as O
from point_sam_amd import get_config, ops
from point_sam_amd.model import PointCloudSAM, GraphPipeline
from point_sam_amd.weights import random_state_dict
from point_sam_amd.synthetic import synthetic_batch
cfg = get_config("large", 512, 64)
sd = random_state_dict(cfg, seed=42)
model = PointCloudSAM(cfg, sd, "cuda", precision=os.environ.get("PREC", "f16x3"))
B, N = 8, 32768
xyz, rgb, prompt, labels = synthetic_batch(B, N, seed=42)
dev = [t.cuda() for t in (xyz, rgb, prompt, labels)]
want = O.predict_masks(sd, cfg, xyz[:1], rgb[:1], prompt[:1], labels[:1], None, True, mode="reference")
m8, i8 = model.predict_masks(*dev)
m1, i1 = model.predict_masks(*[t[:1].contiguous() for t in dev])
torch.cuda.synchronize()
d = lambda a, b: float((a.float().cpu() - b.float().cpu()).abs().max())
print("eager B=8 cloud 0 vs oracle      :", d(m8[:1], want[0]), d(i8[:1], want[1]))
print("eager B=1 cloud 0 vs oracle      :", d(m1, want[0]), d(i1, want[1]))
print("eager B=8 cloud 0 vs eager B=1   :", d(m8[:1], m1))
for flag in ("fuse_mlp", "fuse_attn_pack", "fuse_patch", "fuse_hyper", "upscale_linear_first"):
    setattr(model, flag, False)
    mm, ii = model.predict_masks(*dev)
    print(f"  {flag}=False: B=8 cloud 0 vs oracle:", d(mm[:1], want[0]))
    setattr(model, flag, True)
pipe = GraphPipeline(model, *dev, None, True, slots=3, dense_streams=2)
for k in range(3):
    pipe.submit(*dev)
outs = []
for k in range(6):
    m, i = pipe.next()
    outs.append((m.clone(), i.clone()))
    if k + 3 < 6:
        pipe.submit(*dev)
torch.cuda.synchronize()
for k, (m, i) in enumerate(outs):
    print(f"graph step {k}: vs eager B=8 all clouds {d(m, m8):.3e}  cloud 0 vs oracle {d(m[:1], want[0]):.3e}")
