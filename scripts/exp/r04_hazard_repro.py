"""Reproducer run for one variant library of scripts/exp/r04_hazard_variants.py: the PRE-FIX tree's graph pipeline at the bench configuration
(B=8, 3 slots, 2 dense streams) against its own eager result.  usage: r04_hazard_repro.py <variant> [steps]"""
import os, sys, shutil
TREE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hazard_tree")
n = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
csrc = os.path.join(TREE, "point_sam_amd", "csrc")
shutil.copyfile(os.path.join(csrc, f"libpointsam_hip_haz{n}.so"), os.path.join(csrc, "libpointsam_hip.so"))
sys.path.insert(0, TREE)
import torch
from point_sam_amd import get_config
from point_sam_amd.model import PointCloudSAM, GraphPipeline
from point_sam_amd.weights import random_state_dict
from point_sam_amd.synthetic import synthetic_batch
import point_sam_amd
assert os.path.dirname(point_sam_amd.__file__).startswith(TREE)
cfg = get_config("large", 512, 64)
sd = random_state_dict(cfg, seed=42)
model = PointCloudSAM(cfg, sd, "cuda", precision="f16x3")
B, N = 8, 32768
xyz, rgb, prompt, labels = synthetic_batch(B, N, seed=42)
dev = [t.cuda() for t in (xyz, rgb, prompt, labels)]
m8, i8 = model.predict_masks(*dev)
torch.cuda.synchronize()
import ctypes
lib = ctypes.CDLL(os.path.join(csrc, "libpointsam_hip.so"))
def counters(tag, reset):
    if not hasattr(lib, "psam_haz_read"): return
    buf = (ctypes.c_uint * 20)(); lib.psam_haz_read(buf, reset)
    c = list(buf)
    if any(c):
        print(f"variant {n} [{tag}]: stale operands per lane quarter [0-15 16-31 32-47 48-63]: rsq {c[0:4]} mean {c[4:8]} rstd {c[8:12]} staged float4 {c[12:16]}; checks: first pass {c[16]}, later passes {c[17]} (x64 lanes)", flush=True)
counters("eager", 1)
pipe = GraphPipeline(model, *dev, None, True, slots=3, dense_streams=2)
for k in range(3):
    pipe.submit(*dev)
bad, worst, rows_bad = 0, 0.0, set()
for k in range(steps):
    m, i = pipe.next()
    d = (m.float() - m8.float()).abs()
    e = float(d.max())
    if e != 0.0:
        bad += 1; worst = max(worst, e)
        rows_bad.update(int(c) for c in torch.nonzero(d.amax(dim=tuple(range(1, d.dim()))) > 0).flatten().tolist())
    if k + 3 < steps:
        pipe.submit(*dev)
torch.cuda.synchronize()
counters("graphs", 0)
print(f"variant {n}: {bad} of {steps} graph steps differ from eager; max |d logit| {worst:.3e}; clouds affected {sorted(rows_bad)}", flush=True)
