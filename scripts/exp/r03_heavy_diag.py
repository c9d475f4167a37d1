"""Which fusion loses precision under heavy-tailed weights (tests/test_gpu_e2e.py::test_heavy_tailed_weights_against_oracle)?
Embedding / mask error vs the oracle with each f16x3 fusion switched off in turn."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from dataclasses import replace
from oracle import pointsam_oracle as O
from point_sam_amd.config import ModelConfig, get_config
from point_sam_amd.weights import random_state_dict
from point_sam_amd.model import PointCloudSAM
import importlib.util
spec = importlib.util.spec_from_file_location("e2e", os.path.join(ROOT, "tests", "test_gpu_e2e.py")); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)

def err(a, b): return (a.detach().cpu().double() - b.detach().cpu().double()).abs().max().item()

for name, shift, gain in (("base", 0.0, 15.0), ("large", 0.0, 50.0), ("large", 20.0, 15.0)):
    cfg = get_config(name, 128, 32)
    sd = m._heavy_tailed(random_state_dict(cfg, seed=11), seed=12, fc1_shift=shift, gain=gain)
    xyz, rgb, prompt, labels = O.synthetic_batch(2, 4096, seed=13, num_prompts=1)
    wm, wi, mid = O.predict_masks(sd, cfg, xyz, rgb, prompt, labels, None, True, mode="exact", return_intermediates=True)
    flags = ["row_bounds", "fuse_mlp"]
    print(name, "flags:", flags)
    for off in [()] + [(f,) for f in flags] + [tuple(flags)]:
        model = PointCloudSAM(cfg, sd, "cuda", precision="f16x3")
        for f in off: setattr(model, f, False)
        st = model.encode(xyz.cuda(), rgb.cuda())
        masks, iou = model.decode(st, prompt.cuda(), labels.cuda(), None, True)
        print(f"  off={','.join(off) or '-':60s} emb {err(st.pc_embeddings, mid['pc_embeddings']):.2e} masks {err(masks, wm):.2e} iou {err(iou, wi):.2e}", flush=True)
