import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from point_sam_amd.config import get_config
from point_sam_amd.model import PointCloudSAM
from point_sam_amd.weights import random_state_dict
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import pointsam_oracle as O
cfg = get_config("large", 512, 64)
model = PointCloudSAM(cfg, random_state_dict(cfg, 42), "cuda", precision="f16x3")
xyz, rgb, prompt, labels = [t.cuda() for t in O.synthetic_batch(8, 32768, seed=42)]
for _ in range(2): model.predict_masks(xyz, rgb, prompt, labels, None, True, validate=False)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    model.predict_masks(xyz, rgb, prompt, labels, None, True, validate=False)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=3).table(sort_by="count", row_limit=40, max_name_column_width=40, max_src_column_width=90))
