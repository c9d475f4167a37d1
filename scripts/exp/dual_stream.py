"""Does running two independent half-batches on two streams fill the GEMM tails?  (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from point_sam_amd.config import get_config
from point_sam_amd.model import PointCloudSAM
from point_sam_amd.weights import random_state_dict
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import pointsam_oracle as O
cfg = get_config("large", 512, 64)
model = PointCloudSAM(cfg, random_state_dict(cfg, 42), "cuda", precision="f16x3")
xyz, rgb, prompt, labels = [t.cuda() for t in O.synthetic_batch(8, 32768, seed=42)]
def run_full(n):
    for _ in range(n): model.predict_masks(xyz, rgb, prompt, labels, None, True, validate=False)
halves = [(xyz[i:i + 4].contiguous(), rgb[i:i + 4].contiguous(), prompt[i:i + 4].contiguous(), labels[i:i + 4].contiguous()) for i in (0, 4)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def run_dual(n):
    for _ in range(n):
        for s, h in zip(streams, halves):
            with torch.cuda.stream(s):
                model.predict_masks(*h, None, True, validate=False)
def timeit(f, n=10):
    f(2); torch.cuda.synchronize(); t0 = time.perf_counter(); f(n); torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("full batch 8, one stream      : %.2f ms/step" % timeit(run_full))
print("2 x half batch 4, two streams : %.2f ms/step" % timeit(run_dual))
print("full batch 8, one stream      : %.2f ms/step" % timeit(run_full))
print("2 x half batch 4, two streams : %.2f ms/step" % timeit(run_dual))
def run_two_full(n):   # consecutive steps alternate between two streams (each step = a full batch of 8)
    for i in range(n):
        with torch.cuda.stream(streams[i & 1]):
            model.predict_masks(xyz, rgb, prompt, labels, None, True, validate=False)
print("full batch 8, steps alternating over two streams : %.2f ms/step" % timeit(run_two_full, 12))
print("full batch 8, one stream      : %.2f ms/step" % timeit(run_full))
print("full batch 8, steps alternating over two streams : %.2f ms/step" % timeit(run_two_full, 12))
