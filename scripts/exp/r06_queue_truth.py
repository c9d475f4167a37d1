"""Ground truth for the stream -> hardware-queue mapping: run under AMD_LOG_LEVEL=4, each stream of interest launches one kernel with a unique grid size
(a fill of 256 * tag elements), the ROCclr log names the software / hardware queue of every dispatch."""
import sys
import torch

junk = int(sys.argv[1])
torch.cuda.init()
x = torch.zeros(8, device="cuda")
for s in [torch.cuda.Stream() for _ in range(junk)] + [torch.cuda.Stream(priority=-1) for _ in range(2 if junk else 0)]:
    with torch.cuda.stream(s):
        x.add_(1)
torch.cuda.synchronize()
sys.path.insert(0, ".")
from point_sam_amd.streams import pipeline_streams, pipeline_streams_report
tok, dense = pipeline_streams("cuda:0", 2)
print("POOL", pipeline_streams_report(), flush=True)
bufs = {}
for tag, s in (("null", torch.cuda.default_stream()), ("tok", tok), ("d0", dense[0]), ("d1", dense[1])):
    n = {"null": 1001, "tok": 1003, "d0": 1005, "d1": 1007}[tag]
    bufs[tag] = torch.empty(256 * n, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        bufs[tag].fill_(1.0)
    torch.cuda.synchronize()
    print(f"TAG {tag} grid {256 * n}", flush=True)
