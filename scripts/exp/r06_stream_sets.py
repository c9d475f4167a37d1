"""Round 6: WHICH property of a stream set makes a graph pipeline lose its concurrency?  One giant-model (cfg5) session pipeline per stream set, all in
one process, all streams created (and first used, in creation order) up front.  Prints ms per session for each (tok; dense0, dense1; main) choice."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from point_sam_amd.model import GraphPipeline, streams_alias  # noqa: E402

args = bench.parse_args(["--workload", sys.argv[1] if len(sys.argv) > 1 else "cfg5"])
os.environ["PSAM_PRIVATE_STREAMS"] = "1"
args.graphs = False; args.no_pipeline = True
H = bench.HipHarness(args, 0, 0)          # model + batch only
N, NH = 14, 5
ns = [torch.cuda.Stream() for _ in range(N)]
hs = [torch.cuda.Stream(priority=-1) for _ in range(NH)]
x = torch.zeros(8, device="cuda")
for s in ns + hs:                           # first use in creation order: fixes the stream -> hardware queue mapping
    with torch.cuda.stream(s):
        x.add_(1)
torch.cuda.synchronize()
name = {id(s): f"n{i}" for i, s in enumerate(ns)}
name.update({id(s): f"h{i}" for i, s in enumerate(hs)})


def rate(tok, d0, d1, main=None, steps=16):
    ctx = torch.cuda.stream(main) if main is not None else torch.cuda.stream(torch.cuda.default_stream())
    with ctx:
        gp = GraphPipeline(H.model, *H.batch, None, True, slots=4, dense_streams=2, session=H.session, streams=(tok, [d0, d1]))
        def loop(n):
            out = None
            for _ in range(min(gp.depth, n)):
                gp.submit(*H.batch, None, True)
            for k in range(n):
                out = gp.next()
                if k + gp.depth < n:
                    gp.submit(*H.batch, None, True)
            return out
        loop(3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop(steps)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
    del gp
    return round(ms, 3)


cases = [("h0", "n0", "n1", None), ("h0", "n2", "n3", None), ("h0", "n4", "n5", None), ("h0", "n6", "n7", None), ("h0", "n8", "n9", None), ("h0", "n0", "n2", None),
         ("h0", "n0", "n3", None), ("h0", "n1", "n2", None), ("h1", "n0", "n1", None), ("h2", "n0", "n1", None), ("h3", "n0", "n1", None), ("h4", "n0", "n1", None),
         ("h3", "n6", "n7", None), ("n10", "n0", "n1", None), ("h0", "n0", "n1", "n12"), ("h0", "n6", "n7", "n12"), ("h0", "n0", "n1", None)]
byname = {v: s for s in ns + hs for k, v in name.items() if k == id(s)}
for tok, d0, d1, main in cases:
    ms = rate(byname[tok], byname[d0], byname[d1], byname[main] if main else None)
    print(json.dumps({"tok": tok, "dense": [d0, d1], "main": main or "null", "ms_per_step": ms}), flush=True)
