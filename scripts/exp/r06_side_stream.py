"""Round 6: the per-step gather's side stream (dist.SideStreamGather) beside the pipelines.  One process, the cfg2 pipeline; after every next() a 3 MB
device copy of the results runs on a side stream and the caller's stream waits for it before the slot is submitted again (what the multi-GPU bench does
with its all_gather).  fresh = torch.cuda.Stream() as until round 5; probed = streams.side_stream()."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from point_sam_amd.streams import side_stream, streams_alias, stream_starved_by, pipeline_streams, pipeline_streams_report

args = bench.parse_args(["--workload", sys.argv[1] if len(sys.argv) > 1 else "cfg2"])
H = bench.HipHarness(args, 0, 0)
tok, dense = pipeline_streams(H.dev, 2)
null = torch.cuda.default_stream()


def run(side, steps=40):
    main = torch.cuda.current_stream()
    lat = []
    def loop(n):
        pend = []
        for _ in range(min(H.depth, n)):
            H.submit()
        for k in range(n):
            out = H.next()
            if side is not None:
                e0 = torch.cuda.Event(enable_timing=True); e0.record(main)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    c = out[0].clone()
                    ev = torch.cuda.Event(enable_timing=True); ev.record(side)
                pend.append((e0, ev, c))
                if len(pend) > 1:
                    a, b, _ = pend.pop(0)
                    main.wait_event(b)
                    lat.append((a, b))
            if k + H.depth < n:
                if side is not None:
                    main.wait_stream(side)
                H.submit()
    loop(4)
    torch.cuda.synchronize(); lat.clear()
    t0 = time.perf_counter()
    loop(steps)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    l = sorted(a.elapsed_time(b) for a, b in lat)
    return round(ms, 3), (round(l[len(l) // 2], 3), round(l[-1], 3)) if l else None


print("no side work:", run(None))
for name, mk in (("fresh torch stream", lambda: torch.cuda.Stream()), ("another fresh stream", lambda: torch.cuda.Stream()), ("a third", lambda: torch.cuda.Stream()),
                 ("a fourth", lambda: torch.cuda.Stream()), ("streams.side_stream()", lambda: side_stream(H.dev))):
    s = mk()
    ms, lat = run(s)
    print(f"{name:24s}: {ms} ms per step; side copy latency median / max {lat} ms; aliases default {streams_alias(s, null)}, dense {[streams_alias(s, d) for d in dense]}, "
          f"starved by the tokenizer stream {stream_starved_by(s, tok) if not streams_alias(s, null) else 'n/a'}", flush=True)
print(pipeline_streams_report())
