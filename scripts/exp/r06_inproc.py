"""Round 6, VERDICT r05 item 1: why does a second workload in one process run 1.3-2x slower than the same workload in a fresh process?

    python scripts/exp/r06_inproc.py cfg2,cfg5,cfg3 [--teardown] [--steps 20] [--eager-first]

Runs the named bench workloads one after the other in THIS process (bench.HipHarness, the graph pipeline, the bench's own step loop) and prints,
per workload, its rate and the HIP streams (torch pool ids + raw handles) its pipelines were given.  Discriminators: GPU_MAX_HW_QUEUES in the
environment, --teardown (drop the previous harness, gc, empty the caching allocator before the next), the order of the workloads."""
import argparse
import gc
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def run(H, steps, warmup=3):
    def loop(n):
        out = None
        for _ in range(min(H.depth, n)):
            H.submit()
        for k in range(n):
            out = H.next()
            if k + H.depth < n:
                H.submit()
        return out
    loop(warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = loop(steps)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    assert torch.isfinite(out[0]).all()
    return el / steps * 1e3


def graph_alone(H):
    """One slot's dense graph (and its tokenizer graph) replayed ALONE on its stream: ms per replay (median of 5), nothing else on the device."""
    if H.gpipe is None:
        return None
    st = H.gpipe.slots[0]
    res = {}
    for name, g, s in (("tok", st.g_tok, H.gpipe.tok_stream), ("dense", st.g_dense, st.ds)):
        ts = []
        for _ in range(6):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(s):
                e0.record(s); g.replay(); e1.record(s)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res[name] = round(sorted(ts[1:])[2], 3)
    return res


def alias_of(H):
    from point_sam_amd.streams import streams_alias
    p = H.gpipe
    ss = {"null": torch.cuda.default_stream(), "tok": p.tok_stream, "d0": p.dense[0], "d1": p.dense[1]}
    names = list(ss)
    return [f"{a}~{b}" for i, a in enumerate(names) for b in names[i + 1:] if streams_alias(ss[a], ss[b])]


def streams_of(H):
    d = {}
    for name, p in (("pipe", H.pipe), ("gpipe", H.gpipe)):
        if p is None:
            continue
        d[name] = {"tok": (p.tok_stream.stream_id, hex(p.tok_stream.cuda_stream)), "dense": [(s.stream_id, hex(s.cuda_stream)) for s in p.dense]}
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workloads")
    ap.add_argument("--teardown", action="store_true")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--recheck", action="store_true", help="after the last workload, time every kept harness again (first one included)")
    ap.add_argument("--extra", action="store_true", help="also run the bench's post-region legs (gemm_profile, stage_times) of each workload before the next one")
    ap.add_argument("--junk", type=int, default=0, help="first use this many normal + 2 high-priority pool streams (an application's own streams)")
    a = ap.parse_args()
    if a.junk:
        x = torch.zeros(8, device="cuda")
        for s in [torch.cuda.Stream() for _ in range(a.junk)] + [torch.cuda.Stream(priority=-1) for _ in range(2)]:
            with torch.cuda.stream(s):
                x.add_(1)
        torch.cuda.synchronize()
    res = []
    keep = []
    for w in a.workloads.split(","):
        args = bench.parse_args(["--workload", w])
        H = bench.HipHarness(args, 0, 0)
        rates = [round(run(H, a.steps), 3) for _ in range(a.repeat)]
        from point_sam_amd.streams import mini_pipeline_ms
        mini = mini_pipeline_ms(H.gpipe.tok_stream, H.gpipe.dense)
        rec = {"workload": w, "mini_pipeline_ms": [round(v, 3) for v in mini], "ms_per_step": rates, "graph_alone_ms": graph_alone(H), "aliased_pairs": alias_of(H), "streams": streams_of(H), "mem_GB": round(torch.cuda.memory_allocated() / 2**30, 2),
               "reserved_GB": round(torch.cuda.memory_reserved() / 2**30, 2)}
        if a.extra:
            H.gemm_profile()
            H.stage_times()
            rec["ms_per_step_after_extra"] = round(run(H, a.steps), 3)
        print(json.dumps(rec), flush=True)
        res.append(rec)
        if a.teardown:
            H.pipe = H.gpipe = None
            del H
            gc.collect()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
        else:
            keep.append(H)
    if keep and a.recheck:
        for H in keep:
            print(json.dumps({"recheck": H.args.workload, "ms_per_step": round(run(H, a.steps), 3), "graph_alone_ms": graph_alone(H)}), flush=True)
    from point_sam_amd.streams import pipeline_streams_report
    print("POOL", pipeline_streams_report())
    print("SUMMARY", os.environ.get("GPU_MAX_HW_QUEUES"), "teardown" if a.teardown else "keep", " ".join(f"{r['workload']}={r['ms_per_step'][-1]}" for r in res), flush=True)


if __name__ == "__main__":
    main()
