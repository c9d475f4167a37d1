#!/usr/bin/env python
"""Throughput benchmark of the Point-SAM hot path (encode + 1-prompt decode) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = one full pass (FPS -> kNN grouping -> mini-PointNet -> ViT-L -> two-way decoder -> per-point mask logits)
over one batch of synthetic clouds resident in HBM: BASELINE.json configs[1] = ViT-L, N=32768, group_number=512,
group_size=64, batch=8 per GPU, 1 point prompt, multimask output.  With N>1 every rank processes its own batch
(weak scaling, clouds are independent) and the per-cloud logits are all-gathered over RCCL each step.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md, dense peaks at 256 CUs x 2.4 GHz
F32_MFMA_PEAK_TFLOPS = 157.3    # v_mfma_f32_32x32x2_f32
BF16_MFMA_PEAK_TFLOPS = 2500.0  # v_mfma_f32_32x32x16_bf16


def cpu_baseline(cfg, sd, N, seed, iters=3):
    """The oracle (CPU restatement of the reference, PyTorch fp32 + C tokenizer) on a bounded sample: ONE cloud of the same workload,
    one warm-up run + `iters` timed runs (median).  Reported next to the GPU number; never the thing measured above."""
    from oracle import pointsam_oracle as O
    xyz, rgb, prompt, labels = O.synthetic_batch(1, N, seed=seed)
    cores = torch.get_num_threads()
    O.fps(xyz[:, :4096], 16)  # build/load the C library outside the timed region
    run = lambda: O.predict_masks(sd, cfg, xyz, rgb, prompt, labels, None, True, mode="reference")
    run()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        run()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    dt = ts[len(ts) // 2]
    return {"value": round(1.0 / dt, 4), "unit": "point-clouds/s", "cores": cores, "kind": "port", "iterations": iters,
            "seconds_per_cloud": {"median": round(dt, 3), "min": round(ts[0], 3), "max": round(ts[-1], 3)},
            "sample": f"1 cloud of the workload (ViT-L, N={N}, 512x64, 1 prompt), oracle mode='reference' (torch.cdist+topk), 1 warm-up + {iters} timed runs, "
                      f"median {dt:.1f} s; torch {torch.__version__}, {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="large")
    ap.add_argument("--points", type=int, default=32768)
    ap.add_argument("--groups", type=int, default=512)
    ap.add_argument("--group-size", type=int, default=64)
    ap.add_argument("--batch", type=int, default=8, help="clouds per GPU per step")
    ap.add_argument("--precision", default="f16x3", choices=["f32", "bf16x6", "f16x3"],
                    help="arithmetic of the large GEMMs; all three are fp32-grade and pass the same parity tests (DESIGN.md section 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=2, help="dense-stage HIP streams: batches in flight (1 = only the tokenizer of the next batch overlaps)")
    ap.add_argument("--no-pipeline", action="store_true", help="run FPS/kNN of each batch inline instead of one batch ahead on a side stream")
    ap.add_argument("--no-gemm-profile", action="store_true", help="skip the per-launch HIP-event timing of the GEMM kernel")
    ap.add_argument("--graphs", dest="graphs", action="store_true", default=True,
                    help="replay the two stages of a batch as captured HIP graphs (default; host enqueue ~0.1 ms per step)")
    ap.add_argument("--no-graphs", dest="graphs", action="store_false", help="issue every kernel launch from Python (BatchPipeline)")
    ap.add_argument("--slots", type=int, default=3, help="batches in flight with --graphs (static buffer sets)")
    ap.add_argument("--no-stage-times", action="store_true", help="skip the per-stage timing pass after the timed region")
    args = ap.parse_args()

    from point_sam_amd import dist as psdist
    rank, world, local = psdist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from point_sam_amd.synthetic import synthetic_batch   # (oracle/ is only imported by the cpu_baseline leg below)
    from point_sam_amd import get_config, ops
    from point_sam_amd.model import PointCloudSAM
    from point_sam_amd.weights import random_state_dict

    cfg = get_config(args.config, args.groups, args.group_size)
    sd = random_state_dict(cfg, seed=42)
    model = PointCloudSAM(cfg, sd, dev, precision=args.precision)
    B, N = args.batch, args.points
    xyz, rgb, prompt, labels = synthetic_batch(B, N, seed=42 + rank)
    xyz, rgb, prompt, labels = xyz.to(dev), rgb.to(dev), prompt.to(dev), labels.to(dev)
    total = B * world

    from point_sam_amd.model import BatchPipeline, GraphPipeline
    use_graphs = args.graphs and not args.no_pipeline
    pipe = BatchPipeline(model, dense_streams=args.streams) if not args.no_pipeline else None
    gpipe = GraphPipeline(model, xyz, rgb, prompt, labels, None, True, slots=args.slots, dense_streams=args.streams) if use_graphs else None

    side_gather = psdist.SideStreamGather(dev)       # N > 1: the all_gather of a step's logits runs on its own stream
    pending = []

    def finish(masks, iou):
        """Starts the gather of this step's results (side stream) and completes the previous step's: the collective of step k overlaps
        the dense stage of the batches behind it.  Returns the newest COMPLETED (gathered) results."""
        if world == 1:
            return masks, iou
        pending.append(side_gather.start((masks, iou), total))
        return side_gather.finish(pending.pop(0)) if len(pending) > 1 else (masks, iou)

    def drain():
        out = None
        while pending:
            out = side_gather.finish(pending.pop(0))
        return out

    main_stream = torch.cuda.current_stream(dev)

    def mark(events):
        if events is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record(main_stream)
            events.append(e)

    def run_graph_steps(n, events=None):
        """n full passes through the captured graphs: `slots` batches in flight, next() before the slot is submitted again."""
        out = None
        mark(events)
        for k in range(min(gpipe.depth, n)):
            gpipe.submit(xyz, rgb, prompt, labels, None, True)
        for k in range(n):
            out = finish(*gpipe.next())
            mark(events)
            if k + gpipe.depth < n:
                if side_gather.stream is not None:      # the slot's static outputs are still being read by the gather just started
                    main_stream.wait_stream(side_gather.stream)
                gpipe.submit(xyz, rgb, prompt, labels, None, True)
        return out

    def run_steps(n, prof=None, events=None):
        """n full passes (every batch is tokenized, encoded and decoded inside this call), every launch issued from Python.  With the
        pipeline the tokenizer stage of a step runs on its own stream ahead of the dense stage, and `pipe.depth` batches are in flight.
        prof: sample the GEMM launches of the LAST step (every 3rd one, HIP events on the launch stream, the other dense stream held
        off during a sampled launch)."""
        def dense_call(k, fn):   # fn enqueues the dense stage of step k
            if prof is None or k != n - 1:
                return fn()
            others = [s for i, s in enumerate(pipe.dense) if i != pipe.count % len(pipe.dense)] if (pipe is not None and pipe.dense) else []
            # sampling starts in the second half of the step (the ~150 GEMM launches of a step: the encoder's 96 large ones come in
            # block order): by then the previous batch has drained, so holding the other stream off costs no overlap
            ops._gemm_counter = 0
            ops.GEMM_PROFILE, ops.GEMM_PROFILE_EVERY, ops.GEMM_PROFILE_OTHERS, ops.GEMM_PROFILE_AFTER = prof, 3, tuple(others), 70 if others else 0
            try:
                return fn()
            finally:
                ops.GEMM_PROFILE, ops.GEMM_PROFILE_OTHERS, ops.GEMM_PROFILE_AFTER, ops.GEMM_PROFILE_EVERY = None, (), 0, 29
        out = None
        mark(events)
        if pipe is None:
            for k in range(n):
                out = finish(*dense_call(k, lambda: model.predict_masks(xyz, rgb, prompt, labels, None, True, validate=False)))
                mark(events)
            return out
        sub = lambda: pipe.submit(xyz, rgb, prompt, labels, None, True)
        for k in range(min(pipe.depth, n)):
            dense_call(k, sub) if pipe.dense else sub()
        for k in range(n):
            if k + pipe.depth < n:
                dense_call(k + pipe.depth, sub) if pipe.dense else sub()
            out = finish(*(pipe.next() if pipe.dense else dense_call(k, pipe.next)))
            mark(events)
        return out

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if args.warmup:
        run_graph_steps(args.warmup) if use_graphs else run_steps(args.warmup)
        drain()
    prof = None if args.no_gemm_profile else []
    step_events = []
    fence()
    t0 = time.perf_counter()
    # timed region: exactly --steps passes.  With graphs the per-launch GEMM sampling cannot sit inside a replay: it runs in one
    # extra eager pass AFTER the timed region (same process, same workload, same kernels); without graphs it samples the last timed step.
    out = run_graph_steps(args.steps, step_events) if use_graphs else run_steps(args.steps, prof, step_events)
    out = drain() or out                       # N > 1: the last step's gather completes inside the timed region
    t_enqueued = time.perf_counter() - t0      # host time to issue every launch of the timed region (no sync inside)
    fence()
    elapsed = time.perf_counter() - t0
    model.check_coordinate_range()
    assert torch.isfinite(out[0]).all()
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    gaps = sorted(a.elapsed_time(b) for a, b in zip(step_events[:-1], step_events[1:]))
    step_stats = {"median": round(gaps[len(gaps) // 2], 3), "min": round(gaps[0], 3), "max": round(gaps[-1], 3),
                  "note": "ms between consecutive step completions on the caller's stream (HIP events); the first steps of the region fill the pipeline"} if gaps else None
    if use_graphs and prof is not None:
        run_steps(2, prof)       # untimed eager pass for the per-launch GEMM durations (see above)
        torch.cuda.synchronize()
    stage_ms = tok_roof = None
    if rank == 0 and not args.no_stage_times:
        from point_sam_amd.profiling import stage_times, tokenizer_roofline
        stage_ms = stage_times(model, xyz, rgb, prompt, labels, passes=3, warmup=1)
        tok_roof = tokenizer_roofline(stage_ms, B, N, args.groups, args.group_size)

    roofline = None
    if prof:
        # dominant kernel = the large-GEMM kernel of the selected precision: ALGORITHMIC flops (2*M*N*K) per launch divided
        # by the measured launch duration (HIP events on the launch stream, sampled launches of the timed region)
        kind = args.precision
        sel = [(s.elapsed_time(e), f) for s, e, f, _, _, _, k in prof if k == kind and f >= 1e9]
        if sel:
            tot_ms, tot_fl = sum(m for m, _ in sel), sum(f for _, f in sel)
            ach = tot_fl / (tot_ms * 1e-3) / 1e12
            if kind == "f32":
                peak, kernel, note = F32_MFMA_PEAK_TFLOPS, "gemm_nt_kernel (v_mfma_f32_32x32x2_f32)", "f32-input MFMA dense peak"
            elif kind == "f16x3":
                peak = BF16_MFMA_PEAK_TFLOPS / 3.0
                kernel = "gemm_f16x3p_kernel (v_mfma_f32_32x32x16_f16; operands pre-packed as row-scaled hi|lo fp16, LDS-DMA ring; 3 partial products per fp32-grade product)"
                note = ("fp32-equivalent peak of the scheme = fp16 dense MFMA peak 2500 TFLOP/s / 3 executed products; "
                        f"executed matrix-pipe rate = {ach * 3:.0f} TFLOP/s = {ach * 3 / BF16_MFMA_PEAK_TFLOPS:.3f} of the fp16 peak")
            else:
                peak = BF16_MFMA_PEAK_TFLOPS / 6.0
                kernel = "gemm_bf16x6_kernel (v_mfma_f32_32x32x16_bf16, 6 partial products per fp32-accurate product)"
                note = ("fp32-equivalent peak of the scheme = bf16 dense MFMA peak 2500 TFLOP/s / 6 executed products; "
                        f"executed matrix-pipe rate = {ach * 6:.0f} TFLOP/s = {ach * 6 / BF16_MFMA_PEAK_TFLOPS:.3f} of the bf16 peak")
            traffic = None
            tfile = "r02_traffic.json" if kind == "f16x3" else "r01_v6_traffic.json"
            try:  # HBM bytes per launch from the committed rocprofv3 PMC pass of this same command (profiles/, see its _note)
                tj = json.load(open(os.path.join(ROOT, "profiles", tfile)))
                key = {"bf16x6": "void gemm_bf16x6_kernel<2, 2, 2, 2, true>", "f16x3": "gemm_f16x3p_kernel"}.get(kind)
                if key and key in tj:
                    traffic = tj[key]["hbm_bytes_per_launch"]
            except (OSError, ValueError, KeyError):
                traffic = None
            roofline = {"bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                        "traffic": traffic, "traffic_note": f"bytes/launch (launch-weighted mean over the kernel's tile configurations), rocprofv3 FETCH_SIZE(x2)+WRITE_SIZE (fabric requests, Infinity-Cache hits included), profiles/{tfile}" if traffic else None,
                        "kernel": kernel, "peak_note": note, "sampled_launches": len(sel),
                        "sampling": (("every 3rd GEMM launch of one eager (un-graphed) pass right after the timed region, HIP events on the launch stream" if use_graphs else
                                      "every 3rd GEMM launch of the last step of the timed region, HIP events on the launch stream") +
                                     ("; second half of that step only, when the previous batch has drained, and the other dense stream is held off during a sampled launch: the duration is the kernel's own" if (pipe is not None and pipe.depth > 1) else "")),
                        "avg_launch_ms": round(tot_ms / len(sel), 4), "avg_launch_gflop": round(tot_fl / len(sel) / 1e9, 3)}
            # whole-path figure of SURVEY.md 8(d): algorithmic flops of the path (3.72e11 per cloud at this workload) / step time
            if args.config == "large" and N == 32768 and args.groups == 512 and args.group_size == 64:
                roofline["whole_path_achieved"] = round(3.72e11 * total / world / (elapsed / args.steps) / 1e12, 2)
                roofline["whole_path_frac"] = round(roofline["whole_path_achieved"] / peak, 4)
            roofline["tokenizer"] = tok_roof

    if rank == 0:
        res = {
            "metric": "point-clouds/sec (encode+1-prompt decode)", "value": round(total * args.steps / elapsed, 3), "unit": "point-clouds/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16x6": "f32 (fp32 in/out/accumulate; large GEMMs as exact 3-way bf16 split x 6 MFMA products)",
                      "f16x3": "f32 (fp32 in/out/accumulate; large GEMMs as row-scaled 2-way fp16 split x 3 MFMA products, fp32-grade error)"}[args.precision],
            "data": "synthetic",
            "config": {"workload": f"ViT-{args.config} N={N} g={args.groups}x{args.group_size} batch={B}/GPU 1 point prompt multimask",
                       "global_batch": total, "parallelism": f"dp{world}", "tokenizer_pipeline": pipe is not None,
                       "batches_in_flight": (gpipe.depth if use_graphs else (pipe.depth if pipe is not None else 1)), "dense_streams": args.streams, "hip_graphs": use_graphs,
                       "host_enqueue_ms_per_step": round(t_enqueued / args.steps * 1e3, 3), "gemm_precision": args.precision, "weights": "seeded random init (no checkpoint offline)"},
            "roofline": roofline,
            "step_ms": step_stats,
            "stage_ms": stage_ms,
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(cfg, sd, N, 42)
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
