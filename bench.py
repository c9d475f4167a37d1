#!/usr/bin/env python
"""Throughput benchmark of the Point-SAM hot path (encode + 1-prompt decode) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = one full pass (FPS -> kNN grouping -> mini-PointNet -> ViT-L -> two-way decoder -> per-point mask logits)
over one batch of synthetic clouds resident in HBM: BASELINE.json configs[1] = ViT-L, N=32768, group_number=512,
group_size=64, batch=8 per GPU, 1 point prompt, multimask output.  With N>1 every rank processes its own batch
(weak scaling, clouds are independent) and the per-cloud logits are all-gathered over RCCL each step, on a side stream.
`python bench.py --gpus N` without a launcher spawns the N ranks itself (one process per GPU, 127.0.0.1 rendezvous).
Prints ONE JSON line on rank 0.

Besides the contract fields the line carries: `roofline` (dominant GEMM kernel, live HIP-event durations), `cpu_baseline` (the oracle on
cloud 0 of the SAME batch, host cores), `parity` (the timed configuration's own output for cloud 0 against that oracle run), `sustained`
(a second, much longer timed region), `tokenizer` (FPS us/iteration, distance evaluations per second against the VALU peak), `rccl`
(N > 1: ranks seen and the time of the per-step all_gather).
"""
import argparse
import json
import os
import socket
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md, dense peaks at 256 CUs x 2.4 GHz
F32_MFMA_PEAK_TFLOPS = 157.3    # v_mfma_f32_32x32x2_f32
BF16_MFMA_PEAK_TFLOPS = 2500.0  # v_mfma_f32_32x32x16_bf16 / _f16
TRAFFIC_FILES = {"f16x3": ("r06/r06_traffic.json", "r05/r05_traffic.json", "r04/r04_traffic.json", "r03/r03_traffic.json", "r02/r02_traffic.json"), "bf16x6": ("r01/r01_v6_traffic.json",),
                 "f32": ("r01/r01_v6_traffic.json",)}


# BASELINE.json `configs` that fit one GPU (config #4 is cfg2 across 8 GPUs: --gpus 8; config #1 is the CPU plumbing case of the tests)
WORKLOADS = {
    "cfg2": dict(config="large", points=32768, groups=512, group_size=64, batch=8, clicks=1),
    "cfg3": dict(config="large", points=131072, groups=2048, group_size=256, batch=1, clicks=1),
    "cfg5": dict(config="giant", points=32768, groups=512, group_size=64, batch=1, clicks=5),
}
CLICK_LABELS = (1, 1, 0, 1, 0, 1, 1, 0)      # labels of a session's clicks (positive first; the pattern of tests/test_gpu_e2e.py's config-#5 case)


def parse_args(argv=None):
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
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle run (drops cpu_baseline and parity)")
    ap.add_argument("--cpu-baseline-iters", type=int, default=3, help="timed oracle runs (median); 1 = the single cold run is the sample (the extra legs use it)")
    ap.add_argument("--streams", type=int, default=2, help="dense-stage HIP streams: batches in flight (1 = only the tokenizer of the next batch overlaps)")
    ap.add_argument("--no-pipeline", action="store_true", help="run FPS/kNN of each batch inline instead of one batch ahead on a side stream")
    ap.add_argument("--no-gemm-profile", action="store_true", help="skip the per-launch HIP-event timing of the GEMM kernel")
    ap.add_argument("--graphs", dest="graphs", action="store_true", default=True,
                    help="replay the two stages of a batch as captured HIP graphs (default; host enqueue ~0.1 ms per step)")
    ap.add_argument("--no-graphs", dest="graphs", action="store_false", help="issue every kernel launch from Python (BatchPipeline)")
    ap.add_argument("--slots", type=int, default=3, help="batches in flight with --graphs (static buffer sets)")
    ap.add_argument("--no-stage-times", action="store_true", help="skip the per-stage timing pass after the timed region")
    ap.add_argument("--no-mfma-probe", action="store_true", help="skip the ~2 s register-only MFMA probe behind roofline.measured_mfma_ceiling_tflops")
    ap.add_argument("--sustained-steps", type=int, default=400, help="steps of the second, long timed region (0 = skip); reported under `sustained`")
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)   # CPU/gloo run of this file's control flow with a stand-in pipeline (tests)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS),
                    help="BASELINE.json configuration: cfg2 = ViT-L N=32768 512x64 batch 8 (the metric's configuration, default); cfg3 = ViT-L N=131072 2048x256 "
                         "batch 1 (large scene); cfg5 = ViT-giant N=32768 512x64 batch 1, 5-click session (encoder cached, decoder-only loop).  Sets --config / "
                         "--points / --groups / --group-size / --batch / --clicks unless those are given explicitly")
    ap.add_argument("--clicks", type=int, default=None, help="clicks per cloud: > 1 = interactive session (click t decodes with clicks 0..t and the previous best mask)")
    ap.add_argument("--other-workloads", default="cfg3,cfg5",
                    help="with the default workload on one GPU: BASELINE configurations run as short extra legs after the main measurement and reported under "
                         "`other_workloads` (value, ms_per_step, roofline.frac, parity, a one-run cpu_baseline); '' or --no-other-workloads skips them")
    ap.add_argument("--no-other-workloads", dest="other_workloads", action="store_const", const="")
    ap.add_argument("--other-steps", type=int, default=20, help="timed steps of each extra leg (after 3 warm-ups, as a default run of that workload)")
    ap.add_argument("--data", default="synthetic", choices=["synthetic", "ply"],
                    help="synthetic: uniform clouds in the unit ball; ply: the reference's six demo clouds tiled / jittered to N points (SURVEY.md 8(d))")
    args = ap.parse_args(argv)
    given = {a.split("=")[0] for a in (argv if argv is not None else sys.argv[1:]) if a.startswith("--")}
    for key, val in WORKLOADS[args.workload].items():      # the workload's geometry, unless the flag was given explicitly
        if "--" + key.replace("_", "-") not in given:
            setattr(args, key, val)
    if args.clicks is None:
        args.clicks = 1
    return args


# ------------------------------------------------------------------------------------------ launching the ranks
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_entry(rank, world, port, argv, cores):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    if cores:      # one slice of the host cores per rank: the ranks' launch threads do not migrate onto each other
        try:
            os.sched_setaffinity(0, cores)
        except OSError:
            pass
    worker(parse_args(argv))


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` with no launcher: start N ranks (torch.multiprocessing spawn context), rank r on GPU r."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    port = _free_port()
    try:
        avail = sorted(os.sched_getaffinity(0))
    except AttributeError:
        avail = list(range(os.cpu_count() or 1))
    per = len(avail) // args.gpus
    procs = []
    for r in range(args.gpus):
        cores = set(avail[r * per:(r + 1) * per]) if per >= 1 else None
        p = ctx.Process(target=_rank_entry, args=(r, args.gpus, port, argv, cores))
        p.start()
        procs.append(p)
    code = 0
    for p in procs:
        p.join()
        code = code or (p.exitcode or 0)
    return code


# ------------------------------------------------------------------------------------------ what runs a step
class HipHarness:
    """The product pipeline: GraphPipeline (default) / BatchPipeline / inline predict_masks on this rank's GPU."""

    def __init__(self, args, rank, local):
        from point_sam_amd.synthetic import synthetic_batch   # (oracle/ is only imported by the cpu_baseline leg)
        from point_sam_amd import get_config, ops
        from point_sam_amd.model import PointCloudSAM, BatchPipeline, GraphPipeline
        from point_sam_amd.weights import random_state_dict
        torch.cuda.set_device(local)
        self.args, self.ops = args, ops
        self.dev = torch.device("cuda", local)
        self.cfg = get_config(args.config, args.groups, args.group_size)
        t0 = time.perf_counter()
        self.sd = random_state_dict(self.cfg, seed=42)
        t1 = time.perf_counter()
        self.model = PointCloudSAM(self.cfg, self.sd, self.dev, precision=args.precision)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        self.seed = 42 + rank
        self.session = args.clicks > 1
        if args.data == "ply":
            from point_sam_amd.synthetic import ply_batch
            batch = ply_batch(args.batch, args.points, seed=self.seed, num_prompts=args.clicks)
        else:
            batch = synthetic_batch(args.batch, args.points, seed=self.seed, num_prompts=args.clicks)
        if self.session:
            labels = torch.tensor([CLICK_LABELS[t % len(CLICK_LABELS)] for t in range(args.clicks)], dtype=torch.int64).repeat(args.batch, 1)
            batch = (batch[0], batch[1], batch[2], labels)
        self.batch = tuple(t.to(self.dev) for t in batch)
        self.use_graphs = args.graphs and not args.no_pipeline
        # N ranks on one host: the pipelines' streams are chosen by TIMING probes (point_sam_amd/streams.py); let every rank finish its host-heavy set-up
        # (seeded weights: seconds of all cores) before any rank probes
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            torch.cuda.synchronize()
            torch.distributed.barrier()
        # a session (several decodes on one cached encoder state) runs as a captured graph or inline; the eager stream pipeline issues single decodes
        self.pipe = BatchPipeline(self.model, dense_streams=args.streams) if (not args.no_pipeline and not (self.session and not self.use_graphs)) else None
        self.gpipe = GraphPipeline(self.model, *self.batch, None, True, slots=args.slots, dense_streams=args.streams, session=self.session) if self.use_graphs else None
        self.main_stream = torch.cuda.current_stream(self.dev)
        self.depth = self.gpipe.depth if self.use_graphs else (self.pipe.depth if self.pipe is not None else 1)
        self.inline = self.pipe is None and not self.use_graphs
        self.prof = None
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        # what a rank pays before its first step: N ranks do this concurrently on one host (weight generation and packing are host + device work)
        self.startup = {"weights_s": round(t1 - t0, 3), "model_init_s": round(t2 - t1, 3), "pipeline_capture_s": round(t3 - t2, 3), "total_s": round(t3 - t0, 3)}

    def sync(self):
        torch.cuda.synchronize()

    def event(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record(self.main_stream)
        return e

    def submit(self):
        xyz, rgb, prompt, labels = self.batch
        (self.gpipe if self.use_graphs else self.pipe).submit(xyz, rgb, prompt, labels, None, True)

    def next(self):
        if self.inline:
            if self.session:
                xyz, rgb, clicks, labels = self.batch
                return self.model.click_session(self.model.encode(xyz, rgb), clicks, labels)[-1]
            return self.model.predict_masks(*self.batch, None, True, validate=False)
        return (self.gpipe if self.use_graphs else self.pipe).next()

    def before_resubmit(self, gather):
        # the slot's static outputs are still being read by the gather just started
        if self.use_graphs and gather.stream is not None:
            self.main_stream.wait_stream(gather.stream)

    def finalize(self):
        self.model.check_coordinate_range()

    # ---- measurement legs after the timed region
    def gemm_profile(self):
        """Per-launch durations of the dominant kernel: every 3rd large-GEMM launch of one eager (un-graphed) pass, HIP events on the launch
        stream, second half of the pass with the other dense stream held off (the duration is the kernel's own)."""
        ops, prof = self.ops, []
        pipe = self.pipe
        if pipe is None:
            from point_sam_amd.model import BatchPipeline
            pipe = BatchPipeline(self.model, dense_streams=1)
        xyz, rgb, prompt, labels = self.batch
        prompt, labels = prompt[:, :1].contiguous(), labels[:, :1].contiguous()      # a session's first click: the sampled launches are the encoder's
        n = 2
        c_blocks, self.model.c_blocks = self.model.c_blocks, False      # the sampler wraps the launches the Python host issues: this pass sequences the
        for k in range(n):                                              # blocks' kernels itself (the same launches psam_eva_block issues in the timed region)
            last = k == n - 1
            if last:
                others = [s for i, s in enumerate(pipe.dense) if i != pipe.count % len(pipe.dense)] if pipe.dense else []
                ops._gemm_counter = 0
                ops.GEMM_PROFILE, ops.GEMM_PROFILE_EVERY, ops.GEMM_PROFILE_OTHERS, ops.GEMM_PROFILE_AFTER = prof, 3, tuple(others), 70 if others else 0
            try:
                pipe.submit(xyz, rgb, prompt, labels, None, True)
                if not pipe.dense:
                    pipe.next()
            finally:
                if last:
                    ops.GEMM_PROFILE, ops.GEMM_PROFILE_OTHERS, ops.GEMM_PROFILE_AFTER, ops.GEMM_PROFILE_EVERY = None, (), 0, 29
        while len(pipe):
            pipe.next()
        torch.cuda.synchronize()
        self.model.c_blocks = c_blocks
        return prof

    def stage_times(self):
        from point_sam_amd.profiling import stage_times, tokenizer_metrics
        a = self.args
        xyz, rgb, prompt, labels = self.batch
        st = stage_times(self.model, xyz, rgb, prompt[:, :1].contiguous(), labels[:, :1].contiguous(), passes=3, warmup=1)
        if self.session:      # the decoder-only loop on the cached state: ms per click after the first
            enc = self.model.encode(xyz, rgb)
            self.model.click_session(enc, prompt, labels)
            per = []
            for _ in range(3):
                outs, best, N = [], None, xyz.shape[1]
                m, i = self.model.decode(enc, prompt[:, :1].contiguous(), labels[:, :1].contiguous(), None, True)
                best = torch.gather(m, 1, i.argmax(1).view(-1, 1, 1).expand(-1, 1, N))[:, 0]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for t in range(1, prompt.shape[1]):
                    m, i = self.model.decode(enc, prompt[:, : t + 1].contiguous(), labels[:, : t + 1].contiguous(), best, False)
                    best = m[:, 0]
                e1.record(); torch.cuda.synchronize()
                per.append(e0.elapsed_time(e1) / (prompt.shape[1] - 1))
            st["ms_per_additional_click_eager"] = round(sorted(per)[1], 3)
            # the same clicks 2..T as ONE captured graph on the cached state (what the timed sessions replay): the ~60 launches of a click are issued
            # by the GPU front end instead of Python, so the figure is the kernels' own time
            m, i = self.model.decode(enc, prompt[:, :1].contiguous(), labels[:, :1].contiguous(), None, True)
            best0 = torch.gather(m, 1, i.argmax(1).view(-1, 1, 1).expand(-1, 1, xyz.shape[1]))[:, 0].contiguous()
            from point_sam_amd.streams import pipeline_streams
            side = pipeline_streams(self.dev, 2)[1][0]      # a stream of the process's probed set (point_sam_amd/streams.py)
            side.wait_stream(torch.cuda.current_stream())

            def later_clicks():
                best = best0
                for t in range(1, prompt.shape[1]):
                    mm, _ = self.model.decode(enc, prompt[:, : t + 1].contiguous(), labels[:, : t + 1].contiguous(), best, False)
                    best = mm[:, 0]
                return best

            counters = self.ops.new_counters(self.dev)      # the graph's own arrival-counter block: it may replay on any stream, beside anything
            with torch.cuda.stream(side), self.ops.use_counters(counters):
                later_clicks()      # eager once on the capture stream: one-time set-up must not happen inside the capture
            side.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side), self.ops.use_counters(counters):
                last = later_clicks()
            torch.cuda.synchronize()
            per_g = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
                per_g.append(e0.elapsed_time(e1) / (prompt.shape[1] - 1))
            del last
            st["ms_per_additional_click"] = round(sorted(per_g)[2], 3)
            st["ms_per_additional_click_note"] = ("clicks 2..T on the cached encoder state replayed as one HIP graph (median of 5 replays); "
                                                  "`ms_per_additional_click_eager` = the same clicks issued launch by launch from Python (median of 3)")
        traffic = None
        for cand in TRAFFIC_FILES["f16x3"]:
            try:
                traffic = json.load(open(os.path.join(ROOT, "profiles", cand)))
                break
            except (OSError, ValueError):
                continue
        return st, tokenizer_metrics(st, a.batch, a.points, a.groups, a.group_size, traffic)

    def cpu_baseline(self, out, iters=3):
        """The oracle (CPU restatement of the reference, PyTorch fp32 + C tokenizer) on a bounded sample: cloud 0 of this very batch, one
        warm-up run + `iters` timed runs (median).  Its logits are then the checker of the timed configuration's own output for that cloud
        (`parity`).  Reported next to the GPU number; never the thing measured above."""
        from oracle import pointsam_oracle as O
        a = self.args
        full = tuple(t.cpu() for t in self.batch)      # the very tensors the timed region ran on
        xyz, rgb, prompt, labels = (t[:1] for t in full)
        cores = torch.get_num_threads()
        O.fps(xyz[:, :4096], 16)  # build/load the C library outside the timed region
        # cpu_baseline TIMES the reference's own arithmetic (mode="reference": kNN / 3-NN through torch.cdist + topk, common.py:51-55,238-255); the
        # parity CHECK uses mode="exact" (direct fp32 differences -- what the HIP kernels, the C oracle and the bit-exact index tests implement).
        # cdist's |a|^2 + |b|^2 - 2ab form rounds distances of near neighbours differently: same neighbour sets at cfg #2
        # (tests/test_gpu_e2e.py::test_cfg2_gap_to_reference_cdist_mode), not at cfg #3's 256-neighbour groups in a 131072-point cloud, where the
        # reference-mode logits sit ~6e-3 from both the exact-mode oracle and the HIP path: reported as `reference_mode_gap`.
        if self.session:
            oracle_one = lambda b, mode="exact": O.click_loop(self.sd, self.cfg, full[0][b:b + 1], full[1][b:b + 1], full[2][b:b + 1], full[3][b:b + 1], mode=mode)[-1]
        else:
            oracle_one = lambda b, mode="exact": O.predict_masks(self.sd, self.cfg, full[0][b:b + 1], full[1][b:b + 1], full[2][b:b + 1], full[3][b:b + 1], None, True, mode=mode)
        run = lambda: oracle_one(0, "reference")
        t0 = time.perf_counter()
        want_ref = run()
        first = time.perf_counter() - t0
        want = oracle_one(0)
        ts = []
        for _ in range(iters if iters > 1 else 0):
            t0 = time.perf_counter()
            run()
            ts.append(time.perf_counter() - t0)
        if not ts:      # a one-run leg: the first (cold) run is the sample
            ts = [first]
        ts.sort()
        dt = ts[len(ts) // 2]
        base = {"value": round(1.0 / dt, 4), "unit": "point-clouds/s", "cores": cores, "kind": "port", "iterations": len(ts),
                "seconds_per_cloud": {"median": round(dt, 3), "min": round(ts[0], 3), "max": round(ts[-1], 3)},
                "sample": f"cloud 0 of the timed batch (ViT-{a.config}, N={a.points}, {a.groups}x{a.group_size}, {a.clicks} click(s)), oracle "
                          + ("click loop, " if self.session else "") + "mode='reference' (torch.cdist+topk), "
                          + (f"1 warm-up + {len(ts)} timed runs, median {dt:.1f} s" if iters > 1 else f"one (cold) run, {dt:.1f} s")
                          + f"; torch {torch.__version__}, {cores} threads"}
        masks, iou = out
        masks, iou = masks.float().cpu(), iou.float().cpu()
        per_cloud, scale = [], 0.0
        for b in range(masks.shape[0]):      # EVERY cloud of the timed batch against the oracle (clouds 1.. : one untimed oracle run each)
            w = want if b == 0 else oracle_one(b)
            per_cloud.append((float((masks[b:b + 1] - w[0]).abs().max()), float((iou[b:b + 1] - w[1]).abs().max())))
            scale = max(scale, float(w[0].abs().max()))
        em, ei = max(e for e, _ in per_cloud), max(e for _, e in per_cloud)
        # Against the REFERENCE's own kNN arithmetic (torch.cdist's matmul form + topk, common.py:51-55).  Where cdist's rounding selects the same neighbour
        # sets as the direct differences (cfg #2), the reference-mode logits must themselves be within tolerance.  Where it does not (cfg #3: ties at the
        # 256-th distance of a 131072-point cloud), the two ORACLE modes differ by what the swapped neighbours are worth, and the HIP path -- bit-exact
        # with the exact mode's indices -- may be that far from the reference mode, not farther: |HIP - ref| <= |exact - ref| + 1e-4.
        st0 = self.model.encode(self.batch[0][:1].contiguous(), self.batch[1][:1].contiguous())
        ref_idx = O.knn(O.batch_index_select(xyz, O.fps(xyz, a.groups)), xyz, a.group_size, "reference")[1]
        same = (st0.knn_idx.cpu().sort(-1).values == ref_idx.sort(-1).values).all(-1)
        gap_hip = float((masks[:1] - want_ref[0]).abs().max())
        gap_oracles = float((want[0] - want_ref[0]).abs().max())
        sets_identical = bool(same.all())
        ok_ref = gap_hip < 1e-3 if sets_identical else gap_hip <= gap_oracles + 1e-4
        ok_exact = bool(em < 1e-3 and ei < 1e-3)
        parity = {"checked": f"all {masks.shape[0]} cloud(s) of the last timed step's output (the graph/stream pipeline's own result"
                             + (", last click of the session" if self.session else "") + ") vs the oracle on the same inputs and weights",
                  "max_abs_err_mask_logits": em, "max_abs_err_iou": ei, "per_cloud_max_abs_err_mask_logits": [round(e, 9) for e, _ in per_cloud],
                  "tolerance": 1e-3, "ok": bool(ok_exact and ok_ref), "ok_vs_exact_oracle": ok_exact, "ok_vs_reference": bool(ok_ref), "logit_scale": scale,
                  "oracle_mode": "exact (direct fp32 coordinate differences in kNN / 3-NN)",
                  "reference_mode_gap": {"cloud": 0, "max_abs_err_mask_logits": gap_hip, "oracle_exact_vs_reference": gap_oracles,
                                         "groups_with_other_knn_set": int((~same).sum()), "groups": int(same.numel()),
                                         "rule": "identical neighbour sets: |HIP - reference| < 1e-3; otherwise |HIP - reference| <= |exact - reference| + 1e-4 "
                                                 "(the gap is then a property of the two oracle modes: cdist's rounding picks other neighbours)",
                                         "note": "same cloud against the oracle in mode='reference' (torch.cdist + topk, the arithmetic cpu_baseline times)"}}
        return base, parity


class StubHarness:
    """CPU stand-in with the same interface (tests/test_bench_cpu.py): `depth` batches in flight, each step's "logits" are a known
    function of (rank, step), so that the gathered results of every step can be checked for content and order."""

    def __init__(self, args, rank, local):
        self.args, self.rank, self.depth = args, rank, max(1, args.slots)
        self.submitted, self.taken, self.queue = 0, 0, []
        self.dev = torch.device("cpu")
        self.use_graphs, self.prof = True, None
        self.startup = {"weights_s": 0.0, "model_init_s": 0.0, "pipeline_capture_s": 0.0, "total_s": 0.001 * (rank + 1)}

    def sync(self):
        pass

    def event(self):
        return time.perf_counter()

    def submit(self):
        if len(self.queue) >= self.depth:
            raise RuntimeError("stub: more batches in flight than slots")
        self.queue.append(self.submitted)
        self.submitted += 1

    def next(self):
        k = self.queue.pop(0)
        self.taken += 1
        B = self.args.batch
        base = torch.arange(B, dtype=torch.float32).view(B, 1, 1) + 100.0 * self.rank + 10000.0 * k
        return base.expand(B, 3, 16).contiguous(), base.view(B, 1).expand(B, 3).contiguous()

    def before_resubmit(self, gather):
        pass

    def finalize(self):
        pass

    @staticmethod
    def expected(world, B, k):
        return torch.cat([torch.arange(B, dtype=torch.float32) + 100.0 * r + 10000.0 * k for r in range(world)])


# ------------------------------------------------------------------------------------------ one rank
def worker(args):
    from point_sam_amd import dist as psdist
    rank, world, local = psdist.init_from_env(backend="gloo" if args.stub else None)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, or without a launcher")
    H = (StubHarness if args.stub else HipHarness)(args, rank, local)
    B, total = args.batch, args.batch * world
    side_gather = psdist.SideStreamGather(None if args.stub else H.dev)       # N > 1: the all_gather of a step's logits runs on its own stream
    pending, checks = [], {"steps_checked": 0, "bad": 0}

    def verify(out, k):
        if args.stub and world > 1 and out is not None:
            checks["steps_checked"] += 1
            want = StubHarness.expected(world, B, k)
            if out[0].shape[0] != total or not torch.equal(out[0][:, 0, 0], want) or not torch.equal(out[1][:, 0], want):
                checks["bad"] += 1

    def finish(masks, iou, k):
        """Starts the gather of this step's results (side stream) and completes the previous step's: the collective of step k overlaps
        the dense stage of the batches behind it.  Returns the newest COMPLETED (gathered) results."""
        if world == 1:
            return masks, iou
        pending.append((side_gather.start((masks, iou), total), k))
        if len(pending) > 1:
            h, kk = pending.pop(0)
            out = side_gather.finish(h)
            verify(out, kk)
            return out
        return masks, iou

    def drain():
        out = None
        while pending:
            h, kk = pending.pop(0)
            out = side_gather.finish(h)
            verify(out, kk)
        return out

    counter = {"k": 0}

    def run_steps(n, events=None):
        """n full passes, `depth` batches in flight: next() before the slot is submitted again."""
        out = None
        if events is not None:
            events.append(H.event())
        if getattr(H, "inline", False):
            for _ in range(n):
                out = finish(*H.next(), counter["k"]); counter["k"] += 1
                if events is not None:
                    events.append(H.event())
            return out
        for _ in range(min(H.depth, n)):
            H.submit()
        for k in range(n):
            out = finish(*H.next(), counter["k"]); counter["k"] += 1
            if events is not None:
                events.append(H.event())
            if k + H.depth < n:
                H.before_resubmit(side_gather)
                H.submit()
        return out

    def fence():
        if world > 1:
            torch.distributed.barrier()
        H.sync()

    def timed(n, events=None):
        fence()
        t0 = time.perf_counter()
        out = run_steps(n, events)
        out = drain() or out                       # N > 1: the last step's gather completes inside the timed region
        t_enq = time.perf_counter() - t0           # host time to issue every launch of the region (no sync inside)
        fence()
        el = time.perf_counter() - t0
        per_rank = [el]
        if world > 1:
            t = torch.tensor([el], device=H.dev, dtype=torch.float64)
            allt = [torch.zeros_like(t) for _ in range(world)]
            torch.distributed.all_gather(allt, t)      # every rank's own time between the fences: a straggler shows
            per_rank = [float(x.item()) for x in allt]
            el = max(per_rank)                          # the contract's figure: MAX over ranks
        timed.per_rank = per_rank
        return out, el, t_enq

    if args.warmup:
        run_steps(args.warmup)
        drain()
    step_events = []
    # timed region: exactly --steps passes
    out, elapsed, t_enqueued = timed(args.steps, step_events)
    rank_times = list(timed.per_rank)
    H.finalize()
    assert torch.isfinite(out[0]).all()
    keep = (out[0][:B].clone(), out[1][:B].clone()) if world == 1 else (out[0][rank * B:(rank + 1) * B].clone(), out[1][rank * B:(rank + 1) * B].clone())

    if args.stub:
        gaps = sorted(b - a for a, b in zip(step_events[:-1], step_events[1:]))
        gaps = [g * 1e3 for g in gaps]
    else:
        gaps = sorted(a.elapsed_time(b) for a, b in zip(step_events[:-1], step_events[1:]))
    step_stats = {"median": round(gaps[len(gaps) // 2], 3), "min": round(gaps[0], 3), "max": round(gaps[-1], 3),
                  "note": "ms between consecutive step completions on the caller's stream (HIP events); the first steps of the region fill the pipeline"} if gaps else None

    # second, long timed region: the same passes, enough of them that clock / power state is the sustained one
    sustained = None
    if args.sustained_steps > 0:
        clocks, power, sampler = [], None, None
        if not args.stub and rank == 0:
            from point_sam_amd.profiling import PowerSampler
            sampler = PowerSampler(local).start()
        _, el2, _ = timed(args.sustained_steps)
        if not args.stub:
            from point_sam_amd.profiling import read_sclk_mhz
            clocks = read_sclk_mhz(local)
            power = sampler.stop() if sampler is not None else None
        sustained = {"steps": args.sustained_steps, "seconds": round(el2, 3), "value": round(total * args.sustained_steps / el2, 3), "unit": "point-clouds/s",
                     "ms_per_step": round(el2 / args.sustained_steps * 1e3, 3), "vs_timed_region": round((total * args.sustained_steps / el2) / (total * args.steps / elapsed), 4),
                     "sclk_mhz_after": clocks or None, "power": power,
                     "note": "same passes, run right after the contract's timed region with the same fences; `value` above stays the contract's K-step figure; "
                             "`power` = package power and shader clock of this GPU sampled during these steps (first 0.5 s dropped)"}

    # the collective on its own: ranks seen + time of the per-step gather of the logits
    rccl = None
    if world > 1:
        ids = psdist.gather_results(torch.full((1,), float(rank), device=H.dev), world)
        fence()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            h = side_gather.start(keep, total)
            side_gather.finish(h)
        fence()
        seen = int(torch.unique(ids.cpu()).numel())
        if seen != world:      # a rank missing from the collective: the run does not count
            raise SystemExit(f"rccl: {seen} distinct ranks in the gathered results, expected {world}")
        rccl = {"ranks_seen": seen, "gather_ms": round((time.perf_counter() - t0) / reps * 1e3, 4),
                "gather_bytes_per_rank": int(keep[0].numel() * 4 + keep[1].numel() * 4), "backend": torch.distributed.get_backend(),
                "note": "all_gather_into_tensor of one step's [B,3,N] logits + [B,3] IoU on the side stream, 10 back-to-back, host-timed between barriers"}

    # every rank's set-up time (weights, model, graph capture) -- N ranks pay it concurrently on one host
    startup = [H.startup]
    if world > 1:
        t = torch.tensor([H.startup[k] for k in ("weights_s", "model_init_s", "pipeline_capture_s", "total_s")], device=H.dev, dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(world)]
        torch.distributed.all_gather(allt, t)
        startup = [dict(zip(("weights_s", "model_init_s", "pipeline_capture_s", "total_s"), (round(float(v), 3) for v in x.tolist()))) for x in allt]
        # Nothing below is collective: the ranks part here, so that the measurement legs that only rank 0 runs (GEMM launch sampling, stage times)
        # do not keep N - 1 processes waiting at a final barrier.
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
        if rank != 0:
            return

    prof = stage_ms = tok = None
    if not args.stub:
        if not args.no_gemm_profile:
            prof = H.gemm_profile()
        if not args.no_stage_times:
            stage_ms, tok = H.stage_times()
    roofline = build_roofline(args, prof, total, world, elapsed) if prof else None
    if roofline is not None and not args.stub:
        add_power_ceiling(roofline, args, (sustained or {}).get("power"), local)

    if rank == 0:
        a = args
        res = {
            "metric": "point-clouds/sec (encode+1-prompt decode)", "value": round(total * a.steps / elapsed, 3), "unit": "point-clouds/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16x6": "f32 (fp32 in/out/accumulate; large GEMMs as exact 3-way bf16 split x 6 MFMA products)",
                      "f16x3": "f32 (fp32 in/out/accumulate; large GEMMs as row-scaled 2-way fp16 split x 3 MFMA products, fp32-grade error)"}[a.precision],
            "data": "synthetic" if a.data == "synthetic" else "reference demo clouds tiled/jittered to N (synthetic weights)",
            "config": {"workload": f"{a.workload}: ViT-{a.config} N={a.points} g={a.groups}x{a.group_size} batch={B}/GPU "
                                   + (f"{a.clicks}-click session (encoder cached)" if a.clicks > 1 else "1 point prompt multimask") + f", {a.data} clouds",
                       "global_batch": total, "parallelism": f"dp{world}", "tokenizer_pipeline": not a.no_pipeline,
                       "batches_in_flight": H.depth, "dense_streams": a.streams, "hip_graphs": bool(H.use_graphs),
                       "host_enqueue_ms_per_step": round(t_enqueued / a.steps * 1e3, 3), "gemm_precision": a.precision, "weights": "seeded random init (no checkpoint offline)",
                       "streams": streams_report(args)},
            "roofline": roofline, "step_ms": step_stats, "stage_ms": stage_ms, "tokenizer": tok, "sustained": sustained, "rccl": rccl,
            "per_rank": {"ranks": [{"rank": r, "value": round(B * a.steps / t, 3), "ms_per_step": round(t / a.steps * 1e3, 3)} for r, t in enumerate(rank_times)],
                         "min_value": round(B * a.steps / max(rank_times), 3), "max_value": round(B * a.steps / min(rank_times), 3),
                         "note": "each rank's own clouds/s between the fences of the timed region; `value` uses the slowest rank's time"},
            "startup_s": {"per_rank": startup, "max_total_s": max(x["total_s"] for x in startup),
                          "note": "before the first step: seeded weights on the host, PointCloudSAM.__init__ (upload + packing of the GEMM weights), pipeline set-up "
                                  "incl. HIP-graph capture; the ranks of a multi-GPU run do this concurrently on the shared host"},
        }
        if a.clicks > 1:
            res["config"]["clicks"] = a.clicks
            res["ms_per_additional_click"] = (stage_ms or {}).get("ms_per_additional_click")
        if args.stub:
            res["config"]["stub"] = dict(checks, taken=H.taken)
            res["data"] = "stub"
        elif world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"], res["parity"] = H.cpu_baseline(keep, iters=a.cpu_baseline_iters)
        if not args.stub and world == 1 and a.workload == "cfg2" and a.other_workloads:
            # BASELINE configs #3 / #5 as short legs of the SAME run AND PROCESS, so that whoever runs the default command sees them (each is also a
            # full contract line of its own under --workload)
            import gc
            H.pipe = H.gpipe = H.model = None      # the main workload's graphs and weights are not needed any more
            gc.collect()
            torch.cuda.empty_cache()
            res["other_workloads"] = {w: other_leg(w, a, local, (roofline or {}).get("measured_mfma_ceiling_tflops")) for w in a.other_workloads.split(",")
                                      if w in WORKLOADS and w != a.workload}
        print(json.dumps(res), flush=True)


def streams_report(args):
    if args.stub:
        return None
    from point_sam_amd.streams import pipeline_streams_report
    return pipeline_streams_report()


def other_leg(name, base, local, ceiling=None):
    """One BASELINE configuration as a short leg of THIS process: a second model and a second set of graphs on the process's pipeline streams
    (point_sam_amd/streams.py -- until round 5 a leg had to be a fresh subprocess: its pipeline got other streams and ran 1.3 - 2 x slower,
    profiles/r06/r06_inproc.txt).  `--other-steps` timed passes after 3 warm-ups, ~2 s more of the same passes with package power sampled, the GEMM launch
    sampling, one oracle run as the CPU baseline and the checker."""
    import gc
    import math
    argv = ["--workload", name, "--steps", str(base.other_steps), "--warmup", "3", "--precision", base.precision, "--streams", str(base.streams), "--slots", str(base.slots),
            "--cpu-baseline-iters", "1"]
    for flag, on in (("--no-graphs", not base.graphs), ("--no-gemm-profile", base.no_gemm_profile), ("--no-cpu-baseline", base.no_cpu_baseline), ("--no-mfma-probe", True)):
        if on:
            argv.append(flag)
    a = parse_args(argv)
    try:
        H = HipHarness(a, 0, local)

        def loop(n):
            out = None
            if H.inline:
                for _ in range(n):
                    out = H.next()
                return out
            for _ in range(min(H.depth, n)):
                H.submit()
            for k in range(n):
                out = H.next()
                if k + H.depth < n:
                    H.submit()
            return out

        loop(a.warmup)
        H.sync()
        t0 = time.perf_counter()
        out = loop(a.steps)
        H.sync()
        el = time.perf_counter() - t0
        H.finalize()
        assert torch.isfinite(out[0]).all()
        keep = (out[0].clone(), out[1].clone())
        leg = {"workload": f"{name}: ViT-{a.config} N={a.points} g={a.groups}x{a.group_size} batch={a.batch}/GPU "
                           + (f"{a.clicks}-click session (encoder cached)" if a.clicks > 1 else "1 point prompt multimask") + f", {a.data} clouds",
               "value": round(a.batch * a.steps / el, 3), "unit": "sessions/s" if a.clicks > 1 else "point-clouds/s", "ms_per_step": round(el / a.steps * 1e3, 3),
               "steps": a.steps, "warmup": a.warmup, "batches_in_flight": H.depth, "startup_s": H.startup, "in_process": True}
        from point_sam_amd.profiling import PowerSampler
        n2 = max(a.steps, int(math.ceil(2.0 / (el / a.steps))))
        ps = PowerSampler(local).start()
        H.sync()
        t0 = time.perf_counter()
        loop(n2)
        H.sync()
        el2 = time.perf_counter() - t0
        power = ps.stop()
        leg["sustained"] = {"steps": n2, "seconds": round(el2, 3), "value": round(a.batch * n2 / el2, 3), "ms_per_step": round(el2 / n2 * 1e3, 3), "power": power}
        if not a.no_gemm_profile:
            r = build_roofline(a, H.gemm_profile(), a.batch, 1, el)
            if r:
                leg["roofline"] = {k: r[k] for k in ("bound", "achieved", "peak", "unit", "frac", "sampled_launches", "avg_launch_ms", "avg_launch_gflop")}
                leg["roofline"]["power"] = power
                products = {"f16x3": 3.0, "bf16x6": 6.0}.get(a.precision)
                if ceiling and products:      # the probe ran once, after the main workload: the same box, the same minute
                    leg["roofline"]["measured_mfma_ceiling_tflops"] = ceiling
                    leg["roofline"]["frac_of_measured_ceiling"] = round(r["achieved"] * products / ceiling, 4)
        if not a.no_cpu_baseline:
            leg["cpu_baseline"], leg["parity"] = H.cpu_baseline(keep, iters=1)
        return leg
    except Exception as e:      # a leg must not take the main line down
        import traceback
        return {"error": repr(e), "traceback_tail": traceback.format_exc()[-800:]}
    finally:
        H = None
        gc.collect()
        torch.cuda.empty_cache()


def add_power_ceiling(roofline, args, power, local):
    """What the hardware delivers to this arithmetic on THIS box: package power / clock while the workload ran, and the matrix pipe's
    own sustained rate (register-only MFMA probe, ~2 s) as a second yardstick beside the nominal peak.  `peak` / `frac` stay nominal."""
    from point_sam_amd.profiling import mfma_ceiling
    roofline["power"] = power
    products = {"f16x3": 3.0, "bf16x6": 6.0}.get(args.precision)
    if products is None or args.no_mfma_probe:
        return
    c = mfma_ceiling(2.0, local)
    if c is None:
        return
    roofline["measured_mfma_ceiling_tflops"] = c["tflops"]
    roofline["measured_mfma_ceiling"] = c
    roofline["executed_tflops"] = round(roofline["achieved"] * products, 1)
    roofline["frac_of_measured_ceiling"] = round(roofline["achieved"] * products / c["tflops"], 4)
    roofline["ceiling_note"] = ("measured_mfma_ceiling_tflops = fp16 MFMA products per second of a register-only loop on this box right after the run (no memory traffic; operands "
                                "with the GEMM's hi/lo value distribution), i.e. what clock and power management leave of the nominal 2500; frac_of_measured_ceiling = "
                                "executed products of the dominant kernel (achieved x products per fp32-grade product) / that ceiling")


def build_roofline(args, prof, total, world, elapsed):
    # dominant kernel = the large-GEMM kernel of the selected precision: ALGORITHMIC flops (2*M*N*K) per launch divided
    # by the measured launch duration (HIP events on the launch stream, sampled launches)
    kind = args.precision
    sel = [(s.elapsed_time(e), f) for s, e, f, _, _, _, k in prof if k == kind and f >= 1e9]
    if not sel:
        return None
    tot_ms, tot_fl = sum(m for m, _ in sel), sum(f for _, f in sel)
    ach = tot_fl / (tot_ms * 1e-3) / 1e12
    if kind == "f32":
        peak, kernel, note = F32_MFMA_PEAK_TFLOPS, "gemm_nt_kernel (v_mfma_f32_32x32x2_f32)", "f32-input MFMA dense peak"
    elif kind == "f16x3":
        peak = BF16_MFMA_PEAK_TFLOPS / 3.0
        kernel = "gemm_f16x3p / gemm_f16x3pp kernels (v_mfma_f32_32x32x16_f16; operands pre-packed as row-scaled hi|lo fp16, LDS-DMA ring; 3 partial products per fp32-grade product)"
        note = ("fp32-equivalent peak of the scheme = fp16 dense MFMA peak 2500 TFLOP/s / 3 executed products; "
                f"executed matrix-pipe rate = {ach * 3:.0f} TFLOP/s = {ach * 3 / BF16_MFMA_PEAK_TFLOPS:.3f} of the fp16 peak")
    else:
        peak = BF16_MFMA_PEAK_TFLOPS / 6.0
        kernel = "gemm_bf16x6_kernel (v_mfma_f32_32x32x16_bf16, 6 partial products per fp32-accurate product)"
        note = ("fp32-equivalent peak of the scheme = bf16 dense MFMA peak 2500 TFLOP/s / 6 executed products; "
                f"executed matrix-pipe rate = {ach * 6:.0f} TFLOP/s = {ach * 6 / BF16_MFMA_PEAK_TFLOPS:.3f} of the bf16 peak")
    traffic = tfile = None
    for cand in TRAFFIC_FILES[kind]:   # HBM bytes per launch from the committed rocprofv3 PMC pass of this same command (profiles/, see its _note)
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", cand)))
            key = {"bf16x6": "void gemm_bf16x6_kernel<2, 2, 2, 2, true>", "f16x3": "gemm_f16x3p_kernel"}.get(kind)
            if key and key in tj:
                traffic, tfile = tj[key]["hbm_bytes_per_launch"], cand
                break
        except (OSError, ValueError, KeyError):
            continue
    roofline = {"bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                "traffic": traffic, "traffic_note": f"bytes/launch (launch-weighted mean over the kernel's tile configurations), rocprofv3 FETCH_SIZE(x2)+WRITE_SIZE (fabric requests, Infinity-Cache hits included), profiles/{tfile}" if traffic else None,
                "kernel": kernel, "peak_note": note, "sampled_launches": len(sel),
                "sampling": "every 3rd GEMM launch of one eager (un-graphed) pass right after the timed regions, HIP events on the launch stream; second half of that pass only, "
                            "when the previous batch has drained, and the other dense stream is held off during a sampled launch: the duration is the kernel's own",
                "avg_launch_ms": round(tot_ms / len(sel), 4), "avg_launch_gflop": round(tot_fl / len(sel) / 1e9, 3)}
    # whole-path figure of SURVEY.md 8(d): algorithmic flops of the path (3.72e11 per cloud at this workload) / step time
    if args.config == "large" and args.points == 32768 and args.groups == 512 and args.group_size == 64:
        roofline["whole_path_achieved"] = round(3.72e11 * total / world / (elapsed / args.steps) / 1e12, 2)
        roofline["whole_path_frac"] = round(roofline["whole_path_achieved"] / peak, 4)
    return roofline


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args, argv))
    worker(args)


if __name__ == "__main__":
    main()
