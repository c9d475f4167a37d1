"""bench.py's own control flow on CPU: world_size 2 over gloo with a stand-in pipeline (`--stub`: each step's "logits" are a known
function of (rank, step)), through BOTH launch paths -- `python bench.py --gpus 2` (self-spawned ranks) and torch.distributed.run.
Checks what the driver relies on: ONE JSON line from rank 0 with n_gpus = N, every step's gathered results complete and in step order
(finish() / drain() / the gather on its side path), the rccl field, and the long `sustained` region."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def _check(res, world, steps, warmup, sustained, batch):
    assert res["n_gpus"] == world and res["steps"] == steps and res["warmup"] == warmup
    assert res["config"]["parallelism"] == f"dp{world}" and res["config"]["global_batch"] == world * batch
    assert res["scaling"] == "weak" and res["higher_is_better"] is True and res["value"] > 0
    stub = res["config"]["stub"]
    assert stub["taken"] == steps + warmup + sustained
    assert res["sustained"]["steps"] == sustained and res["sustained"]["value"] > 0
    if world > 1:
        assert stub["bad"] == 0 and stub["steps_checked"] == steps + warmup + sustained     # every step gathered, right content, right order
        assert res["rccl"]["ranks_seen"] == world and res["rccl"]["gather_ms"] > 0
    else:
        assert res["rccl"] is None
    # every rank's own rate is reported (a straggler shows); the contract's value uses the slowest rank's time
    pr = res["per_rank"]
    assert [r["rank"] for r in pr["ranks"]] == list(range(world)) and all(r["value"] > 0 and r["ms_per_step"] > 0 for r in pr["ranks"])
    assert pr["min_value"] <= pr["max_value"] and abs(pr["min_value"] * world - res["value"]) <= 1e-2 * res["value"]
    # every rank reports what it spent before its first step (the stub's stand-in: 1 ms x (rank + 1)), gathered in rank order
    su = res["startup_s"]
    assert len(su["per_rank"]) == world and [round(x["total_s"] * 1e3) for x in su["per_rank"]] == list(range(1, world + 1))
    assert su["max_total_s"] == pytest.approx(1e-3 * world)


def test_bench_self_spawns_two_ranks():
    res = _run([sys.executable, "bench.py", "--gpus", "2", "--stub", "--steps", "7", "--warmup", "2", "--sustained-steps", "5", "--batch", "3"])
    _check(res, 2, 7, 2, 5, 3)


def test_bench_under_torch_distributed_run():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    res = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                "bench.py", "--gpus", "2", "--stub", "--steps", "5", "--warmup", "1", "--sustained-steps", "4", "--batch", "2", "--slots", "2"])
    _check(res, 2, 5, 1, 4, 2)


def test_bench_eight_ranks_end_to_end():
    """BASELINE config #4's launch shape (`bench.py --gpus 8`, what the driver runs on an 8-GPU node) through the REAL file on CPU / gloo: eight
    self-spawned ranks, every step's gathered results complete and in rank order, eight distinct ranks in the collective, per-rank fields for
    all eight, one JSON line."""
    res = _run([sys.executable, "bench.py", "--gpus", "8", "--stub", "--steps", "4", "--warmup", "1", "--sustained-steps", "2", "--batch", "2"])
    _check(res, 8, 4, 1, 2, 2)
    assert res["rccl"]["ranks_seen"] == 8 and res["config"]["global_batch"] == 16


def test_bench_single_rank_stub():
    res = _run([sys.executable, "bench.py", "--stub", "--steps", "4", "--warmup", "1", "--sustained-steps", "3"])
    _check(res, 1, 4, 1, 3, 8)


def test_bench_refuses_mismatched_world():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--stub"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
