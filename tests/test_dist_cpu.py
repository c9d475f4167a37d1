"""N>1 path on CPU: 2 processes over gloo shard a batch of independent clouds, compute their shards (the CPU oracle
stands in for the device compute here -- this is a test of the sharding/gather plumbing, not a product path) and
all_gather the per-cloud logits; every rank must end up with exactly the unsharded result."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from point_sam_amd.dist import gather_results, shard_range


def test_shard_range_covers_batch():
    for total in (1, 7, 8, 64):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import pointsam_oracle as O
    from point_sam_amd import dist as psdist
    from point_sam_amd.config import get_config
    from point_sam_amd.weights import random_state_dict
    torch.set_num_threads(2)
    r, w, _ = psdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    cfg = get_config("tiny")
    sd = random_state_dict(cfg, 5)
    xyz, rgb, prompt, labels = O.synthetic_batch(total, 512, seed=9)
    lo, hi = shard_range(total, rank, world)
    masks, iou = O.predict_masks(sd, cfg, xyz[lo:hi], rgb[lo:hi], prompt[lo:hi], labels[lo:hi])
    all_masks = gather_results(masks, total)
    fin, work = gather_results(iou, total, async_op=True)
    work.wait()
    all_iou = fin()
    if rank == 0:
        ref_masks, ref_iou = O.predict_masks(sd, cfg, xyz, rgb, prompt, labels)
        # CPU BLAS may block a batch of 2 differently from a batch of 4: compare to fp32 round-off, not bitwise (the
        # bit-exact batch-independence property of the HIP path is checked in tests/test_gpu_e2e.py)
        q.put((torch.allclose(all_masks, ref_masks, atol=1e-5, rtol=0), torch.allclose(all_iou, ref_iou, atol=1e-5, rtol=0),
               tuple(all_masks.shape)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [4, 5])
def test_two_rank_gloo_gather_matches_unsharded(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok_masks, ok_iou, shape = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok_masks and ok_iou, "sharded+gathered logits differ from the unsharded batch"
    assert shape[0] == total
