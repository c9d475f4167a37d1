"""The voronoi (PointCloudSAMNN) and hier (PointCloudSAMHier) model variants on the HIP library (point_sam_amd/variants.py) against
  (a) the golden vectors produced by the reference's own classes (tests/golden/ref_tiny_voronoi.npz, ref_tiny_hier.npz), and
  (b) the CPU oracle (oracle/variants_oracle.py) on seeded inputs at sizes where the large-GEMM paths are active.
Tolerance 1e-3 on mask logits / IoU; FPS, nearest-centre and kNN indices bit-exact."""
from dataclasses import replace

import pytest
import torch

from oracle import pointsam_oracle as O
from oracle import variants_oracle as V
from point_sam_amd.config import VIT_TINY_SWIGLU, ModelConfig, get_config
from point_sam_amd.weights import random_state_dict

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(scope="module")
def build():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from point_sam_amd.variants import build_model
    return build_model


def _err(a, b):
    return (a.detach().cpu().double() - b.detach().cpu().double()).abs().max().item()


@pytest.mark.parametrize("which,precision", [("golden_voronoi", "f32"), ("golden_voronoi", "f16x3"), ("golden_hier", "f32"), ("golden_hier", "f16x3")])
def test_variants_against_reference_golden(build, which, precision, request):
    meta, a = request.getfixturevalue(which)
    cfg = get_config(meta["cfg"])
    sd = random_state_dict(cfg, seed=meta["seed"])
    model = build(cfg, sd, "cuda", precision=precision)
    st = model.encode(a["xyz"].cuda(), a["rgb"].cuda())
    if cfg.variant == "voronoi":
        assert torch.equal(st.centers.cpu(), a["centers"]) and torch.equal(st.knn_idx.cpu(), a["nn_idx"])
    else:
        assert torch.equal(st.extra["centers1"].cpu(), a["centers1"]) and torch.equal(st.centers.cpu(), a["centers2"])
        assert torch.equal(st.extra["knn1"].cpu().sort(-1).values, a["knn_idx1"].sort(-1).values)
        assert torch.equal(st.knn_idx.cpu().sort(-1).values, a["knn_idx2"].sort(-1).values)
        assert _err(st.extra["embeddings1"], a["embeddings1"]) < 2e-4
    assert _err(st.patch_embeddings, a["patch_embeddings"]) < 5e-4 and _err(st.pc_embeddings, a["pc_embeddings"]) < 5e-4
    pc, pl = a["prompt_coords"].cuda(), a["prompt_labels"].cuda()
    m1, i1 = model.decode(st, pc, pl, None, True)
    m2, i2 = model.decode(st, pc, pl, a["prompt_masks_click2"].cuda(), False)
    errs = [_err(m1, a["masks_click1"]), _err(i1, a["iou_click1"]), _err(m2, a["masks_click2"]), _err(i2, a["iou_click2"])]
    print(f"\n[{which} {precision}] max|err| click1 masks {errs[0]:.2e} iou {errs[1]:.2e} | click2 masks {errs[2]:.2e} iou {errs[3]:.2e}")
    assert max(errs) < TOL, errs


def test_voronoi_forward_eval_against_reference_golden(build, golden_voronoi):
    """PointCloudSAMNN.forward(..., is_eval=True): the same simulated clicks and logits as the reference's own run, iteration by iteration."""
    meta, a = golden_voronoi
    cfg = get_config(meta["cfg"])
    model = build(cfg, random_state_dict(cfg, seed=meta["seed"]), "cuda", precision="f16x3")
    model.prompt_iters = meta["iters"]
    outs = model(a["xyz"].cuda(), a["rgb"].cuda(), a["gt_masks"].bool().cuda(), is_eval=True)
    for i, o in enumerate(outs):
        assert torch.equal(o["prompt_coords"].cpu(), a[f"fwd_prompt_coords_{i}"]) and torch.equal(o["prompt_labels"].cpu(), a[f"fwd_prompt_labels_{i}"].bool()), i
        assert _err(o["masks"], a[f"fwd_masks_{i}"]) < TOL and _err(o["iou_preds"], a[f"fwd_iou_preds_{i}"]) < TOL, i


@pytest.mark.parametrize("name", ["tiny_hier", "tiny_voronoi"])
def test_variant_forward_eval_against_oracle(build, name):
    """The click loop (encoder once, then prompt_iters x {simulated click from the error region, decode with all clicks and the previous best
    mask}) of both variants against the oracle's loop: the same clicks bit for bit, logits within tolerance."""
    cfg = get_config(name)
    sd = random_state_dict(cfg, seed=41)
    xyz, rgb, _, _ = O.synthetic_batch(2, 900, seed=42)
    gt = torch.stack([xyz[..., 0] > 0.1, (xyz - torch.tensor([0.2, 0.1, -0.1])).norm(dim=-1) < 0.3], 1)
    want = V.forward_eval(sd, cfg, xyz, rgb, gt, prompt_iters=3, mode="exact")
    model = build(cfg, sd, "cuda", precision="f16x3")
    model.prompt_iters = 3
    outs = model(xyz.cuda(), rgb.cuda(), gt.cuda(), is_eval=True)
    for i, (o, w) in enumerate(zip(outs, want)):
        assert torch.equal(o["prompt_coords"].cpu(), w["prompt_coords"]) and torch.equal(o["prompt_labels"].cpu(), w["prompt_labels"]), i
        assert _err(o["masks"], w["masks"]) < TOL and _err(o["iou_preds"], w["iou_preds"]) < TOL, i


CASES = {
    # wide enough that the packed-operand GEMMs run (rows >= 256, widths >= 128)
    "voronoi_256": lambda: ModelConfig(VIT_TINY_SWIGLU, 64, 1, in_channels=7, prompt_iters=3, variant="voronoi", nn_hidden=256),
    "hier_mid": lambda: ModelConfig(VIT_TINY_SWIGLU, 64, 16, prompt_iters=3, variant="hier", hier_groups=(256, 64), hier_sizes=(16, 16), hier_radius=(0.1, 0.2)),
    "hier_noradius": lambda: ModelConfig(VIT_TINY_SWIGLU, 32, 8, prompt_iters=3, variant="hier", hier_groups=(128, 32), hier_sizes=(8, 8), hier_radius=None),
}


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("name,B,N,M", [("voronoi_256", 2, 3000, 1), ("voronoi_256", 1, 2048, 3), ("hier_mid", 2, 4096, 2), ("hier_noradius", 1, 1500, 1)])
def test_variants_against_oracle(build, name, B, N, M, precision):
    cfg = CASES[name]()
    sd = random_state_dict(cfg, seed=31)
    xyz, rgb, prompt, labels = O.synthetic_batch(B, N, seed=32, num_prompts=2)
    prompt, labels = prompt.repeat_interleave(M, 0), labels.repeat_interleave(M, 0)
    st_o = V.encode(sd, cfg, xyz, rgb, "exact")
    wm, wi = V.decode(sd, cfg, st_o, prompt, labels, None, True, "exact")
    best = torch.gather(wm, 1, wi.argmax(1).view(-1, 1, 1).expand(-1, 1, N))[:, 0]
    wm2, wi2 = V.decode(sd, cfg, st_o, prompt, labels, best, False, "exact")
    model = build(cfg, sd, "cuda", precision=precision)
    st = model.encode(xyz.cuda(), rgb.cuda())
    if cfg.variant == "voronoi":
        assert torch.equal(st.knn_idx.cpu(), st_o["patches"]["nn_idx"]) and torch.equal(st.fps_idx.cpu(), st_o["patches"]["fps_idx"])
    else:
        assert torch.equal(st.extra["knn1"].cpu(), st_o["patches"][0]["knn_idx"]) and torch.equal(st.knn_idx.cpu(), st_o["patches"][1]["knn_idx"])
    m1, i1 = model.decode(st, prompt.cuda(), labels.cuda(), None, True)
    m2, i2 = model.decode(st, prompt.cuda(), labels.cuda(), best.cuda(), False)
    errs = [_err(st.pc_embeddings, st_o["pc_embeddings"]), _err(m1, wm), _err(i1, wi), _err(m2, wm2), _err(i2, wi2)]
    print(f"\n[{name} B={B} N={N} M={M} {precision}] max|err| emb {errs[0]:.2e} | click1 {errs[1]:.2e} {errs[2]:.2e} | click2 {errs[3]:.2e} {errs[4]:.2e} "
          f"(|logit| max {wm.abs().max():.1f})")
    assert max(errs[1:]) < TOL, errs
    # predict_masks = encode + decode; results do not depend on what was decoded before
    pm, pi = model.predict_masks(xyz.cuda(), rgb.cuda(), prompt.cuda(), labels.cuda())
    assert torch.equal(pm, m1) and torch.equal(pi, i1)


def test_scatter_amax_and_group_feats_kernels(build):
    """psam_scatter_amax against torch.scatter_reduce('amax') in both include_self modes (negative values, empty cells, no batch offset) and
    psam_nn_group_feats against the oracle's expressions."""
    from point_sam_amd import ops
    g = torch.Generator().manual_seed(5)
    B, N, G, C = 3, 777, 40, 33
    x = torch.randn(B * N, C, generator=g) - 0.5
    x[::7, ::5] = -0.0                                           # negative zeros must still beat negative values
    idx = torch.randint(0, G - 3, (B, N), generator=g)          # the last three cells stay empty
    want = x.view(B, N, C).new_zeros(B, G, C).scatter_reduce(1, idx.unsqueeze(-1).expand(B, N, C), x.view(B, N, C), "amax", include_self=False)
    got = ops.scatter_amax(x.cuda(), idx.cuda(), B * G, rows_per_set=N, set_stride=G, include_self=False)
    assert torch.equal(got.cpu().view(B, G, C), want)
    want0 = torch.zeros(B * G, C).scatter_reduce(0, idx.reshape(-1, 1).expand(-1, C), x, "amax")       # as MaskEncoderNN does it: no batch offset
    got0 = ops.scatter_amax(x.cuda(), idx.cuda(), B * G, rows_per_set=N, set_stride=0, include_self=True)
    assert torch.equal(got0.cpu(), want0)
    # NaN propagates like torch's amax (whatever arrives before or after it); an index outside the output is skipped, never written through
    xn = x.clone(); xn[5, 2] = float("nan"); xn[B * N - 1, 0] = float("nan")
    want_n = x.view(B, N, C).new_zeros(B, G, C).scatter_reduce(1, idx.unsqueeze(-1).expand(B, N, C), xn.view(B, N, C), "amax", include_self=False)
    got_n = ops.scatter_amax(xn.cuda(), idx.cuda(), B * G, rows_per_set=N, set_stride=G, include_self=False).cpu().view(B, G, C)
    assert torch.equal(torch.isnan(got_n), torch.isnan(want_n)) and int(torch.isnan(want_n).sum()) == 2
    assert torch.equal(torch.nan_to_num(got_n, nan=0.0), torch.nan_to_num(want_n, nan=0.0))
    bad = idx.clone(); bad[0, :5] = 10 ** 9; bad[1, :3] = -(10 ** 9)
    keep = torch.ones(B, N, dtype=torch.bool); keep[0, :5] = False; keep[1, :3] = False
    xs = torch.where(keep.view(-1, 1), x, torch.full_like(x, -float("inf")))
    want_b = x.view(B, N, C).new_zeros(B, G, C).scatter_reduce(1, idx.unsqueeze(-1).expand(B, N, C), xs.view(B, N, C), "amax", include_self=False)
    want_b = torch.where(torch.isinf(want_b), torch.zeros_like(want_b), want_b)
    guard = torch.full((B * G + 8, C), 123.0, device="cuda")
    from point_sam_amd import _lib
    L = _lib.load()
    xc, bc = x.cuda(), bad.cuda()
    ops.check(L.psam_scatter_amax(xc.data_ptr(), C, bc.data_ptr(), B * N, C, N, G, 1, guard.data_ptr(), B * G, 0, None), "psam_scatter_amax")
    torch.cuda.synchronize()
    assert torch.equal(guard[: B * G].cpu().view(B, G, C), want_b) and bool((guard[B * G:] == 123.0).all())
    xyz = torch.rand(B, N, 3, generator=g) * 2 - 1
    centers, feats = xyz[:, :G].contiguous(), torch.rand(B, N, 3, generator=g)
    _, nn = O.knn(xyz, centers, 1, "exact")
    nn = nn.squeeze(-1)
    nbr, dist = V.nn_offsets(xyz, centers, nn)
    f = ops.nn_group_feats(xyz.cuda(), centers.cuda(), nn.cuda(), feats=feats.cuda(), width=8).cpu().view(B, N, 8)
    torch.testing.assert_close(f[..., :7], torch.cat([nbr / torch.clamp(dist, min=1e-8), dist, feats], -1), atol=1e-6, rtol=1e-6)
    assert (f[..., 7] == 0).all()
    logits = torch.randn(B * 2, N, generator=g)
    m = ops.nn_group_feats(xyz.cuda(), centers.cuda(), nn.cuda(), logits=logits.cuda(), width=8).cpu().view(B * 2, N, 8)
    torch.testing.assert_close(m[..., :5], torch.cat([logits.unsqueeze(-1), nbr.repeat_interleave(2, 0), dist.repeat_interleave(2, 0)], -1), atol=1e-6, rtol=1e-6)


@pytest.mark.parametrize("dense_streams", [1, 2])
@pytest.mark.parametrize("name", ["tiny_hier", "tiny_voronoi"])
def test_variants_through_batch_pipeline(build, name, dense_streams):
    """The variants through BatchPipeline (tokenizer on its own stream, several batches in flight): results bit-identical to predict_masks.
    The hier tokenizer state carries its level-1 tensors in `extra` and leaves the interpolation fields empty (TokenizerState.tensors must
    cover the first and skip the second); an EncoderState owns its own copy of `extra` (a later encode() on the same tokenizer state must
    not replace the level-1 embeddings an earlier state decodes with)."""
    from point_sam_amd.model import BatchPipeline
    cfg = get_config(name)
    model = build(cfg, random_state_dict(cfg, seed=5), "cuda", precision="f16x3")
    batches = []
    for i in range(5):
        xyz, rgb, prompt, labels = O.synthetic_batch(2, 1500 + 300 * i, seed=60 + i)
        batches.append(tuple(t.cuda() for t in (xyz, rgb, prompt, labels)))
    want = [model.predict_masks(*b) for b in batches]
    pipe = BatchPipeline(model, dense_streams=dense_streams)
    got = []
    for k in range(min(pipe.depth, len(batches))):
        pipe.submit(*batches[k])
    for k in range(len(batches)):
        if k + pipe.depth < len(batches):
            pipe.submit(*batches[k + pipe.depth])
        got.append(pipe.next())
    torch.cuda.synchronize()
    for k, ((m1, i1), (m2, i2)) in enumerate(zip(want, got)):
        assert torch.equal(m1, m2) and torch.equal(i1, i2), (k, _err(m1, m2), _err(i1, i2))
    # two encoder states from ONE tokenizer state: the first keeps decoding with its own level-1 embeddings
    xyz, rgb, prompt, labels = batches[0]
    tok = model.tokenize(xyz)
    st1 = model.encode(xyz, rgb, tok)
    ref = model.decode(st1, prompt, labels, None, True)
    model.encode(xyz, rgb.flip(0).contiguous(), tok)          # different features through the same tokenizer state
    again = model.decode(st1, prompt, labels, None, True)
    assert torch.equal(ref[0], again[0]) and torch.equal(ref[1], again[1])
