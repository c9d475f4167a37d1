"""The variants' oracle (oracle/variants_oracle.py) against golden vectors produced by the reference's own PointCloudSAMNN / PointCloudSAMHier
classes (tests/golden/make_golden.py variants).  CPU only."""
import pytest
import torch

from oracle import pointsam_oracle as O
from oracle import variants_oracle as V
from point_sam_amd.config import get_config
from point_sam_amd.weights import check_state_dict, expected_shapes, random_state_dict, state_dict_checksum


def _setup(golden):
    meta, a = golden
    cfg = get_config(meta["cfg"])
    sd = random_state_dict(cfg, seed=meta["seed"])
    assert state_dict_checksum(sd) == pytest.approx(meta["weights_checksum"], rel=1e-12), "weight RNG drifted"
    return meta, a, cfg, sd


def test_variant_state_dict_names():
    """The variants' parameter names and shapes: loaded with strict=True into the reference's classes by the golden generator; here the
    counts and a few telling entries."""
    v, h = expected_shapes(get_config("voronoi")), expected_shapes(get_config("hier"))
    assert v["pc_encoder.patch_embed.in_proj.weight"] == (256, 7) and v["mask_encoder.first_nn.weight"] == (1024, 5)
    assert v["mask_encoder.second_nn.mlp.6.weight"] == (256, 1024) and "mask_encoder.patch_encoder.conv1.0.weight" not in v
    assert h["pc_encoder.patch_embed.patch_encoder2.conv1.0.weight"] == (128, 131) and h["mask_decoder.output_upscaling2.0.weight"] == (256, 384)
    assert h["mask_decoder.output_hypernetworks_mlps.0.layers.2.weight"] == (128, 256) and h["mask_decoder.output_upscaling1.3.weight"] == (128, 128)
    for name in ("tiny_voronoi", "tiny_hier"):
        cfg = get_config(name)
        check_state_dict(cfg, random_state_dict(cfg, 3))


@pytest.mark.parametrize("mode,tol", [("reference", 3e-4), ("exact", 1e-3)])
def test_voronoi_oracle_matches_reference(golden_voronoi, mode, tol):
    meta, a, cfg, sd = _setup(golden_voronoi)
    st = V.encode(sd, cfg, a["xyz"], a["rgb"], mode)
    p = st["patches"]
    assert torch.equal(p["centers"], a["centers"]) and torch.equal(p["nn_idx"], a["nn_idx"])
    torch.testing.assert_close(p["features"], a["group_features"], atol=1e-6, rtol=1e-6)
    torch.testing.assert_close(p["embeddings"], a["patch_embeddings"], atol=5e-5, rtol=1e-4)
    torch.testing.assert_close(st["pc_embeddings"], a["pc_embeddings"], atol=1e-4, rtol=1e-4)
    masks, iou = V.decode(sd, cfg, st, a["prompt_coords"], a["prompt_labels"], None, True, mode)
    assert (masks - a["masks_click1"]).abs().max() < tol and (iou - a["iou_click1"]).abs().max() < tol
    masks2, iou2, dense = V.decode(sd, cfg, st, a["prompt_coords"], a["prompt_labels"], a["prompt_masks_click2"], False, mode, return_dense=True)
    torch.testing.assert_close(dense, a["dense_click2"], atol=1e-4, rtol=1e-4)
    assert (masks2 - a["masks_click2"]).abs().max() < tol and (iou2 - a["iou_click2"]).abs().max() < tol


def test_voronoi_forward_eval_matches_reference(golden_voronoi):
    """PointCloudSAMNN.forward(..., is_eval=True): the same clicks, then the same logits, iteration by iteration."""
    meta, a, cfg, sd = _setup(golden_voronoi)
    outs = V.forward_eval(sd, cfg, a["xyz"], a["rgb"], a["gt_masks"].bool(), prompt_iters=meta["iters"], mode="reference")
    for i, o in enumerate(outs):
        assert torch.equal(o["prompt_coords"], a[f"fwd_prompt_coords_{i}"]) and torch.equal(o["prompt_labels"], a[f"fwd_prompt_labels_{i}"].bool()), i
        assert (o["masks"] - a[f"fwd_masks_{i}"]).abs().max() < 5e-4 and (o["iou_preds"] - a[f"fwd_iou_preds_{i}"]).abs().max() < 5e-4, i


@pytest.mark.parametrize("mode,tol", [("reference", 3e-4), ("exact", 1e-3)])
def test_hier_oracle_matches_reference(golden_hier, mode, tol):
    meta, a, cfg, sd = _setup(golden_hier)
    st = V.encode(sd, cfg, a["xyz"], a["rgb"], mode)
    p1, p2 = st["patches"]
    assert torch.equal(p1["centers"], a["centers1"]) and torch.equal(p2["centers"], a["centers2"])
    assert torch.equal(p1["knn_idx"].sort(-1).values, a["knn_idx1"].sort(-1).values) and torch.equal(p2["knn_idx"].sort(-1).values, a["knn_idx2"].sort(-1).values)
    torch.testing.assert_close(p1["embeddings"], a["embeddings1"], atol=5e-5, rtol=1e-4)
    torch.testing.assert_close(p2["embeddings"], a["patch_embeddings"], atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(st["pc_embeddings"], a["pc_embeddings"], atol=1e-4, rtol=1e-4)
    masks, iou = V.decode(sd, cfg, st, a["prompt_coords"], a["prompt_labels"], None, True, mode)
    assert (masks - a["masks_click1"]).abs().max() < tol and (iou - a["iou_click1"]).abs().max() < tol
    masks2, iou2, dense = V.decode(sd, cfg, st, a["prompt_coords"], a["prompt_labels"], a["prompt_masks_click2"], False, mode, return_dense=True)
    torch.testing.assert_close(dense, a["dense_click2"], atol=1e-4, rtol=1e-4)
    assert (masks2 - a["masks_click2"]).abs().max() < tol and (iou2 - a["iou_click2"]).abs().max() < tol
