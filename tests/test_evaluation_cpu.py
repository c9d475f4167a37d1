"""Host-side evaluation utilities (CPU): PLY reader, normalisation conventions, IoU, tokenizer-size policy."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from point_sam_amd import evaluation as E


def test_load_ply_and_normalisation(tmp_path):
    rng = np.random.default_rng(0)
    pts = np.concatenate([rng.normal(size=(50, 3)) * 3 + 5, rng.integers(0, 256, size=(50, 3))], 1)
    p = tmp_path / "cloud.ply"
    with open(p, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex 50\nproperty float x\nproperty float y\nproperty float z\n"
                "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n")
        for r in pts:
            f.write("%f %f %f %d %d %d\n" % tuple(r))
    got = E.load_ply(str(p))
    assert got.shape == (50, 6) and np.allclose(got, pts, atol=1e-5)
    xyz = E.normalize_points(got[:, :3])
    assert np.allclose(xyz.mean(0), 0, atol=1e-12) and np.isclose(np.linalg.norm(xyz, axis=1).max(), 1.0)
    rgb = E.normalize_colors(got[:, 3:])
    assert rgb.min() >= -1 and rgb.max() <= 1 and np.isclose(E.normalize_colors(np.array([255.0]))[0], 1.0)
    bad = tmp_path / "bad.ply"
    bad.write_text("ply\nend_header\n")
    with pytest.raises(ValueError):
        E.load_ply(str(bad))


def test_compute_iou():
    logits = torch.tensor([[2.0, -1.0, 3.0, -2.0], [-1.0, -1.0, 1.0, 1.0]])
    target = torch.tensor([[True, True, False, False], [False, False, True, True]])
    assert torch.allclose(E.compute_iou(logits, target), torch.tensor([1 / 3, 1.0]))
    assert torch.allclose(E.compute_iou(logits, target, threshold=0.5), torch.tensor([1 / 3, 1.0]))
    with pytest.raises(AssertionError):
        E.compute_iou(logits, target.float())


def test_grouper_policy():
    m = SimpleNamespace(pc_encoder=SimpleNamespace(patch_embed=SimpleNamespace(grouper=SimpleNamespace(num_groups=512, group_size=64))))
    g = m.pc_encoder.patch_embed.grouper
    E.apply_grouper_policy(m, 50000); assert (g.num_groups, g.group_size) == (2048, 256)
    E.apply_grouper_policy(m, 10000); assert (g.num_groups, g.group_size) == (2048, 256)
    E.apply_grouper_policy(m, 1000); assert (g.num_groups, g.group_size) == (1000, 256)
    E.apply_grouper_policy(m, 100); assert (g.num_groups, g.group_size) == (100, 2)


@pytest.mark.parametrize("big_endian", [False, True])
def test_binary_ply_round_trip_and_crop_loader(tmp_path, big_endian):
    """Binary PLY reader (evaluation/eval_kitti.py:117-241) on files written here: every scalar type, both byte orders, the
    labelled-crop convention (x y z R G B label), the fixed scene rotation, ASCII files rejected like the reference does."""
    rng = np.random.default_rng(3)
    n = 321
    fields = {"x": rng.normal(size=n).astype(np.float32), "y": rng.normal(size=n).astype(np.float32), "z": rng.normal(size=n).astype(np.float32),
              "R": rng.integers(0, 256, n).astype(np.uint8), "G": rng.integers(0, 256, n).astype(np.uint8), "B": rng.integers(0, 256, n).astype(np.uint8),
              "label": (rng.random(n) > 0.6).astype(np.int32), "t": rng.normal(size=n), "s": rng.integers(-9, 9, n).astype(np.int16)}
    p = str(tmp_path / "crop.ply")
    E.write_ply_binary(p, fields, big_endian=big_endian)
    data = E.read_ply_binary(p)
    assert len(data) == n and set(data.dtype.names) == set(fields)
    for k, v in fields.items():
        assert np.array_equal(data[k], v), k
    crop = E.load_labelled_crop(p)
    xyz = np.column_stack([fields["x"], fields["y"], fields["z"]])
    assert crop["xyz"].dtype == np.float32 and np.allclose(crop["xyz"], xyz @ E.scene_rotation().T, atol=1e-6)
    rot = E.scene_rotation()
    assert np.allclose(rot @ rot.T, np.eye(3)) and np.isclose(np.linalg.det(rot), 1.0)
    assert np.allclose(rot @ np.array([0.0, 0.0, 1.0]), [0.0, 1.0, 0.0])         # z-up scenes become y-up ...
    assert np.allclose(rot @ np.array([1.0, 0.0, 0.0]), [-1.0, 0.0, 0.0])        # ... turned half a turn about the new vertical
    assert np.array_equal(crop["mask"], fields["label"]) and crop["rgb"].max() <= 255
    s = E.crop_to_sample(crop, device="cpu")
    assert s["coords"].shape == (1, n, 3) and s["features"].shape == (1, n, 3) and s["gt_masks"].shape == (1, 1, n) and s["gt_masks"].dtype == torch.bool
    assert np.isclose(s["coords"].norm(dim=-1).max().item(), 1.0, atol=1e-6) and s["features"].abs().max() <= 1
    a = tmp_path / "ascii.ply"
    a.write_text("ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nend_header\n0.5\n")
    with pytest.raises(ValueError):
        E.read_ply_binary(str(a))


def test_binary_ply_mesh(tmp_path):
    v = np.zeros(4, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4")])
    v["x"] = [0, 1, 0, 1]; v["y"] = [0, 0, 1, 1]
    faces = np.array([(3, 0, 1, 2), (3, 1, 3, 2)], dtype=[("k", "<u1"), ("v1", "<i4"), ("v2", "<i4"), ("v3", "<i4")])
    p = tmp_path / "mesh.ply"
    with open(p, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\n"
                b"element face 2\nproperty list uchar int vertex_indices\nend_header\n")
        v.tofile(f); faces.tofile(f)
    verts, tri = E.read_ply_binary(str(p), triangular_mesh=True)
    assert np.array_equal(verts["x"], v["x"]) and np.array_equal(tri, [[0, 1, 2], [1, 3, 2]])


def test_instance_mask_filter_and_object_means():
    masks = np.zeros((4, 200), dtype=bool)
    masks[0, :10] = True          # too few points (< 25)
    masks[1, :60] = True          # kept
    masks[2, :185] = True         # >= 90 % of the cloud
    masks[3, 100:130] = True      # kept
    kept = E.filter_instance_masks(masks)
    assert kept.shape == (2, 200) and np.array_equal(kept[0], masks[1]) and np.array_equal(kept[1], masks[3])
    assert E.filter_instance_masks(masks[:1]).shape == (0, 200)

    class Stub:      # a "model" whose click-i prediction has IoU (i+1)/4 with the ground truth of cloud c scaled by c's quality
        prompt_iters = 3
        pc_encoder = SimpleNamespace(patch_embed=SimpleNamespace(grouper=SimpleNamespace(num_groups=1, group_size=1)))
        def __call__(self, coords, features, gt_masks, is_eval):
            gt = gt_masks.flatten(0, 1)
            outs = []
            for i in range(3):
                logit = torch.full(gt.shape, -1.0)
                k = int(gt.sum()) * (i + 1) // 4
                idx = gt[0].nonzero()[:k, 0]
                logit[0, idx] = 1.0
                outs.append(dict(prompt_masks=logit))
            return outs
    samples, names = [], []
    for c, nm in enumerate(["car", "car", "tree"]):
        gt = torch.zeros(1, 1, 400, dtype=torch.bool); gt[0, 0, : 100 + 100 * c] = True
        samples.append(dict(coords=torch.zeros(1, 400, 3), features=torch.zeros(1, 400, 3), gt_masks=gt)); names.append(nm)
    res = E.evaluate_clouds(Stub(), samples, adapt_grouper=False, names=names)
    assert res["per_cloud"].shape == (3, 3) and np.allclose(res["per_cloud"][0], [0.25, 0.5, 0.75])
    assert set(res["per_object"]) == {"car", "tree"}
    assert np.allclose(res["per_object"]["car"], res["per_cloud"][:2].mean(0))
    assert np.allclose(res["object_mean_iou_at_click"], (res["per_object"]["car"] + res["per_object"]["tree"]) / 2)
