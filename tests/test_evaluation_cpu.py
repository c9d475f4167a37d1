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
