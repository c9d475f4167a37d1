"""Evaluation harness around the hot path: the interactive-segmentation protocol of the reference's
evaluation/eval_kitti.py (IoU after each simulated click), its input conventions and its ASCII PLY reader.

  compute_iou            pc_sam/model/loss.py:80-98
  normalize_points/colors evaluation/eval_kitti.py:73-88 (unit sphere; (rgb/255 - 0.5)/0.5)
  load_ply               pc_sam/ply_utils.py:5-29 / demo/utils.py:4-29 (ASCII, "x y z r g b" per vertex)
  apply_grouper_policy   evaluation/eval_kitti.py:352-362 (tokenizer size chosen from the point count)
  evaluate_clouds        evaluation/eval_kitti.py:343-390 (per-cloud IoU@click, running mean)
Host-side glue only (numpy / torch bookkeeping); the model call is ``model(coords=..., features=..., gt_masks=..., is_eval=True)``.
"""
from typing import Dict, Iterable, List

import numpy as np
import torch


def compute_iou(logits: torch.Tensor, targets: torch.Tensor, threshold: float = None) -> torch.Tensor:
    """IoU of thresholded logits against boolean targets over the last dimension."""
    if logits.shape != targets.shape:
        raise AssertionError((tuple(logits.shape), tuple(targets.shape)))
    if targets.dtype != torch.bool:
        raise AssertionError(targets.dtype)
    preds = logits > 0 if threshold is None else logits.sigmoid() > threshold
    return (preds & targets).sum(-1) / (preds | targets).sum(-1)


def normalize_points(points: np.ndarray) -> np.ndarray:
    """Centre at the mean and scale the farthest point onto the unit sphere."""
    if points.ndim != 2 or points.shape[1] != 3:
        raise AssertionError(points.shape)
    centred = points - points.mean(axis=0)
    return centred / np.linalg.norm(centred, ord=2, axis=1).max()


def normalize_colors(colors: np.ndarray, mean: float = 0.5, std: float = 0.5) -> np.ndarray:
    """0..255 -> the trained convention (c/255 - mean)/std."""
    c = colors / 255
    if mean is not None:
        c = c - mean
    if std is not None:
        c = c / std
    return c


def load_ply(path: str) -> np.ndarray:
    """ASCII PLY with six numbers per vertex -> float64 [n, 6] (xyz, rgb 0..255)."""
    with open(path, "r") as f:
        n = None
        for line in f:
            if "element vertex" in line:
                n = int(line.split()[2])
            if "end_header" in line:
                break
        if n is None:
            raise ValueError(f"{path}: no 'element vertex' line in the PLY header")
        pts = np.zeros((n, 6))
        for i in range(n):
            vals = f.readline().split()
            if len(vals) != 6:
                raise AssertionError(f"{path}: vertex {i} has {len(vals)} fields, expected 6")
            pts[i] = [float(v) for v in vals]
    return pts


def apply_grouper_policy(model, num_points: int) -> None:
    """Tokenizer size by cloud size, exactly the reference's run-time mutation (eval_kitti.py:352-362)."""
    g = model.pc_encoder.patch_embed.grouper
    if num_points > 30000:
        g.num_groups, g.group_size = 2048, 256
    else:
        g.num_groups = min(num_points, 2048)
        g.group_size = 256
        if num_points < 256:
            g.group_size = 2


def prepare_sample(xyz: np.ndarray, rgb: np.ndarray, instance_labels: np.ndarray, device="cuda") -> Dict[str, torch.Tensor]:
    """Raw arrays -> the model's keyword arguments; one boolean mask per instance id (eval_kitti.py:90-115)."""
    ids = [i for i in np.unique(instance_labels) if i >= 0]
    masks = np.stack([instance_labels == i for i in ids])
    return dict(
        coords=torch.tensor(normalize_points(xyz), dtype=torch.float32, device=device)[None],
        features=torch.tensor(normalize_colors(rgb), dtype=torch.float32, device=device)[None],
        gt_masks=torch.tensor(masks, dtype=torch.bool, device=device)[None],
    )


@torch.no_grad()
def evaluate_clouds(model, samples: Iterable[Dict[str, torch.Tensor]], adapt_grouper: bool = True) -> Dict[str, np.ndarray]:
    """IoU after each click, averaged over the masks of a cloud, then over clouds (eval_kitti.py:343-380)."""
    per_cloud: List[np.ndarray] = []
    for data in samples:
        if adapt_grouper:
            apply_grouper_policy(model, data["coords"].shape[1])
        outputs = model(**data, is_eval=True)
        gt = data["gt_masks"].flatten(0, 1)
        ious = [compute_iou(o["prompt_masks"], gt).float().cpu().numpy() for o in outputs]  # [iters][B*M]
        per_cloud.append(np.array(ious).mean(axis=1))
    per_cloud_a = np.array(per_cloud)
    return dict(per_cloud=per_cloud_a, mean_iou_at_click=per_cloud_a.mean(axis=0))
