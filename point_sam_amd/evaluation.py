"""Evaluation harness around the hot path: the interactive-segmentation protocol of the reference's
evaluation/eval_kitti.py (IoU after each simulated click), its input conventions and its ASCII PLY reader.

  compute_iou            pc_sam/model/loss.py:80-98
  normalize_points/colors evaluation/eval_kitti.py:73-88 (unit sphere; (rgb/255 - 0.5)/0.5)
  load_ply               pc_sam/ply_utils.py:5-29 / demo/utils.py:4-29 (ASCII, "x y z r g b" per vertex)
  read_ply_binary        evaluation/eval_kitti.py:117-241 (binary little/big-endian PLY, point clouds and triangle meshes)
  load_labelled_crop     evaluation/eval_kitti.py:336-347 (x y z R G B label crops, fixed scene rotation, single binary mask)
  filter_instance_masks  evaluation/eval_kitti.py:244-284 (masks with >= 25 points and < 90 % of the cloud, one sample per mask)
  apply_grouper_policy   evaluation/eval_kitti.py:352-362 (tokenizer size chosen from the point count)
  evaluate_clouds        evaluation/eval_kitti.py:343-390 (per-cloud IoU@click, running mean, per-object means)
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


# PLY scalar type names -> numpy codes (the table of evaluation/eval_kitti.py:23-42)
_PLY_TYPES = {"int8": "i1", "char": "i1", "uint8": "u1", "uchar": "u1", "int16": "i2", "short": "i2", "uint16": "u2", "ushort": "u2",
              "int32": "i4", "int": "i4", "uint32": "u4", "uint": "u4", "float32": "f4", "float": "f4", "float64": "f8", "double": "f8"}
_PLY_ENDIAN = {"binary_big_endian": ">", "binary_little_endian": "<"}


def read_ply_binary(path: str, triangular_mesh: bool = False):
    """Binary PLY -> numpy structured array with one field per vertex property (``data["x"]``, ``data["label"]`` ...); with
    ``triangular_mesh`` -> [vertex array, int32 faces [F, 3]].  ASCII files raise ValueError as the reference's reader does
    (use load_ply for those)."""
    with open(path, "rb") as f:
        if b"ply" not in f.readline():
            raise ValueError(f"{path}: does not start with 'ply'")
        fmt = f.readline().split()[1].decode()
        if fmt == "ascii":
            raise ValueError(f"{path}: the file is not binary")
        if fmt not in _PLY_ENDIAN:
            raise ValueError(f"{path}: unknown PLY format {fmt!r}")
        ext = _PLY_ENDIAN[fmt]
        props, n_vertex, n_face, element = [], None, None, None
        while True:
            line = f.readline()
            if line == b"":
                raise ValueError(f"{path}: no end_header")
            if b"end_header" in line:
                break
            tok = line.split()
            if tok[:1] == [b"element"]:
                element = tok[1].decode()
                if element == "vertex":
                    n_vertex = int(tok[2])
                elif element == "face":
                    n_face = int(tok[2])
            elif tok[:1] == [b"property"]:
                if element == "vertex":
                    props.append((tok[2].decode(), ext + _PLY_TYPES[tok[1].decode()]))
                elif element == "face" and tok[1:4] != [b"list", b"uchar", b"int"]:
                    raise ValueError(f"{path}: unsupported face property {line!r}")
        if n_vertex is None:
            raise ValueError(f"{path}: no 'element vertex' in the header")
        vertices = np.fromfile(f, dtype=props, count=n_vertex)
        if not triangular_mesh:
            return vertices
        faces = np.fromfile(f, dtype=[("k", ext + "u1"), ("v1", ext + "i4"), ("v2", ext + "i4"), ("v3", ext + "i4")], count=n_face or 0)
        return [vertices, np.vstack((faces["v1"], faces["v2"], faces["v3"])).T]


def write_ply_binary(path: str, fields: Dict[str, np.ndarray], big_endian: bool = False) -> None:
    """Writes equally long 1-D arrays as the vertex properties of a binary PLY (test fixtures / exporting predictions)."""
    ext = ">" if big_endian else "<"
    names = {v: k for k, v in _PLY_TYPES.items() if k in ("char", "uchar", "short", "ushort", "int", "uint", "float", "double")}
    n = len(next(iter(fields.values())))
    dtype = [(k, ext + np.dtype(v.dtype).str[1:]) for k, v in fields.items()]
    rec = np.empty(n, dtype=dtype)
    for k, v in fields.items():
        rec[k] = v
    with open(path, "wb") as f:
        f.write(b"ply\nformat " + (b"binary_big_endian" if big_endian else b"binary_little_endian") + b" 1.0\n")
        f.write(f"element vertex {n}\n".encode())
        for k, v in fields.items():
            f.write(f"property {names[np.dtype(v.dtype).str[1:]]} {k}\n".encode())
        f.write(b"end_header\n")
        rec.tofile(f)


def scene_rotation() -> np.ndarray:
    """Rotation.from_euler("xyz", [-90, 180, 0], degrees=True) of evaluation/eval_kitti.py:19 as a matrix (rows applied to points:
    p' = R p): extrinsic x by -90 degrees, then y by 180 degrees."""
    rx = np.array([[1.0, 0.0, 0.0], [0.0, 0.0, 1.0], [0.0, -1.0, 0.0]])      # -90 degrees about x
    ry = np.array([[-1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, -1.0]])     # 180 degrees about y
    return ry @ rx


def load_labelled_crop(path: str, rotate: bool = True) -> Dict[str, np.ndarray]:
    """One object crop of the reference's KITTI-360 evaluation (binary PLY with x y z R G B label): xyz float32 (rotated by the
    fixed scene rotation like eval_kitti.py:340), rgb float32 0..255, mask = the per-point 0/1 label."""
    pc = read_ply_binary(path)
    xyz = np.column_stack([pc["x"], pc["y"], pc["z"]]).astype(np.float32)
    if rotate:
        xyz = np.float32(xyz @ scene_rotation().T)
    rgb = np.column_stack([pc["R"], pc["G"], pc["B"]]).astype(np.float32)
    return dict(xyz=xyz, rgb=rgb, mask=pc["label"].astype(np.int32))


def crop_to_sample(crop: Dict[str, np.ndarray], device="cuda") -> Dict[str, torch.Tensor]:
    """transform_fn of evaluation/eval_kitti.py:90-115: normalise, one binary ground-truth mask [1, 1, N]."""
    return dict(
        coords=torch.tensor(normalize_points(crop["xyz"]), dtype=torch.float32, device=device)[None],
        features=torch.tensor(normalize_colors(crop["rgb"]), dtype=torch.float32, device=device)[None],
        gt_masks=torch.tensor(crop["mask"], dtype=torch.bool, device=device)[None, None],
    )


def filter_instance_masks(masks: np.ndarray, min_points: int = 25, max_fraction: float = 0.9) -> np.ndarray:
    """build_dataloader of evaluation/eval_kitti.py:244-284: keep the instance masks with at least `min_points` points and fewer
    than `max_fraction` of the cloud."""
    keep = [m for m in masks if m.sum() >= min_points and m.sum() < max_fraction * masks.shape[1]]
    return np.stack(keep) if keep else np.zeros((0, masks.shape[1]), dtype=bool)


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
def evaluate_clouds(model, samples: Iterable[Dict[str, torch.Tensor]], adapt_grouper: bool = True, names: Iterable[str] = None) -> Dict[str, np.ndarray]:
    """IoU after each click, averaged over the masks of a cloud, then over clouds (eval_kitti.py:343-380).  names (optional, one
    per sample): object names -- the per-object and object-mean IoU of eval_kitti.py:381-390 are reported too.
    Aggregation: `per_cloud` [clouds, iters] is the mean over a cloud's B*M masks, so every CLOUD weighs the same in `mean_iou_at_click`.
    The reference keeps the mask axis ([iters, 1, B*M] averaged over the singleton axis) and averages masks later; for the single-mask KITTI
    crops of eval_kitti.py both are the same number, for multi-mask samples (prepare_sample) with a varying mask count they differ --
    `per_mask` (list of [iters, B*M] arrays, one per cloud) is returned so that a caller can reproduce either.  A mask with an empty union
    (no ground truth, no prediction) has IoU NaN here as in the reference and propagates into its cloud's mean."""
    per_cloud: List[np.ndarray] = []
    per_mask: List[np.ndarray] = []
    samples = list(samples)
    for data in samples:
        if adapt_grouper:
            apply_grouper_policy(model, data["coords"].shape[1])
        outputs = model(**data, is_eval=True)
        gt = data["gt_masks"].flatten(0, 1)
        ious = [compute_iou(o["prompt_masks"], gt).float().cpu().numpy() for o in outputs]  # [iters][B*M]
        per_mask.append(np.array(ious))
        per_cloud.append(per_mask[-1].mean(axis=1))
    per_cloud_a = np.array(per_cloud)
    out = dict(per_cloud=per_cloud_a, mean_iou_at_click=per_cloud_a.mean(axis=0), per_mask=per_mask)
    if names is not None:
        names = list(names)
        objs = {n: per_cloud_a[[i for i, m in enumerate(names) if m == n]].mean(axis=0) for n in dict.fromkeys(names)}
        out["per_object"] = objs
        out["object_mean_iou_at_click"] = np.array(list(objs.values())).mean(axis=0)
    return out


def evaluate_crop_files(model, paths: Iterable[str], rotate: bool = True) -> Dict[str, np.ndarray]:
    """The reference's evaluation loop over object-crop PLY files (eval_kitti.py:329-390): object name = file name up to the first '_'."""
    import os
    paths = list(paths)
    samples = (crop_to_sample(load_labelled_crop(p, rotate)) for p in paths)
    return evaluate_clouds(model, samples, adapt_grouper=True, names=[os.path.basename(p).split("_")[0] for p in paths])
