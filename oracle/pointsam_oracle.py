"""
ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU (PyTorch fp32 + a small C library) restatement of the Point-SAM inference hot path
(encode + prompt decode), written functionally over a state dict keyed by the reference's parameter
names.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product path (``point_sam_amd/``) never does and fails loudly without its HIP library.

What pins it:
  * every pure-PyTorch module of the reference (pc_sam/model/{common,pc_encoder,prompt_encoder,
    mask_decoder,transformer,pc_sam}.py) IS importable in the build container once the absent
    third-party imports are stubbed; ``tests/golden/make_golden.py`` runs the reference's own classes on
    seeded inputs and commits their outputs; ``tests/test_oracle_golden.py`` checks this file against them.
  * PARITY UNPINNED (no source, no tests, no pinned version in the reference checkout) for the three
    third-party pieces: torkit3d ``sample_farthest_points`` (FPS), timm's Eva blocks, apex FusedLayerNorm.
    They are restated from their published algorithms; the spec chosen is documented at each function.

Distance modes: ``"reference"`` uses torch.cdist + topk exactly like common.py:51-55 (matmul-form
distances, what the published model saw); ``"exact"`` uses direct differences in fp32 with a defined
(distance, index) tie order -- the spec the HIP kernels implement bit-exactly.
"""
import ctypes
import math
import os
import subprocess
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build_c_oracle(force: bool = False) -> str:
    so = os.path.join(_HERE, "libtokenizer_oracle.so")
    src = os.path.join(_HERE, "tokenizer_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libtokenizer_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build_c_oracle())
        i64, p = ctypes.c_int64, ctypes.c_void_p
        _LIB.fps_f32.argtypes = [p, i64, i64, p, p]
        _LIB.knn_f32.argtypes = [p, p, i64, i64, i64, p, p]
        _LIB.three_nn_f32.argtypes = [p, p, i64, i64, ctypes.c_float, p, p]
        _LIB.border_farthest_f32.argtypes = [p, p, i64, p, p]
        for f in (_LIB.fps_f32, _LIB.knn_f32, _LIB.three_nn_f32, _LIB.border_farthest_f32):
            f.restype = ctypes.c_int
    return _LIB


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous().cpu()


# --------------------------------------------------------------------------------------------------
# Tokenizer: FPS, kNN, grouping                                      (pc_sam/model/common.py:59-123)
# --------------------------------------------------------------------------------------------------
def fps(xyz: torch.Tensor, num_samples: int) -> torch.Tensor:
    """[B,N,3] -> [B,G] int64.  Call site common.py:91; algorithm = torkit3d (absent; see header).
    Spec: start index 0, fp32 d = dx*dx+dy*dy+dz*dz (no FMA), running min, argmax with lowest index on ties."""
    xyz = _f32c(xyz)
    B, N, _ = xyz.shape
    out = torch.empty(B, num_samples, dtype=torch.int64)
    scratch = torch.empty(N, dtype=torch.float32)
    for b in range(B):
        rc = _lib().fps_f32(xyz[b].data_ptr(), N, num_samples, out[b].data_ptr(), scratch.data_ptr())
        if rc != 0:
            raise ValueError(f"fps_f32 failed rc={rc} (N={N}, G={num_samples})")
    return out


def fps_numpy(xyz: np.ndarray, num_samples: int) -> np.ndarray:
    """Independent pure-numpy statement of the same spec (small inputs; validates the C code)."""
    xyz = np.asarray(xyz, dtype=np.float32)
    N = xyz.shape[0]
    mind = np.full(N, np.inf, dtype=np.float32)
    idx = np.zeros(num_samples, dtype=np.int64)
    for j in range(1, num_samples):
        d = xyz - xyz[idx[j - 1]]
        d2 = d[:, 0] * d[:, 0]
        d2 = d2 + d[:, 1] * d[:, 1]
        d2 = d2 + d[:, 2] * d[:, 2]
        mind = np.minimum(mind, d2)
        idx[j] = int(np.argmax(mind))  # np.argmax returns the first maximum
    return idx


def batch_index_select(x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """out[b,j,:] = x[b, idx[b,j], :]  (torkit3d.nn.functional.batch_index_select, call site common.py:92)."""
    return torch.gather(x, 1, idx.unsqueeze(-1).expand(-1, -1, x.shape[-1]))


def knn(query: torch.Tensor, key: torch.Tensor, k: int, mode: str = "exact") -> Tuple[torch.Tensor, torch.Tensor]:
    """K nearest keys of each query: ([B,Q,K] distances (NOT squared), [B,Q,K] int64 indices).
    Reference: common.py:27-56."""
    if mode == "reference":
        distance = torch.cdist(query, key)  # common.py:51
        if k == 1:
            return torch.min(distance, dim=2, keepdim=True)
        return torch.topk(distance, k, dim=2, largest=False, sorted=False)  # common.py:55
    assert mode == "exact"
    q, kk = _f32c(query), _f32c(key)
    B, Q, _ = q.shape
    N = kk.shape[1]
    idx = torch.empty(B, Q, k, dtype=torch.int64)
    d2 = torch.empty(B, Q, k, dtype=torch.float32)
    for b in range(B):
        rc = _lib().knn_f32(q[b].data_ptr(), kk[b].data_ptr(), Q, N, k, idx[b].data_ptr(), d2[b].data_ptr())
        if rc != 0:
            raise ValueError(f"knn_f32 failed rc={rc}")
    return d2.sqrt(), idx


def knn_numpy(centers: np.ndarray, xyz: np.ndarray, k: int) -> np.ndarray:
    """Independent numpy statement of the exact-mode kNN spec (validates the C code)."""
    c = np.asarray(centers, np.float32)[:, None, :]
    x = np.asarray(xyz, np.float32)[None, :, :]
    d = c - x
    d2 = d[..., 0] * d[..., 0]
    d2 = d2 + d[..., 1] * d[..., 1]
    d2 = d2 + d[..., 2] * d[..., 2]
    return np.argsort(d2, axis=1, kind="stable")[:, :k].astype(np.int64)


def group_points(xyz, features, centers, knn_idx, radius=None, center_idx=None) -> torch.Tensor:
    """[B,G,K,3+C]: neighbour xyz relative to its center (divided by `radius` when given, common.py:107-108), then neighbour
    features (common.py:99-120); with center_idx [B,G] (`centralize_features`, common.py:116-118 / 183-186) also the neighbour
    features minus the features of the group's centre point: [B,G,K,3+2C]."""
    B, N, _ = xyz.shape
    G, K = knn_idx.shape[1:]
    flat = (knn_idx + torch.arange(B).view(B, 1, 1) * N).reshape(-1)
    nbr_xyz = xyz.reshape(-1, 3)[flat].reshape(B, G, K, 3) - centers.unsqueeze(2)
    if radius is not None:
        nbr_xyz = nbr_xyz / radius
    nbr_f = features.reshape(-1, features.shape[-1])[flat].reshape(B, G, K, -1)
    parts = [nbr_xyz, nbr_f]
    if center_idx is not None:
        parts.append(nbr_f - batch_index_select(features, center_idx).unsqueeze(2))
    return torch.cat(parts, dim=-1)


def knn_grouper(xyz, features, num_groups, group_size, mode="exact", radius=None, centralize_features=False) -> Dict[str, torch.Tensor]:
    """KNNGrouper.forward (common.py:73-123)."""
    fps_idx = fps(xyz, num_groups)
    centers = batch_index_select(xyz, fps_idx)
    _, knn_idx = knn(centers, xyz, group_size, mode)
    feats = group_points(xyz, features, centers, knn_idx, radius, fps_idx if centralize_features else None)
    return dict(features=feats, centers=centers, knn_idx=knn_idx, fps_idx=fps_idx)


# --------------------------------------------------------------------------------------------------
# Small dense helpers
# --------------------------------------------------------------------------------------------------
def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def _ln(sd, name, x, eps):
    w = sd[name + ".weight"]
    return F.layer_norm(x, (w.shape[0],), w, sd[name + ".bias"], eps)


def patch_encoder(sd, prefix: str, patches: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """Mini-PointNet over each group (common.py:477-506): [B,G,K,Cin] -> [B,G,Cout]."""
    x = _lin(sd, prefix + ".conv1.0", patches)
    x = F.gelu(_ln(sd, prefix + ".conv1.1", x, eps))
    x = _lin(sd, prefix + ".conv1.3", x)
    y = x.max(dim=-2, keepdim=True).values
    x = torch.cat([y.expand_as(x), x], dim=-1)
    x = _lin(sd, prefix + ".conv2.0", x)
    x = F.gelu(_ln(sd, prefix + ".conv2.1", x, eps))
    x = _lin(sd, prefix + ".conv2.3", x)
    return x.max(dim=-2).values


# --------------------------------------------------------------------------------------------------
# ViT blocks (timm Eva; absent third party -> our spec)
# --------------------------------------------------------------------------------------------------
def eva_block(sd, p: str, x: torch.Tensor, vit) -> torch.Tensor:
    """x += Attn(LN1(x)); x += MLP(LN2(x)).  No RoPE: the reference calls block(x) without ``rope``
    (pc_encoder.py:138-139).  eva02: q_proj(bias)/k_proj(no bias)/v_proj(bias), SwiGLU with inner LN.
    eva_giant: fused qkv with (q_bias, 0, v_bias), GELU MLP."""
    B, L, D = x.shape
    H, hd = vit.heads, vit.head_dim
    h = _ln(sd, p + ".norm1", x, vit.ln_eps)
    if vit.swiglu:
        q, k, v = _lin(sd, p + ".attn.q_proj", h), _lin(sd, p + ".attn.k_proj", h), _lin(sd, p + ".attn.v_proj", h)
    else:
        bias = torch.cat([sd[p + ".attn.q_bias"], torch.zeros(D), sd[p + ".attn.v_bias"]])
        q, k, v = F.linear(h, sd[p + ".attn.qkv.weight"], bias).split(D, dim=-1)
    q, k, v = (t.reshape(B, L, H, hd).transpose(1, 2) for t in (q, k, v))
    attn = torch.softmax((q @ k.transpose(-1, -2)) * (hd ** -0.5), dim=-1)
    a = (attn @ v).transpose(1, 2).reshape(B, L, D)
    x = x + _lin(sd, p + ".attn.proj", a)
    h = _ln(sd, p + ".norm2", x, vit.ln_eps)
    if vit.swiglu:
        g = F.silu(_lin(sd, p + ".mlp.fc1_g", h)) * _lin(sd, p + ".mlp.fc1_x", h)
        m = _lin(sd, p + ".mlp.fc2", _ln(sd, p + ".mlp.norm", g, vit.ln_eps))
    else:
        m = _lin(sd, p + ".mlp.fc2", F.gelu(_lin(sd, p + ".mlp.fc1", h)))
    return x + m


def pc_encoder(sd, cfg, coords, features, mode="exact"):
    """PointCloudEncoder.forward (pc_encoder.py:118-145) -> (embeddings [B,G,E], patches dict)."""
    patches = knn_grouper(coords, features, cfg.num_groups, cfg.group_size, mode, getattr(cfg, "radius", None), getattr(cfg, "centralize_features", False))
    emb = patch_encoder(sd, "pc_encoder.patch_embed.patch_encoder", patches["features"], cfg.ln_eps)
    patches["embeddings"] = emb
    x = _lin(sd, "pc_encoder.patch_proj", emb)
    pos = _lin(sd, "pc_encoder.pos_embed.2", F.gelu(_lin(sd, "pc_encoder.pos_embed.0", patches["centers"])))
    x = x + pos
    for i in range(cfg.vit.depth):
        x = eva_block(sd, f"pc_encoder.transformer.blocks.{i}", x, cfg.vit)
    x = _ln(sd, "pc_encoder.transformer.fc_norm", x, cfg.vit.ln_eps)
    return _lin(sd, "pc_encoder.out_proj", x), patches


# --------------------------------------------------------------------------------------------------
# Prompt encoders                                               (pc_sam/model/prompt_encoder.py:13-133)
# --------------------------------------------------------------------------------------------------
def pe_encoding(sd, coords: torch.Tensor) -> torch.Tensor:
    """PositionEmbeddingRandom.forward (prompt_encoder.py:27-48): range check then [sin, cos](2*pi*x@G)."""
    if (coords < -1 - 1e-6).any() or (coords > 1 + 1e-6).any():
        raise ValueError("Input coordinates must be normalized to [-1, 1].")
    c = coords @ sd["point_encoder.pe_layer.positional_encoding_gaussian_matrix"]
    c = 2 * np.pi * c
    return torch.cat([torch.sin(c), torch.cos(c)], dim=-1)


def point_encoder(sd, points: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """PointEncoder.forward (prompt_encoder.py:63-77)."""
    assert points.shape[:-1] == labels.shape
    e = pe_encoding(sd, points)
    lab = labels.to(torch.int64)
    e = e + (lab == 0).unsqueeze(-1) * sd["point_encoder.point_embeddings.0.weight"]
    e = e + (lab == 1).unsqueeze(-1) * sd["point_encoder.point_embeddings.1.weight"]
    return e


def mask_encoder(sd, cfg, masks: Optional[torch.Tensor], coords, centers, knn_idx, center_idx=None) -> torch.Tensor:
    """MaskEncoder.forward (prompt_encoder.py:97-133) + group_with_centers_and_knn (common.py:126-187).  The mask encoder has its OWN
    radius / centralize_features options (prompt_encoder.py:78-93); center_idx [B,G] is passed by PointCloudSAM.forward only."""
    if masks is None:
        return sd["mask_encoder.no_mask_embed.weight"].reshape(1, 1, -1).expand(centers.shape[0], centers.shape[1], -1)
    B = coords.shape[0]
    rep = masks.shape[0] // B
    radius = cfg.mask_encoder_radius if hasattr(cfg, "mask_encoder_radius") else getattr(cfg, "radius", None)
    rel = group_points(coords, coords, centers, knn_idx, radius)[..., :3]  # [B,G,K,3]
    rel = rel.repeat_interleave(rep, dim=0)
    kidx = knn_idx.repeat_interleave(rep, dim=0)
    logit = torch.gather(masks, 1, kidx.reshape(masks.shape[0], -1)).reshape(*kidx.shape, 1)
    parts = [rel, logit]
    if getattr(cfg, "mask_centralize_features", False):
        cidx = center_idx.repeat_interleave(rep, dim=0)
        parts.append(logit - torch.gather(masks, 1, cidx).reshape(masks.shape[0], -1, 1, 1))
    return patch_encoder(sd, "mask_encoder.patch_encoder", torch.cat(parts, dim=-1), cfg.ln_eps)


# --------------------------------------------------------------------------------------------------
# Two-way transformer + mask decoder      (pc_sam/model/transformer.py, pc_sam/model/mask_decoder.py)
# --------------------------------------------------------------------------------------------------
def attention(sd, p: str, q, k, v, heads: int) -> torch.Tensor:
    """Attention.forward (transformer.py:214-236)."""
    q, k, v = _lin(sd, p + ".q_proj", q), _lin(sd, p + ".k_proj", k), _lin(sd, p + ".v_proj", v)

    def sep(t):
        b, n, c = t.shape
        return t.reshape(b, n, heads, c // heads).transpose(1, 2)

    q, k, v = sep(q), sep(k), sep(v)
    a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(q.shape[-1]), dim=-1)
    o = (a @ v).transpose(1, 2)
    return _lin(sd, p + ".out_proj", o.reshape(o.shape[0], o.shape[1], -1))


def two_way_transformer(sd, cfg, src, pos_src, tokens):
    """TwoWayTransformer.forward (transformer.py:61-100) with TwoWayAttentionBlock.forward (:144-176)."""
    P = "mask_decoder.transformer"
    H, eps = cfg.dec_heads, cfg.ln_eps
    queries, keys = tokens, src
    for i in range(cfg.dec_depth):
        L = f"{P}.layers.{i}"
        if i == 0:  # skip_first_layer_pe: output REPLACES the queries (transformer.py:149-150)
            queries = attention(sd, L + ".self_attn", queries, queries, queries, H)
        else:
            q = queries + tokens
            queries = queries + attention(sd, L + ".self_attn", q, q, queries, H)
        queries = _ln(sd, L + ".norm1", queries, eps)
        q, k = queries + tokens, keys + pos_src
        queries = _ln(sd, L + ".norm2", queries + attention(sd, L + ".cross_attn_token_to_image", q, k, keys, H), eps)
        m = _lin(sd, L + ".mlp.lin2", F.relu(_lin(sd, L + ".mlp.lin1", queries)))
        queries = _ln(sd, L + ".norm3", queries + m, eps)
        q, k = queries + tokens, keys + pos_src
        keys = _ln(sd, L + ".norm4", keys + attention(sd, L + ".cross_attn_image_to_token", k, q, queries, H), eps)
    q, k = queries + tokens, keys + pos_src
    queries = queries + attention(sd, P + ".final_attn_token_to_image", q, k, keys, H)
    return _ln(sd, P + ".norm_final_attn", queries, eps), keys


def interp_weights(query, key, mode="exact", eps=1e-8):
    """compute_interp_weights (common.py:238-255): 3-NN indices + normalised 1/clamp(d^2, eps)."""
    if mode == "reference":
        dist, idx = knn(query, key, 3, "reference")
        inv = 1.0 / torch.clamp(dist.square(), min=eps)
        return idx, inv / inv.sum(dim=2, keepdim=True)
    q, k = _f32c(query), _f32c(key)
    B, N, _ = q.shape
    idx = torch.empty(B, N, 3, dtype=torch.int64)
    w = torch.empty(B, N, 3, dtype=torch.float32)
    for b in range(B):
        rc = _lib().three_nn_f32(q[b].data_ptr(), k[b].data_ptr(), N, k.shape[1], eps, idx[b].data_ptr(), w[b].data_ptr())
        if rc != 0:
            raise ValueError(f"three_nn_f32 failed rc={rc}")
    return idx, w


def interpolate(x, index, weight):
    """interpolate_features (common.py:258-274): sum_k w_k * x[idx_k]."""
    B, Nq, K = index.shape
    flat = (index + torch.arange(B).view(B, 1, 1) * x.shape[1]).flatten()
    return (x.flatten(0, 1)[flat].reshape(B, Nq, K, -1) * weight.unsqueeze(-1)).sum(-2)


def _mlp3(sd, p, x):
    """MLP.forward with 3 layers, ReLU between (mask_decoder.py:189-211)."""
    x = F.relu(_lin(sd, p + ".layers.0", x))
    x = F.relu(_lin(sd, p + ".layers.1", x))
    return _lin(sd, p + ".layers.2", x)


@dataclass
class Aux:
    """AuxInputs (mask_decoder.py:12-18)."""
    coords: torch.Tensor
    centers: torch.Tensor
    interp_index: Optional[torch.Tensor] = None
    interp_weight: Optional[torch.Tensor] = None


def mask_decoder(sd, cfg, pc_emb, pc_pe, sparse, dense, aux: Aux, multimask_output: bool, mode="exact"):
    """MaskDecoder.forward/predict_masks (mask_decoder.py:65-184) -> (masks [BM,C,N], iou [BM,C])."""
    BM = sparse.shape[0]
    rep = BM // pc_emb.shape[0]
    out_tok = torch.cat([sd["mask_decoder.iou_token.weight"], sd["mask_decoder.mask_tokens.weight"]], 0)
    tokens = torch.cat([out_tok.unsqueeze(0).expand(BM, -1, -1), sparse], dim=1)
    src = pc_emb.repeat_interleave(rep, 0) + dense
    pos = pc_pe.repeat_interleave(rep, 0)
    hs, src = two_way_transformer(sd, cfg, src, pos, tokens)
    iou_tok, mask_tok = hs[:, 0], hs[:, 1:1 + cfg.num_mask_tokens]
    if aux.interp_index is None:
        aux.interp_index, aux.interp_weight = interp_weights(aux.coords, aux.centers, mode)
    ii = aux.interp_index.repeat_interleave(BM // aux.interp_index.shape[0], 0)
    iw = aux.interp_weight.repeat_interleave(BM // aux.interp_weight.shape[0], 0)
    up = interpolate(src, ii, iw)
    up = _lin(sd, "mask_decoder.output_upscaling.0", up)
    up = F.gelu(_ln(sd, "mask_decoder.output_upscaling.1", up, cfg.ln_eps))
    up = F.gelu(_lin(sd, "mask_decoder.output_upscaling.3", up))
    sel = list(range(cfg.num_mask_tokens))[1:] if multimask_output else [0]
    hyper = torch.stack([_mlp3(sd, f"mask_decoder.output_hypernetworks_mlps.{i}", mask_tok[:, i]) for i in sel], 1)
    masks = hyper @ up.transpose(-1, -2)
    iou = _mlp3(sd, "mask_decoder.iou_prediction_head", iou_tok)[:, sel]
    return masks, iou


# --------------------------------------------------------------------------------------------------
# Model assembly                                                    (pc_sam/model/pc_sam.py:37-88)
# --------------------------------------------------------------------------------------------------
@torch.no_grad()
def predict_masks(sd, cfg, coords, features, prompt_coords, prompt_labels, prompt_masks=None,
                  multimask_output=True, mode="exact", return_intermediates=False):
    """PointCloudSAM.predict_masks (pc_sam.py:37-88)."""
    pc_emb, patches = pc_encoder(sd, cfg, coords, features, mode)
    centers, knn_idx = patches["centers"], patches["knn_idx"]
    aux = Aux(coords=coords, centers=centers)
    pc_pe = pe_encoding(sd, centers)
    sparse = point_encoder(sd, prompt_coords, prompt_labels)
    dense = mask_encoder(sd, cfg, prompt_masks, coords, centers, knn_idx)
    dense = dense.repeat_interleave(sparse.shape[0] // dense.shape[0], 0)
    masks, iou = mask_decoder(sd, cfg, pc_emb, pc_pe, sparse, dense, aux, multimask_output, mode)
    if return_intermediates:
        return masks, iou, dict(pc_embeddings=pc_emb, patches=patches, pc_pe=pc_pe, sparse=sparse, dense=dense, aux=aux)
    return masks, iou


@torch.no_grad()
def click_loop(sd, cfg, coords, features, clicks, labels, mode="exact"):
    """Decoder-only multi-click loop with the encoder cached: the data flow of PointCloudSAM.forward
    (pc_sam.py:112-194) with the clicks supplied by the caller instead of sampled from ground truth
    (BASELINE config #5).  clicks [BM, T, 3], labels [BM, T]; returns list of (masks, iou) per click."""
    pc_emb, patches = pc_encoder(sd, cfg, coords, features, mode)
    centers, knn_idx = patches["centers"], patches["knn_idx"]
    aux = Aux(coords=coords, centers=centers)
    pc_pe = pe_encoding(sd, centers)
    outs, prompt_masks = [], None
    for t in range(clicks.shape[1]):
        sparse = point_encoder(sd, clicks[:, : t + 1], labels[:, : t + 1])
        dense = mask_encoder(sd, cfg, prompt_masks, coords, centers, knn_idx)
        dense = dense.repeat_interleave(sparse.shape[0] // dense.shape[0], 0)
        masks, iou = mask_decoder(sd, cfg, pc_emb, pc_pe, sparse, dense, aux, t == 0, mode)
        if t == 0:  # pc_sam.py:176-180
            prompt_masks = torch.gather(masks, 1, iou.argmax(1).view(-1, 1, 1).expand(-1, 1, masks.shape[-1]))[:, 0]
        else:       # pc_sam.py:181-183
            prompt_masks = masks[:, 0]
        outs.append((masks, iou))
    return outs


# --------------------------------------------------------------------------------------------------
# Prompt simulation for the evaluation protocol          (pc_sam/model/common.py:287-316,368-474; pc_sam.py:139-194)
# --------------------------------------------------------------------------------------------------
def border_farthest(coords: torch.Tensor, region: torch.Tensor):
    """sample_furthest_points_from_border (common.py:443-474) for one cloud: (index, squared distance) of the region point
    farthest from the region's complement, or (-1, -1.0) when either is empty."""
    xyz = _f32c(coords)
    reg = region.to(torch.uint8).contiguous()
    idx = ctypes.c_int64(-1)
    dist = ctypes.c_float(-1.0)
    _lib().border_farthest_f32(xyz.data_ptr(), reg.data_ptr(), xyz.shape[0], ctypes.byref(idx), ctypes.byref(dist))
    return int(idx.value), float(dist.value)


def sample_eval_prompts(points, gt_masks, pred_logits):
    """sample_prompts_adapter(..., is_eval=True) (common.py:287-316) -> sample_fixed_points (common.py:368-441).
    points [B,N,3], gt_masks [B,M,N] bool, pred_logits None or [B*M,N] -> (coords [B*M,1,3], labels [B*M,1] bool)."""
    B, M, N = gt_masks.shape
    coords, labels = [], []
    for i in range(B):
        for j in range(M):
            gt = gt_masks[i, j]
            if pred_logits is None:  # from_error_region=True with fn = gt, fp = 0
                n, _ = border_farthest(points[i], gt)
            else:
                pred = pred_logits.reshape(B, M, N)[i, j] > 0
                pn, pd = border_farthest(points[i], gt & ~pred)   # false negatives
                nn_, nd = border_farthest(points[i], ~gt & pred)   # false positives
                if pd > nd:
                    n = pn
                elif nd == -1:
                    n, _ = border_farthest(points[i], gt)
                else:
                    n = nn_
            if n < 0:
                raise ValueError("empty region: the reference would fail in torch.stack on None (common.py:439)")
            coords.append(points[i][n][None])
            labels.append(gt[n][None])
    return torch.stack(coords), torch.stack(labels)


@torch.no_grad()
def forward_eval(sd, cfg, coords, features, gt_masks, prompt_iters=None, mode="exact"):
    """PointCloudSAM.forward(coords, features, gt_masks, is_eval=True) in eval mode (pc_sam.py:90-196)."""
    B, M, N = gt_masks.shape
    iters = cfg.prompt_iters if prompt_iters is None else prompt_iters
    pc_emb, patches = pc_encoder(sd, cfg, coords, features, mode)
    centers, knn_idx = patches["centers"], patches["knn_idx"]
    aux = Aux(coords=coords, centers=centers)
    pc_pe = pe_encoding(sd, centers)
    prompt_coords = coords.new_empty((B * M, 0, 3))
    prompt_labels = gt_masks.new_empty((B * M, 0))
    prompt_masks, outputs = None, []
    for i in range(iters):
        nc, nl = sample_eval_prompts(coords, gt_masks, prompt_masks)
        prompt_coords = torch.cat([prompt_coords, nc], dim=1)
        prompt_labels = torch.cat([prompt_labels, nl], dim=1)
        sparse = point_encoder(sd, prompt_coords, prompt_labels)
        dense = mask_encoder(sd, cfg, prompt_masks, coords, centers, knn_idx, patches["fps_idx"])   # pc_sam.py:151-157
        dense = dense.repeat_interleave(sparse.shape[0] // dense.shape[0], 0)
        masks, iou = mask_decoder(sd, cfg, pc_emb, pc_pe, sparse, dense, aux, i == 0, mode)
        if i == 0:
            best = iou.argmax(1)
            prompt_masks = torch.gather(masks, 1, best.view(-1, 1, 1).expand(-1, 1, N))[:, 0]
        else:
            best = 0
            prompt_masks = masks[:, 0]
        outputs.append(dict(prompt_coords=prompt_coords, prompt_labels=prompt_labels, masks=masks, iou_preds=iou,
                            max_iou_pred_ind=best, prompt_masks=prompt_masks))
    return outputs


# --------------------------------------------------------------------------------------------------
# Synthetic inputs -- shared by tests, smoke and bench
# --------------------------------------------------------------------------------------------------
def synthetic_batch(B: int, N: int, seed: int = 42, num_prompts: int = 1):
    """Seeded clouds normalised like evaluation/inference.py:58-59 (per cloud), rgb in [-1,1],
    one positive point prompt taken from the cloud itself."""
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(B, N, 3, generator=g) * 2 - 1
    xyz = xyz - xyz.mean(dim=1, keepdim=True)
    xyz = xyz / xyz.norm(dim=2).max(dim=1).values.view(B, 1, 1)
    rgb = torch.rand(B, N, 3, generator=g) * 2 - 1
    pidx = torch.randint(0, N, (B, num_prompts), generator=g)
    prompt = torch.gather(xyz, 1, pidx.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    labels = torch.ones(B, num_prompts, dtype=torch.int64)
    return xyz.contiguous(), rgb.contiguous(), prompt, labels
