"""CPU oracle for the reference's two model VARIANTS -- TEST INFRASTRUCTURE ONLY (imported by tests/ and tests/golden/make_golden.py;
never by the product path).

  voronoi (configs/model/voronoi.yaml): PointCloudSAMNN, pc_sam/model/pc_sam.py:199-374 -- every point belongs to its nearest FPS centre
      NNGrouper                common.py:190-212
      PatchEmbedNN + Block     pc_encoder.py:147-198
      MaskEncoderNN + ResMlp   prompt_encoder.py:186-211, 255-300
      MaskDecoder              (the base one, oracle/pointsam_oracle.py)
  hier (configs/model/hier.yaml): PointCloudSAMHier, pc_sam.py:377-496 -- two kNN levels
      PatchEmbedHier           pc_encoder.py:201-239 (KNNGrouper twice, the second one without FPS: common.py:92-96)
      MaskEncoderHier          prompt_encoder.py:136-183
      MaskDecoderHier          mask_decoder.py:214-370 (two-stage upscaling, level-1 embeddings concatenated)

Pinned by tests/golden/ref_tiny_voronoi.npz / ref_tiny_hier.npz: outputs of the reference's OWN classes on seeded inputs and weights
(tests/golden/make_golden.py variants).  Notes on the reference as found:
  * PointCloudSAMNN.predict_masks (pc_sam.py:232-262) reads patches["knn_idx"], a key NNGrouper does not return, and passes the mask
    encoder's arguments in another order than MaskEncoderNN.forward takes: it cannot run.  Its forward (pc_sam.py:264-374) is consistent
    (mask_encoder(prompt_masks, nn_idx, centers, coords)); `decode` below is that per-iteration call.
  * PointCloudSAMHier only defines forward; its click sampler (sample_prompts, common.py:319-365) draws random points, so the golden run
    records the decoder on given prompts instead of the loop.
"""
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from . import pointsam_oracle as O


# ---------------------------------------------------------------------------------------------- shared trunk
def encoder_trunk(sd, cfg, emb, centers):
    """PointCloudEncoder.forward after the patch embedding (pc_encoder.py:128-145)."""
    x = O._lin(sd, "pc_encoder.patch_proj", emb)
    x = x + O._lin(sd, "pc_encoder.pos_embed.2", F.gelu(O._lin(sd, "pc_encoder.pos_embed.0", centers)))
    for i in range(cfg.vit.depth):
        x = O.eva_block(sd, f"pc_encoder.transformer.blocks.{i}", x, cfg.vit)
    x = O._ln(sd, "pc_encoder.transformer.fc_norm", x, cfg.vit.ln_eps)
    return O._lin(sd, "pc_encoder.out_proj", x)


# ---------------------------------------------------------------------------------------------- voronoi
def nn_offsets(xyz, centers, nn_idx):
    """xyz - centre of its cell, and the offset's L2 norm (common.py:205-208)."""
    nbr = xyz - O.batch_index_select(centers, nn_idx)
    return nbr, torch.linalg.norm(nbr, dim=-1, keepdim=True, ord=2)


def nn_grouper(xyz, features, num_groups, mode="exact") -> Dict[str, torch.Tensor]:
    """NNGrouper.forward (common.py:197-212): features = [unit offset to the nearest centre (3), its length (1), point features]."""
    fps_idx = O.fps(xyz, num_groups)
    centers = O.batch_index_select(xyz, fps_idx)
    _, nn_idx = O.knn(xyz, centers, 1, mode)
    nn_idx = nn_idx.squeeze(-1)
    nbr, dist = nn_offsets(xyz, centers, nn_idx)
    nbr = nbr / torch.clamp(dist, min=1e-8)
    return dict(features=torch.cat([nbr, dist, features], dim=-1), centers=centers, nn_idx=nn_idx, fps_idx=fps_idx)


def nn_block(sd, p, x, eps):
    """Block.forward (pc_encoder.py:147-162): x + Linear(LayerNorm(GELU(Linear(LayerNorm(x)))))."""
    h = O._ln(sd, p + ".norm", x, eps)
    h = F.gelu(O._lin(sd, p + ".mlp.0", h))
    h = O._ln(sd, p + ".mlp.2", h, eps)
    return x + O._lin(sd, p + ".mlp.3", h)


def scatter_amax(x, idx, num_groups, include_self_zero: bool):
    """y[b, g, :] = max over the points n of cloud b with idx[b, n] == g of x[b, n, :].  include_self_zero: the destination's zeros take
    part in the maximum (torch.scatter_reduce(zeros, ..., 'amax'), prompt_encoder.py:291-297); otherwise only contributions do and a cell
    without points stays 0 (scatter_reduce_(..., include_self=False), pc_encoder.py:190-193)."""
    y = x.new_zeros(x.shape[0], num_groups, x.shape[-1])
    return y.scatter_reduce(1, idx.unsqueeze(-1).expand_as(x), x, "amax", include_self=include_self_zero)


def patch_embed_nn(sd, cfg, coords, features, mode="exact"):
    """PatchEmbedNN.forward (pc_encoder.py:181-198)."""
    P = "pc_encoder.patch_embed"
    patches = nn_grouper(coords, features, cfg.num_groups, mode)
    x = O._lin(sd, P + ".in_proj", patches["features"])
    for i in range(3):
        x = nn_block(sd, f"{P}.blocks1.{i}", x, cfg.ln_eps)
    patches["point_features"] = x
    y = scatter_amax(x, patches["nn_idx"], cfg.num_groups, include_self_zero=False)
    patches["pooled"] = y
    for i in range(3):
        y = nn_block(sd, f"{P}.blocks2.{i}", y, cfg.ln_eps)
    y = O._ln(sd, P + ".norm", y, cfg.ln_eps)
    patches["embeddings"] = O._lin(sd, P + ".out_proj", y)
    return patches


def res_mlp(sd, p, x, eps):
    """ResMlp.forward (prompt_encoder.py:186-211): Linear, LayerNorm, GELU, 3 x (GELU(LayerNorm(Linear(x))) + x), Linear."""
    x = F.gelu(O._ln(sd, p + ".mlp.1", O._lin(sd, p + ".mlp.0", x), eps))
    for i in (3, 4, 5):
        x = F.gelu(O._ln(sd, f"{p}.mlp.{i}.mlp.1", O._lin(sd, f"{p}.mlp.{i}.mlp.0", x), eps)) + x
    return O._lin(sd, p + ".mlp.6", x)


def mask_encoder_nn(sd, cfg, masks: Optional[torch.Tensor], nn_idx, centers, xyz):
    """MaskEncoderNN.forward (prompt_encoder.py:262-300): per point [logit, offset to its cell centre (NOT normalised), its length] ->
    Linear(5, 1024) -> max per cell (zeros included) -> ResMlp.  masks [B*M, N] | None.
    AS FOUND: the scatter index is the per-cloud cell index with NO batch offset (prompt_encoder.py:286-297 scatters rows of all B*M mask
    sets into rows [0, G) of the [B*M*G, 1024] buffer), so with more than one mask set the first one's cells receive the maximum over ALL
    sets and the others stay zero.  Restated as it is -- results must equal the reference's; for B*M == 1 it is the intended pooling."""
    if masks is None:
        return sd["mask_encoder.no_mask_embed.weight"].reshape(1, 1, -1).expand(centers.shape[0], centers.shape[1], -1)
    rep = masks.shape[0] // xyz.shape[0]
    nbr, dist = nn_offsets(xyz, centers, nn_idx)
    pg = torch.cat([masks.unsqueeze(-1), nbr.repeat_interleave(rep, 0), dist.repeat_interleave(rep, 0)], dim=-1)
    feat = O._lin(sd, "mask_encoder.first_nn", pg)
    BM, N, C = feat.shape
    G = centers.shape[1]
    flat_idx = nn_idx.repeat_interleave(rep, 0).reshape(-1, 1).expand(-1, C)        # values in [0, G): no batch offset (see above)
    agg = torch.zeros(BM * G, C).scatter_reduce(0, flat_idx, feat.reshape(-1, C), "amax")
    return res_mlp(sd, "mask_encoder.second_nn", agg, cfg.ln_eps).view(BM, G, -1)


# ---------------------------------------------------------------------------------------------- hier
def _patch_encoder_h(sd, prefix, patches, eps):
    return O.patch_encoder(sd, prefix, patches, eps)


def hier_groups(coords, cfg, mode="exact"):
    """The two KNNGrouper levels of PatchEmbedHier (pc_encoder.py:231-237): level 1 by FPS over the cloud; level 2 takes the FIRST
    num_patches[1] level-1 centres (use_fps=False, common.py:92-96: a prefix of an FPS order is an FPS sample) and their kNN among the
    level-1 centres."""
    G1, G2 = cfg.hier_groups
    K1, K2 = cfg.hier_sizes
    fps_idx = O.fps(coords, G1)
    centers1 = O.batch_index_select(coords, fps_idx)
    _, knn1 = O.knn(centers1, coords, K1, mode)
    centers2 = centers1[:, :G2].contiguous()
    _, knn2 = O.knn(centers2, centers1, K2, mode)
    return dict(fps_idx=fps_idx, centers1=centers1, knn1=knn1, centers2=centers2, knn2=knn2)


def patch_embed_hier(sd, cfg, coords, features, mode="exact"):
    """PatchEmbedHier.forward (pc_encoder.py:231-239) -> [patches1, patches2]."""
    g = hier_groups(coords, cfg, mode)
    r1, r2 = cfg.hier_radius if cfg.hier_radius else (None, None)
    f1 = O.group_points(coords, features, g["centers1"], g["knn1"], r1)
    x1 = _patch_encoder_h(sd, "pc_encoder.patch_embed.patch_encoder1", f1, cfg.ln_eps)
    f2 = O.group_points(g["centers1"], x1, g["centers2"], g["knn2"], r2)
    x2 = _patch_encoder_h(sd, "pc_encoder.patch_embed.patch_encoder2", f2, cfg.ln_eps)
    return [dict(centers=g["centers1"], knn_idx=g["knn1"], fps_idx=g["fps_idx"], embeddings=x1, features=f1),
            dict(centers=g["centers2"], knn_idx=g["knn2"], embeddings=x2, features=f2)]


def mask_encoder_hier(sd, cfg, masks, coords, centers1, knn1, centers2, knn2):
    """MaskEncoderHier.forward (prompt_encoder.py:152-183) -> the level-2 embedding [B*M, G2, E] (pc_sam.py:449-457 keeps the last one)."""
    if masks is None:
        return sd["mask_encoder.no_mask_embed.weight"].reshape(1, 1, -1).expand(centers2.shape[0], centers2.shape[1], -1)
    B = coords.shape[0]
    rep = masks.shape[0] // B
    r1, r2 = cfg.hier_radius if cfg.hier_radius else (None, None)
    rel1 = O.group_points(coords, coords, centers1, knn1, r1)[..., :3].repeat_interleave(rep, 0)
    k1 = knn1.repeat_interleave(rep, 0)
    logit = torch.gather(masks, 1, k1.reshape(masks.shape[0], -1)).reshape(*k1.shape, 1)
    x1 = _patch_encoder_h(sd, "mask_encoder.patch_encoder1", torch.cat([rel1, logit], -1), cfg.ln_eps)          # [BM, G1, 128]
    c1, c2, k2 = centers1.repeat_interleave(rep, 0), centers2.repeat_interleave(rep, 0), knn2.repeat_interleave(rep, 0)
    f2 = O.group_points(c1, x1, c2, k2, r2)
    return _patch_encoder_h(sd, "mask_encoder.patch_encoder2", f2, cfg.ln_eps)


def mask_decoder_hier(sd, cfg, pc_emb, pc_pe, sparse, dense, coords, patches, multimask_output, mode="exact", cache=None):
    """MaskDecoderHier.predict_masks (mask_decoder.py:289-352): transformer, then G2 -> G1 interpolation + level-1 embeddings ->
    output_upscaling2, G1 -> N interpolation -> output_upscaling1, hyper-network products."""
    BM = sparse.shape[0]
    rep = BM // pc_emb.shape[0]
    out_tok = torch.cat([sd["mask_decoder.iou_token.weight"], sd["mask_decoder.mask_tokens.weight"]], 0)
    tokens = torch.cat([out_tok.unsqueeze(0).expand(BM, -1, -1), sparse], dim=1)
    src = pc_emb.repeat_interleave(rep, 0) + dense
    hs, src = O.two_way_transformer(sd, cfg, src, pc_pe.repeat_interleave(rep, 0), tokens)
    iou_tok, mask_tok = hs[:, 0], hs[:, 1:1 + cfg.num_mask_tokens]
    cache = cache if cache is not None else {}
    if "i2" not in cache:
        cache["i2"] = O.interp_weights(patches[0]["centers"], patches[1]["centers"], mode)      # aux_inputs2 (pc_sam.py:408-412)
        cache["i1"] = O.interp_weights(coords, patches[0]["centers"], mode)                      # aux_inputs1 (pc_sam.py:405-407)
    (ii2, iw2), (ii1, iw1) = cache["i2"], cache["i1"]
    up = O.interpolate(src, ii2.repeat_interleave(rep, 0), iw2.repeat_interleave(rep, 0))
    up = torch.cat([up, patches[0]["embeddings"].repeat_interleave(rep, 0)], dim=-1)
    up = O._lin(sd, "mask_decoder.output_upscaling2.0", up)
    up = F.gelu(O._ln(sd, "mask_decoder.output_upscaling2.1", up, cfg.ln_eps))
    up = O._lin(sd, "mask_decoder.output_upscaling2.3", up)
    up = O.interpolate(up, ii1.repeat_interleave(rep, 0), iw1.repeat_interleave(rep, 0))
    up = O._lin(sd, "mask_decoder.output_upscaling1.0", up)
    up = F.gelu(O._ln(sd, "mask_decoder.output_upscaling1.1", up, cfg.ln_eps))
    up = F.gelu(O._lin(sd, "mask_decoder.output_upscaling1.3", up))
    sel = list(range(cfg.num_mask_tokens))[1:] if multimask_output else [0]
    hyper = torch.stack([O._mlp3(sd, f"mask_decoder.output_hypernetworks_mlps.{i}", mask_tok[:, i]) for i in sel], 1)
    masks = hyper @ up.transpose(-1, -2)
    iou = O._mlp3(sd, "mask_decoder.iou_prediction_head", iou_tok)[:, sel]
    return masks, iou


# ---------------------------------------------------------------------------------------------- assembly
@torch.no_grad()
def encode(sd, cfg, coords, features, mode="exact"):
    """pc_encoder(coords, features) of the variant (pc_encoder.py:118-145) + pe_layer(centers)."""
    if cfg.variant == "voronoi":
        patches = patch_embed_nn(sd, cfg, coords, features, mode)
        emb, centers = patches["embeddings"], patches["centers"]
    elif cfg.variant == "hier":
        patches = patch_embed_hier(sd, cfg, coords, features, mode)
        emb, centers = patches[-1]["embeddings"], patches[-1]["centers"]
    else:
        raise ValueError(cfg.variant)
    pc_emb = encoder_trunk(sd, cfg, emb, centers)
    return dict(pc_embeddings=pc_emb, pc_pe=O.pe_encoding(sd, centers), patches=patches, coords=coords, cache={})


@torch.no_grad()
def decode(sd, cfg, st, prompt_coords, prompt_labels, prompt_masks=None, multimask_output=True, mode="exact", return_dense=False):
    """One iteration of the variant's forward loop (pc_sam.py:326-352 / :437-471): prompt encoders + mask decoder on cached encoder outputs."""
    sparse = O.point_encoder(sd, prompt_coords, prompt_labels)
    coords, patches = st["coords"], st["patches"]
    if cfg.variant == "voronoi":
        dense = mask_encoder_nn(sd, cfg, prompt_masks, patches["nn_idx"], patches["centers"], coords)
        dense = dense.repeat_interleave(sparse.shape[0] // dense.shape[0], 0)
        aux = st["cache"].setdefault("aux", O.Aux(coords=coords, centers=patches["centers"]))
        masks, iou = O.mask_decoder(sd, cfg, st["pc_embeddings"], st["pc_pe"], sparse, dense, aux, multimask_output, mode)
    else:
        dense = mask_encoder_hier(sd, cfg, prompt_masks, coords, patches[0]["centers"], patches[0]["knn_idx"], patches[1]["centers"], patches[1]["knn_idx"])
        dense = dense.repeat_interleave(sparse.shape[0] // dense.shape[0], 0)
        masks, iou = mask_decoder_hier(sd, cfg, st["pc_embeddings"], st["pc_pe"], sparse, dense, coords, patches, multimask_output, mode, st["cache"])
    return (masks, iou, dense) if return_dense else (masks, iou)


@torch.no_grad()
def forward_eval(sd, cfg, coords, features, gt_masks, prompt_iters=None, mode="exact"):
    """PointCloudSAMNN.forward(..., is_eval=True) (pc_sam.py:264-374): encoder once, then prompt_iters x {deterministic click from the error
    region, decode with all clicks so far and the previous best mask}."""
    st = encode(sd, cfg, coords, features, mode)
    B, M, N = gt_masks.shape
    pc = coords.new_empty((B * M, 0, 3))
    pl = torch.empty((B * M, 0), dtype=torch.bool)
    pm, outs = None, []
    for i in range(prompt_iters or cfg.prompt_iters):
        nc, nl = O.sample_eval_prompts(coords, gt_masks, pm)
        pc, pl = torch.cat([pc, nc], 1), torch.cat([pl, nl], 1)
        masks, iou = decode(sd, cfg, st, pc, pl, pm, multimask_output=(i == 0), mode=mode)
        if i == 0:
            pm = torch.gather(masks, 1, iou.argmax(1).view(-1, 1, 1).expand(-1, 1, N))[:, 0]
        else:
            pm = masks[:, 0]
        outs.append(dict(prompt_coords=pc, prompt_labels=pl, masks=masks, iou_preds=iou, prompt_masks=pm))
    return outs
