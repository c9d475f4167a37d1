"""The reference's two model variants on the HIP library, behind the same host interface as PointCloudSAM.

  PointCloudSAMNN   (configs/model/voronoi.yaml; pc_sam/model/pc_sam.py:199-374): every point belongs to its nearest FPS centre.
      NNGrouper common.py:190-212 | PatchEmbedNN + Block pc_encoder.py:147-198 | MaskEncoderNN + ResMlp prompt_encoder.py:186-211,255-300 |
      MaskDecoder (the base one).
  PointCloudSAMHier (configs/model/hier.yaml; pc_sam.py:377-496): two kNN levels.
      PatchEmbedHier pc_encoder.py:201-239 | MaskEncoderHier prompt_encoder.py:136-183 | MaskDecoderHier mask_decoder.py:214-370.

Both reuse the base class for everything they share with it (the ViT blocks, the prompt point encoder, the two-way transformer, the IoU head,
the click simulation of `forward`); what differs -- tokenizer, patch embedding, mask encoder, and for hier the upscaling -- is overridden here as
sequences of kernels of csrc/libpointsam_hip.so (new for the voronoi model: psam_nn_group_feats, psam_scatter_amax).

The reference as found (see oracle/variants_oracle.py, which restates it and is pinned by golden runs of the reference's own classes):
  * PointCloudSAMNN.predict_masks cannot run in the reference (pc_sam.py:232-262 reads a `knn_idx` NNGrouper does not return); here it is
    the per-iteration call of its forward (pc_sam.py:326-352), which is consistent.
  * MaskEncoderNN scatters the rows of ALL mask sets into the cells of the first one (no batch offset, prompt_encoder.py:286-297); results
    must equal the reference's, so this path does the same (set_stride = 0 below) -- for one mask set it is the intended pooling.
  * PointCloudSAMHier defines only forward, with the random click sampler (common.py:319-365); `forward` here uses the deterministic
    evaluation sampler of the base model (the reference's own choice for evaluation, common.py:287-316), predict_masks is the same decode.
"""
import warnings

import torch

from . import ops
from .model import EncoderState, PointCloudSAM, TokenizerState
from .ops import ACT_GELU, ACT_NONE


def _pad_cols(w: torch.Tensor, k: int) -> torch.Tensor:
    """[N, K0] -> [N, k] with zero columns (a Linear on zero-padded rows: the same sums)."""
    out = torch.zeros(w.shape[0], k, dtype=w.dtype, device=w.device)
    out[:, :w.shape[1]] = w
    return out.contiguous()


def _r4(n: int) -> int:
    return (n + 3) // 4 * 4


class _GenericPatchEncoder:
    """PatchEncoder.forward (common.py:499-506) for any (Cin, hidden, Cout) on already gathered rows [groups*K, width >= Cin] (zero-padded to a
    multiple of 4 columns): Linear, LayerNorm, GELU, Linear | max over the group | cat([max, x]) Linear (as two GEMMs: the pooled half once
    per group, added as a row bias) , LayerNorm, GELU, Linear | max over the group."""

    @staticmethod
    def run(model, prefix, rows, K):
        w, eps = model.w, model.cfg.ln_eps
        w0 = model._padded(prefix + ".conv1.0.weight", rows.shape[1])
        h = ops.linear(rows, w0, w[prefix + ".conv1.0.bias"])
        model._ln(prefix + ".conv1.1", h, eps, act=ACT_GELU, out=h)
        h = model._lin(prefix + ".conv1.3", h)
        y = ops.group_max(h, K)
        w2a = w[prefix + ".conv2.0.weight"]
        a = h.shape[1]
        g1 = ops.linear(y, model._slice(prefix + ".conv2.0.weight", 0, a), w[prefix + ".conv2.0.bias"])
        h2 = ops.linear(h, model._slice(prefix + ".conv2.0.weight", a, w2a.shape[1]), None, rowbias=g1, rowgroup=K)
        model._ln(prefix + ".conv2.1", h2, eps, act=ACT_GELU, out=h2)
        return ops.group_max(model._lin(prefix + ".conv2.3", h2), K)


class _VariantBase(PointCloudSAM):
    def __init__(self, cfg, state_dict, device="cuda", precision: str = "f16x3"):
        super().__init__(cfg, state_dict, device, precision)
        self._derived = {}

    def _padded(self, name, k):
        key = (name, k)
        if key not in self._derived:
            self._derived[key] = _pad_cols(self.w[name], k)
        return self._derived[key]

    def _slice(self, name, c0, c1):
        key = (name, c0, c1)
        if key not in self._derived:
            self._derived[key] = self.w[name][:, c0:c1].contiguous()
        return self._derived[key]

    @torch.no_grad()
    def predict_masks(self, coords, features, prompt_coords, prompt_labels, prompt_masks=None, multimask_output=True, validate=True):
        """Encoder + one decode (the per-iteration call of the variant's forward, pc_sam.py:326-352 / :437-471)."""
        return super().predict_masks(coords, features, prompt_coords, prompt_labels, prompt_masks, multimask_output, validate)


class PointCloudSAMNN(_VariantBase):
    """Voronoi variant.  TokenizerState / EncoderState.knn_idx hold nn_idx [B, N]: the index of every point's nearest centre."""

    def __init__(self, cfg, state_dict, device="cuda", precision: str = "f16x3"):
        assert cfg.variant == "voronoi", cfg.variant
        super().__init__(cfg, state_dict, device, precision)

    @torch.no_grad()
    def tokenize(self, coords, with_interp: bool = True) -> TokenizerState:
        """NNGrouper's index work (common.py:198-201): FPS centres, nearest centre per point.  The 3-NN search of the decoder's interpolation
        (common.py:238-255) is the same search: its first neighbour is the nearest centre (lowest index on ties)."""
        coords = coords.to(self.device, torch.float32).contiguous()
        G = int(self.pc_encoder.patch_embed.grouper.num_groups)
        fps_idx, centers = ops.fps(coords, G)
        ii, iw = ops.three_nn(coords, centers)
        return TokenizerState(fps_idx, centers, ii[..., 0].contiguous(), ii, iw)

    def _nn_block(self, p, x):
        """Block.forward (pc_encoder.py:147-162): x + Linear(LayerNorm(GELU(Linear(LayerNorm(x)))))."""
        eps = self.cfg.ln_eps
        h = self._ln(p + ".norm", x, eps)
        h = self._lin(p + ".mlp.0", h, act=ACT_GELU)
        self._ln(p + ".mlp.2", h, eps, out=h)
        return self._lin(p + ".mlp.3", h, residual=x)

    def _patch_tokens(self, coords, features, tok):
        """PatchEmbedNN.forward (pc_encoder.py:181-198)."""
        P = "pc_encoder.patch_embed"
        B, N, _ = coords.shape
        G = tok.centers.shape[1]
        f = ops.nn_group_feats(coords, tok.centers, tok.knn_idx, feats=features, width=_r4(4 + features.shape[-1]))
        x = ops.linear(f, self._padded(P + ".in_proj.weight", f.shape[1]), self.w[P + ".in_proj.bias"])
        for i in range(3):
            x = self._nn_block(f"{P}.blocks1.{i}", x)
        y = ops.scatter_amax(x, tok.knn_idx, B * G, rows_per_set=N, set_stride=G, include_self=False)
        for i in range(3):
            y = self._nn_block(f"{P}.blocks2.{i}", y)
        y = self._ln(P + ".norm", y, self.cfg.ln_eps)
        return self._lin(P + ".out_proj", y)

    def _dense_prompt(self, st, pm, Z, N, use_center_idx):
        """MaskEncoderNN.forward (prompt_encoder.py:262-300), scatter as found (module docstring)."""
        eps, S = self.cfg.ln_eps, "mask_encoder.second_nn.mlp"
        B, G = st.coords.shape[0], st.centers.shape[1]
        pg = ops.nn_group_feats(st.coords, st.centers, st.knn_idx, logits=pm, width=8)
        feat = ops.linear(pg, self._padded("mask_encoder.first_nn.weight", 8), self.w["mask_encoder.first_nn.bias"])
        fixed = bool(getattr(self.cfg, "nn_mask_scatter_fixed", False))
        if Z > 1 and not fixed:
            warnings.warn("MaskEncoderNN with more than one mask set pools every set into the first set's cells, as the reference does (prompt_encoder.py:291-297); "
                          "ModelConfig.nn_mask_scatter_fixed=True gives each set its own cells", RuntimeWarning, stacklevel=2)
        agg = ops.scatter_amax(feat, st.knn_idx, Z * G, rows_per_set=N, set_stride=G if fixed else 0, idx_rep=Z // B, include_self=True)
        del feat
        x = self._lin(S + ".0", agg)
        self._ln(S + ".1", x, eps, act=ACT_GELU, out=x)
        for i in (3, 4, 5):
            t = self._lin(f"{S}.{i}.mlp.0", x)
            self._ln(f"{S}.{i}.mlp.1", t, eps, act=ACT_GELU, out=t)
            nx = torch.empty_like(x)
            ops.add_bcast(t, 1, x, nx, 1, x.shape[0], x.shape[1])
            x = nx
        return self._lin(S + ".6", x)


class PointCloudSAMHier(_VariantBase):
    """Two-level variant.  The transformer's tokens are the level-2 groups: TokenizerState / EncoderState.centers, .knn_idx are level 2
    (knn_idx indexes the level-1 centres); level 1 and both interpolation searches are in `.extra`."""

    def __init__(self, cfg, state_dict, device="cuda", precision: str = "f16x3"):
        assert cfg.variant == "hier", cfg.variant
        super().__init__(cfg, state_dict, device, precision)
        E, nmt = cfg.embed_dim, cfg.num_mask_tokens
        mlp = lambda pfx: [(self.w[f"{pfx}.layers.{j}.weight"], self.w[f"{pfx}.layers.{j}.bias"]) for j in range(3)]
        hyp = [mlp(f"mask_decoder.output_hypernetworks_mlps.{i}") for i in range(nmt)]
        self.hyper_mw = {True: ops.Mlp3Weights(hyp[1:]), False: ops.Mlp3Weights(hyp[:1])}      # output width E // 2 (mask_decoder.py:240-245)

    @torch.no_grad()
    def tokenize(self, coords, with_interp: bool = True) -> TokenizerState:
        """The two KNNGrouper levels of PatchEmbedHier (pc_encoder.py:231-237; level 2 = the first G2 level-1 centres, use_fps=False,
        common.py:92-96) and the two 3-NN searches of MaskDecoderHier's upscaling (mask_decoder.py:316-319, pc_sam.py:405-412)."""
        coords = coords.to(self.device, torch.float32).contiguous()
        (G1, G2), (K1, K2) = self.cfg.hier_groups, self.cfg.hier_sizes
        fps_idx, centers1 = ops.fps(coords, G1)
        knn1 = ops.knn(centers1, coords, K1)
        centers2 = centers1[:, :G2].contiguous()
        knn2 = ops.knn(centers2, centers1, K2)
        i1 = ops.three_nn(coords, centers1)
        i2 = ops.three_nn(centers1, centers2)
        return TokenizerState(fps_idx, centers2, knn2, None, None, extra=dict(centers1=centers1, knn1=knn1, interp1=i1, interp2=i2))

    def _patch_tokens(self, coords, features, tok):
        """PatchEmbedHier.forward (pc_encoder.py:231-239); keeps the level-1 embeddings for the decoder (mask_decoder.py:316)."""
        B = coords.shape[0]
        ex = tok.extra
        r1, r2 = self.cfg.hier_radius if self.cfg.hier_radius else (None, None)
        (G1, G2), (K1, K2) = self.cfg.hier_groups, self.cfg.hier_sizes
        f1 = ops.group_gather(coords, features, ex["centers1"], ex["knn1"], radius=r1, width=_r4(3 + features.shape[-1]))
        x1 = _GenericPatchEncoder.run(self, "pc_encoder.patch_embed.patch_encoder1", f1.view(B * G1 * K1, -1), K1)          # [B*G1, 128]
        ex["embeddings1"] = x1.view(B, G1, -1)
        f2 = ops.group_gather(ex["centers1"], ex["embeddings1"], tok.centers, tok.knn_idx, radius=r2, width=_r4(3 + x1.shape[1]))
        return _GenericPatchEncoder.run(self, "pc_encoder.patch_embed.patch_encoder2", f2.view(B * G2 * K2, -1), K2)      # [B*G2, patch_out]

    def _dense_prompt(self, st, pm, Z, N, use_center_idx):
        """MaskEncoderHier.forward (prompt_encoder.py:152-183) -> the level-2 embedding (pc_sam.py:449-457)."""
        ex = st.extra
        r1, r2 = self.cfg.hier_radius if self.cfg.hier_radius else (None, None)
        (G1, G2), (K1, K2) = self.cfg.hier_groups, self.cfg.hier_sizes
        f1 = ops.group_gather(st.coords, pm.view(Z, N, 1), ex["centers1"], ex["knn1"], radius=r1, width=4)
        x1 = _GenericPatchEncoder.run(self, "mask_encoder.patch_encoder1", f1.view(Z * G1 * K1, -1), K1)                     # [Z*G1, 128]
        f2 = ops.group_gather(ex["centers1"], x1.view(Z, G1, -1), st.centers, st.knn_idx, radius=r2, width=_r4(3 + x1.shape[1]))
        return _GenericPatchEncoder.run(self, "mask_encoder.patch_encoder2", f2.view(Z * G2 * K2, -1), K2)                  # [Z*G2, E]

    def _masks_from_keys(self, st, keys, hs, Z, T, rep, multimask_output, hyper=None):
        """MaskDecoderHier's upscaling (mask_decoder.py:312-332): G2 -> G1 interpolation, level-1 embeddings concatenated, output_upscaling2;
        G1 -> N interpolation, output_upscaling1; hyper-network products over E // 2 channels."""
        cfg, w, E = self.cfg, self.w, self.cfg.embed_dim
        ex = st.extra
        B, N = st.coords.shape[:2]
        G1, G2 = cfg.hier_groups
        nmt, Eh, D1 = cfg.num_mask_tokens, E // 2, cfg.hier_dim1
        U2, U1 = "mask_decoder.output_upscaling2", "mask_decoder.output_upscaling1"
        (ii2, iw2), (ii1, iw1) = ex["interp2"], ex["interp1"]
        # cat([interp(keys), level-1 embeddings]) @ W^T = interp(keys) @ W[:, :E]^T + embeddings1 @ W[:, E:]^T (the second term once per cloud)
        up2 = torch.empty(Z * G1, E, device=self.device)
        ops.interp3(keys.view(Z, G2, E), ii2, iw2, up2, rep)
        e1 = ops.linear(ex["embeddings1"].reshape(B * G1, D1), self._slice(U2 + ".0.weight", E, E + D1), w[U2 + ".0.bias"])        # [B*G1, E]
        h = ops.linear(up2, self._slice(U2 + ".0.weight", 0, E), None)
        h2 = torch.empty_like(h)
        ops.add_bcast(e1, rep, h, h2, Z, G1, E)
        self._ln(U2 + ".1", h2, cfg.ln_eps, act=ACT_GELU, out=h2)
        h2 = self._lin(U2 + ".3", h2)                                                                                               # [Z*G1, E]
        up1 = torch.empty(Z * N, E, device=self.device)
        ops.interp3(h2.view(Z, G1, E), ii1, iw1, up1, rep)
        u = self._lin(U1 + ".0", up1)                                                                                               # [Z*N, E/2]
        self._ln(U1 + ".1", u, cfg.ln_eps, act=ACT_GELU, out=u)
        u = self._lin(U1 + ".3", u, act=ACT_GELU)
        sel = list(range(1, nmt)) if multimask_output else [0]
        C = len(sel)
        if hyper is None:
            hyper = torch.empty(Z, C, Eh, device=self.device)
            ops.mlp3(hs[:, 1 + sel[0], :], T * E, E, self.hyper_mw[bool(multimask_output)], hyper, C * Eh, Eh, Z)
        masks = torch.empty(Z, C, N, device=self.device)
        ops.gemm_batched(hyper, u, masks, C, N, Eh, Eh, Eh, N, C * Eh, N * Eh, C * N, Z)
        return masks, sel


def build_model(cfg, state_dict, device="cuda", precision: str = "f16x3"):
    """The model class of a configuration (cfg.variant): PointCloudSAM | PointCloudSAMNN | PointCloudSAMHier."""
    cls = {"knn": PointCloudSAM, "voronoi": PointCloudSAMNN, "hier": PointCloudSAMHier}[cfg.variant]
    return cls(cfg, state_dict, device, precision)
