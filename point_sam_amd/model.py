"""Host-side mirror of the reference model interface for the inference hot path, executing on the HIP library.

Mirrors (same names, argument meaning, return values, error behaviour):
  pc_sam/model/pc_sam.py:20-88      PointCloudSAM.predict_masks
  pc_sam/model/pc_sam.py:112-194    the encoder-once / decoder-per-click data flow (``encode`` + ``decode``)
  evaluation/eval_kitti.py:352-362  run-time mutation of ``model.pc_encoder.patch_embed.grouper.{num_groups,group_size}``

Python here only sequences kernel launches and owns buffers (torch tensors); all arithmetic is in
csrc/libpointsam_hip.so.  There is no CPU path: tensors must be on the GPU.
"""
import math
from dataclasses import dataclass, field
from types import SimpleNamespace
from typing import Optional

import torch

from . import ops
from .config import ModelConfig
from .ops import ACT_GELU, ACT_NONE, ACT_RELU
from .streams import pipeline_streams
from .weights import check_state_dict


def _round_up(x, m):
    return (x + m - 1) // m * m


@dataclass
class TokenizerState:
    """Index work that depends on the coordinates only (FPS, kNN grouping, 3-NN interpolation weights): the part of the
    path that is latency-bound on a handful of CUs and can run ahead on its own stream (see BatchPipeline)."""
    fps_idx: torch.Tensor        # [B, G] int64
    centers: torch.Tensor        # [B, G, 3]
    knn_idx: torch.Tensor        # [B, G, K] int64
    interp_index: torch.Tensor   # [B, N, 3] int64
    interp_weight: torch.Tensor  # [B, N, 3]
    extra: Optional[dict] = None  # model variants (point_sam_amd/variants.py): e.g. the hier model's level-1 groups

    def tensors(self):
        """Every device tensor of the state (for stream bookkeeping): the fixed fields that are present plus the tensors inside `extra`
        (nested tuples / lists included)."""
        out = [t for t in (self.fps_idx, self.centers, self.knn_idx, self.interp_index, self.interp_weight) if t is not None]

        def walk(v):
            if isinstance(v, torch.Tensor):
                out.append(v)
            elif isinstance(v, (tuple, list)):
                for u in v:
                    walk(u)
            elif isinstance(v, dict):
                for u in v.values():
                    walk(u)
        walk(self.extra)
        return tuple(out)


@dataclass
class EncoderState:
    """What the reference keeps between encoder and decoder: the ``patches`` dict (common.py:121-123) and
    ``AuxInputs`` (mask_decoder.py:12-18)."""
    coords: torch.Tensor
    features: torch.Tensor
    pc_embeddings: torch.Tensor      # [B, G, E]
    pc_pe: torch.Tensor              # [B, G, E]
    centers: torch.Tensor            # [B, G, 3]
    knn_idx: torch.Tensor            # [B, G, K] int64
    fps_idx: torch.Tensor            # [B, G] int64
    patch_embeddings: torch.Tensor   # [B, G, patch_out]
    interp_index: Optional[torch.Tensor] = None   # [B, N, 3] int64 (cached after the first decode)
    interp_weight: Optional[torch.Tensor] = None  # [B, N, 3]
    extra: Optional[dict] = None                  # model variants (point_sam_amd/variants.py)


class PointCloudSAM:
    """HIP implementation behind the reference's ``PointCloudSAM`` inference interface."""

    def __init__(self, cfg: ModelConfig, state_dict, device="cuda", precision: str = "f16x3"):
        """precision: arithmetic of the large GEMMs (and, for "f16x3", of the encoder attention) --
        "f32": v_mfma_f32_32x32x2_f32, exact fp32 products;
        "bf16x6": exact 3-way bf16 split of both operands, 6 partial products on the bf16 matrix pipe (csrc/gemm_split.hip);
        "f16x3": power-of-two row scales + 2-way fp16 split, 3 partial products on the fp16 matrix pipe
        (csrc/gemm_f16x3.hip, csrc/attention.hip) -- the fastest; bench.py's default.
        All three accumulate in fp32, keep fp32 tensors in HBM and meet the same parity tests at the same error level."""
        check_state_dict(cfg, state_dict)
        if precision not in ops.GEMM_MODES:
            raise ValueError(f"precision must be one of {ops.GEMM_MODES}")
        self.precision = precision
        self.c_blocks = True      # "f16x3", EVA02 blocks: one psam_eva_block call per layer (csrc/blocks.hip: the same eight launches, sequenced and
                                  # packed by the library) instead of sequencing them here; False: the Python sequence below (tests A/B both)
        self.fuse_tokens = False  # the decoder's token side as one launch per two-way layer (csrc/experiments/twoway.hip) instead of ~22: parity-green but
                                  # slower (0.56 vs 0.48 ms at cfg #2, profiles/r03/r03_twoway.txt), so off; tests A/B both
        self._tw = None
        self.row_bounds = True    # "f16x3": the fused MLP's packed rows are scaled by a per-row bound from ||h||_2 (False: the (k1 / scale + k2)^2 form)
        self.fuse_mlp = True      # "f16x3": EVA02 MLP as two GEMMs with nothing in between (False = separate inner LayerNorm; tests A/B both)
        self.fuse_attn_pack = True  # "f16x3": the attention kernel writes its output packed for the output projection (bound-derived scale)
        # "f16x3", head dim 64: the qkv GEMM writes q | k | v already packed (one a-priori power-of-two scale) and the attention kernel consumes
        # them as they are -- K / V tiles by LDS-DMA, V transposed on read; no per-tile conversion, scaling or maximum (csrc/attention.hip)
        self.fuse_attn_operands = True
        # upscaling MLP: the first Linear runs on the G patch rows BEFORE the 3-NN interpolation (an affine combination: it commutes with a
        # Linear layer) and the LayerNorm + GELU after it ride on the interpolation kernel -- the [N, 256] GEMM and LayerNorm pass are gone
        self.upscale_linear_first = True
        self.fuse_hyper = True      # "f16x3": the hyper-network products taken inside the last upscaling GEMM's epilogue
        self.fuse_upscale = True  # "f16x3": the 3-NN interpolation hands its rows to the upscaling MLP packed (no pack pass)
        # "f16x3": the upscaling MLP's LayerNorm/GELU and the hyper-network products inside two full-row GEMM epilogues.  Parity-tested,
        # but OFF: the 128x256 one-wave-per-SIMD tile it needs runs 383 us against 188 us for the 128x128 tile at [262144, 256, 256],
        # so the chain measured 1146 us fused against 853 us (scripts/exp/upscale_bench.py, profiles/r02/r02_upscale_chain.log).
        self.fuse_upscale_rows = False
        self.fuse_patch = True    # "f16x3": mini-PointNet hand-overs packed, max-pools in the GEMM epilogues (False = separate kernels)
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("point_sam_amd.PointCloudSAM runs on the GPU only (HIP kernels; no CPU fallback)")
        ops._lib.load()  # fail loudly now if the HIP library is missing
        self.prompt_iters = cfg.prompt_iters
        # attribute tree the reference's callers mutate (evaluation/eval_kitti.py:353-362)
        grouper = SimpleNamespace(num_groups=cfg.num_groups, group_size=cfg.group_size)
        self.pc_encoder = SimpleNamespace(patch_embed=SimpleNamespace(grouper=grouper), embed_dim=cfg.embed_dim)
        self.w = {k: v.to(self.device, torch.float32).contiguous() for k, v in state_dict.items()}
        self._pack()
        self._flag = torch.zeros(1, dtype=torch.int32, device=self.device)

    # ------------------------------------------------------------------------------------------ weights
    def _pack(self):
        w, cfg, D = self.w, self.cfg, self.cfg.vit.dim
        self.blocks = []
        for i in range(cfg.vit.depth):
            p = f"pc_encoder.transformer.blocks.{i}"
            blk = SimpleNamespace()
            if cfg.vit.swiglu:
                blk.wqkv = torch.cat([w[p + ".attn.q_proj.weight"], w[p + ".attn.k_proj.weight"], w[p + ".attn.v_proj.weight"]], 0).contiguous()
                blk.bqkv = torch.cat([w[p + ".attn.q_proj.bias"], torch.zeros(D, device=self.device), w[p + ".attn.v_proj.bias"]]).contiguous()
                H = cfg.vit.mlp_hidden
                Hp = _round_up(H, 32)
                # fc1_g / fc1_x packed as alternating 32-row blocks (PSAM_ACT_SWIGLU epilogue: g and x of the same hidden
                # unit land in the same lane/register of adjacent accumulator tiles); hidden padded to Hp with zero rows.
                def pad_rows(t):
                    out = torch.zeros((Hp,) + tuple(t.shape[1:]), device=self.device)
                    out[:H] = t
                    return out
                gw, xw = pad_rows(w[p + ".mlp.fc1_g.weight"]), pad_rows(w[p + ".mlp.fc1_x.weight"])
                gb, xb = pad_rows(w[p + ".mlp.fc1_g.bias"]), pad_rows(w[p + ".mlp.fc1_x.bias"])
                w1 = torch.stack([gw.view(Hp // 32, 32, D), xw.view(Hp // 32, 32, D)], 1).reshape(2 * Hp, D).contiguous()
                b1 = torch.stack([gb.view(Hp // 32, 32), xb.view(Hp // 32, 32)], 1).reshape(2 * Hp).contiguous()
                w2 = torch.zeros(D, Hp, device=self.device)
                w2[:, :H] = w[p + ".mlp.fc2.weight"]
                blk.w1, blk.b1, blk.w2, blk.hp = w1, b1, w2, Hp
            else:
                blk.wqkv = w[p + ".attn.qkv.weight"]
                blk.bqkv = torch.cat([w[p + ".attn.q_bias"], torch.zeros(D, device=self.device), w[p + ".attn.v_bias"]]).contiguous()
                blk.w1, blk.b1, blk.w2 = w[p + ".mlp.fc1.weight"], w[p + ".mlp.fc1.bias"], w[p + ".mlp.fc2.weight"]
            blk.p = p
            self.blocks.append(blk)
        self.out_tokens = torch.cat([w["mask_decoder.iou_token.weight"], w["mask_decoder.mask_tokens.weight"]], 0).contiguous()
        # hyper-networks and IoU head as one launch each (psam_mlp3): stacked weights; multimask uses MLPs 1.., single mask MLP 0
        mlp = lambda pfx: [(w[f"{pfx}.layers.{j}.weight"], w[f"{pfx}.layers.{j}.bias"]) for j in range(3)]
        hyp = [mlp(f"mask_decoder.output_hypernetworks_mlps.{i}") for i in range(cfg.num_mask_tokens)]
        self.hyper_mw = {True: ops.Mlp3Weights(hyp[1:]), False: ops.Mlp3Weights(hyp[:1])}
        self.iou_mw = ops.Mlp3Weights([mlp("mask_decoder.iou_prediction_head")])
        # "f16x3": every static weight that can feed the packed-operand GEMM (csrc/gemm_f16x3p.hip) is scaled, split and packed ONCE,
        # here, and owned by this model (fw: name -> ops.F16Weight; launches below the split thresholds use its fp32 original)
        self.fw = {}
        self.pe_bound = {}
        self.ln_bound = {}       # prefix -> a-priori bound of the conv2.1 LayerNorm's output (packed output of the fused conv2.0 GEMM)
        if self.precision == "f16x3":
            h0 = cfg.patch_hidden[0]
            for name, t in w.items():
                if name.endswith(".weight") and t.dim() == 2 and ops.F16Weight.eligible(*t.shape) and not name.startswith("pc_encoder.transformer."):
                    self.fw[name] = ops.F16Weight(t)
            self.pe_bound = {}
            if "mask_decoder.output_upscaling.1.weight" in w:
                self.up_ln_bound = ops.row_ln_bound(w["mask_decoder.output_upscaling.1.weight"], w["mask_decoder.output_upscaling.1.bias"])
            for prefix in ("pc_encoder.patch_embed.patch_encoder", "mask_encoder.patch_encoder"):   # cat([max, x]) @ W^T as two GEMMs
                if prefix + ".conv1.3.weight" not in w:      # model variants have other patch embeddings (point_sam_amd/variants.py)
                    continue
                w13 = w[prefix + ".conv1.3.weight"]     # bound of |conv1.3 row| from the scale of its (packed) input row: see psam_gemm_fuse_t
                self.pe_bound[prefix] = (float(2.0 ** 15 * math.sqrt(w13.shape[1]) * w13.double().norm(dim=1).max().item()),
                                         float(w[prefix + ".conv1.3.bias"].abs().max().item()))
                w2a = w[prefix + ".conv2.0.weight"]
                for tag, sl in (("#max", w2a[:, :h0]), ("#x", w2a[:, h0:])):
                    if ops.F16Weight.eligible(*sl.shape):
                        self.fw[prefix + ".conv2.0.weight" + tag] = ops.F16Weight(sl)
            for blk in self.blocks:
                if cfg.vit.swiglu and ops.F16Weight.eligible(*blk.w2.shape):
                    # fused MLP (csrc/gemm_f16x3p.hip, psam_gemm_fuse_t): fc1 emits the gated rows already packed for fc2 (scale from the
                    # bound B^2, B = 2^15 sqrt(D) max_n ||W1[n]|| / scale(h) + max|b1|) plus LayerNorm partials; the inner LayerNorm is
                    # folded into fc2:  fc2(LN(u)) = rstd (u (W2 gamma)^T - mean c) + d,  c = (W2 gamma) 1,  d = W2 beta + b2.
                    H, Hp = cfg.vit.mlp_hidden, blk.hp
                    gam = torch.zeros(Hp, device=self.device, dtype=torch.float64)
                    gam[:H] = w[blk.p + ".mlp.norm.weight"].double()
                    w2g = blk.w2.double() * gam[None, :]
                    blk.ln_c = w2g.sum(1).float().contiguous()
                    blk.ln_d = (blk.w2.double()[:, :H] @ w[blk.p + ".mlp.norm.bias"].double() + w[blk.p + ".mlp.fc2.bias"].double()).float().contiguous()
                    blk.w2g = ops.F16Weight(w2g.float().contiguous())
                    blk.k1 = float(2.0 ** 15 * math.sqrt(D) * blk.w1.double().norm(dim=1).max().item())
                    blk.k2 = float(blk.b1.abs().max().item())
                    # per-row bound of the gated rows from t = ||h||_2 (the LayerNorm kernel emits it): hidden unit n has
                    # |silu(g_n) x_n| <= (a_n t + b_n)(c_n t + d_n) with a, c the fc1_g / fc1_x weight row norms and b, d the |biases|;
                    # the maximum over n of the quadratic is bounded coefficient-wise.  Outlier rows of fc1_g and fc1_x rarely share an n,
                    # and ||h||_2 is far below sqrt(D) max|h| when h has outlier channels: the (k1 / scale + k2)^2 form lost 10+ bits on
                    # heavy-tailed weights (tests/test_gpu_e2e.py::test_heavy_tailed_weights_against_oracle, profiles/r03/r03_heavy_diag.txt)
                    a_n, c_n = w[blk.p + ".mlp.fc1_g.weight"].double().norm(dim=1), w[blk.p + ".mlp.fc1_x.weight"].double().norm(dim=1)
                    b_n, d_n = w[blk.p + ".mlp.fc1_g.bias"].double().abs(), w[blk.p + ".mlp.fc1_x.bias"].double().abs()
                    blk.u_bound = (1.002 * float((a_n * c_n).max()), 1.002 * float((a_n * d_n + b_n * c_n).max()), 1.002 * float((b_n * d_n).max()) + 1e-30)
                # a-priori bounds of |q|, |k|, |v| (packed qkv for the operand-packed attention) and of |v| alone (its packed output):
                # |W_n . h + b_n| <= ||W_n||_2 ||h||_2 + |b_n|, and a LayerNorm output h = z * gamma + beta has ||h||_2 <= max|gamma| sqrt(D) + ||beta||_2
                g1, b1n = w[blk.p + ".norm1.weight"].double(), w[blk.p + ".norm1.bias"].double()
                hnorm = float(g1.abs().max() * math.sqrt(D) + b1n.norm())
                if not cfg.vit.swiglu:
                    # GELU MLP (the giant encoder): fc1 writes GELU(.) packed for fc2 with one a-priori scale, |GELU(t)| <= |t|
                    g2, b2n = w[blk.p + ".norm2.weight"].double(), w[blk.p + ".norm2.bias"].double()
                    blk.fc1_bound = 1.001 * (float(blk.w1.double().norm(dim=1).max()) * float(g2.abs().max() * math.sqrt(D) + b2n.norm())
                                             + float(blk.b1.abs().max())) + 1e-30
                    blk.u_bound = (0.0, 1.002 * float(blk.w1.double().norm(dim=1).max()), 1.002 * float(blk.b1.abs().max()) + 1e-30)      # per row, from ||h||_2
                wq_all = blk.wqkv.double()
                blk.qkv_bound = 1.001 * (float(wq_all.norm(dim=1).max()) * hnorm + float(blk.bqkv.abs().max())) + 1e-30
                blk.v_bound = 1.001 * (float(wq_all[2 * D:].norm(dim=1).max()) * hnorm + float(blk.bqkv[2 * D:].abs().max())) + 1e-30
                # bound of |V| from the scale of the (LayerNorm) row that produced it: the attention kernel packs its output with it
                wv = blk.wqkv[2 * D:].double()
                blk.vk1 = float(2.0 ** 15 * math.sqrt(D) * wv.norm(dim=1).max().item())
                blk.vk2 = float(blk.bqkv[2 * D:].abs().max().item())
                for attr in ("wqkv", "w1", "w2"):
                    t = getattr(blk, attr)
                    if ops.F16Weight.eligible(*t.shape):
                        setattr(blk, attr, ops.F16Weight(t))
                pw = w[blk.p + ".attn.proj.weight"]
                if ops.F16Weight.eligible(*pw.shape):
                    self.fw[blk.p + ".attn.proj.weight"] = ops.F16Weight(pw)
            self.c_patch, self.c_upscale = {}, None      # coarse C-ABI entries (csrc/blocks.hip): the library's own packing + sequencing
            for prefix in ("pc_encoder.patch_embed.patch_encoder", "mask_encoder.patch_encoder"):
                if prefix in self.pe_bound and ops.CPatchEncoder.supported(w[prefix + ".conv1.0.weight"].shape[0], w[prefix + ".conv2.0.weight"].shape[0],
                                                                            w[prefix + ".conv2.3.weight"].shape[0]):
                    self.c_patch[prefix] = ops.CPatchEncoder(w, prefix, cfg.ln_eps)
            if cfg.embed_dim == 256 and "mask_decoder.output_upscaling.0.weight" in w:
                self.c_upscale = ops.CUpscale(w, cfg.ln_eps)
            self.c_twoway = ops.CTwoWay(w, "mask_decoder.transformer", cfg.dec_depth, cfg.embed_dim, cfg.dec_heads, cfg.dec_mlp, cfg.dec_downsample, cfg.ln_eps) \
                if cfg.dec_depth <= 4 and cfg.embed_dim % 32 == 0 and cfg.dec_mlp % 32 == 0 and (cfg.embed_dim // cfg.dec_downsample) % cfg.dec_heads == 0 \
                and cfg.embed_dim % cfg.dec_heads == 0 else None      # psam_twoway_decoder_prepare's preconditions: anything else runs the Python sequence
            if cfg.vit.swiglu and ops.EvaBlock.supported(D, cfg.vit.heads, cfg.vit.mlp_hidden):
                for blk in self.blocks:     # the library's own packing of the block (psam_eva_block_prepare)
                    blk.c_block = ops.EvaBlock(w, blk.p, D, cfg.vit.heads, cfg.vit.mlp_hidden, cfg.vit.ln_eps)
            elif not cfg.vit.swiglu and ops.EvaGeluBlock.supported(D, cfg.vit.heads, cfg.vit.mlp_hidden):
                for blk in self.blocks:     # the giant encoder's block (psam_eva_gelu_block_prepare)
                    blk.c_block = ops.EvaGeluBlock(w, blk.p, D, cfg.vit.heads, cfg.vit.mlp_hidden, cfg.vit.ln_eps)
            torch.cuda.current_stream(self.device).synchronize()   # the packed weights are consumed from several streams afterwards

    # ------------------------------------------------------------------------------------------ building blocks
    def _lin(self, name, x, **kw):
        return ops.linear(x, self.fw.get(name + ".weight", self.w[name + ".weight"]), self.w.get(name + ".bias"), **kw)

    def _ln(self, name, x, eps, **kw):
        return ops.layernorm(x, self.w[name + ".weight"], self.w[name + ".bias"], eps, **kw)

    def _ln_feeds_gemm(self, x, consumer):
        """(pack, row-scale buffer) for a LayerNorm whose only consumer is a large "f16x3" GEMM over all of its columns: the LN
        kernel then emits the GEMM's row scales and, on its float4 path, the packed hi|lo operand itself."""
        if self.precision != "f16x3" or x.shape[0] < ops.SPLIT_MIN_M or (consumer + ".weight") not in self.fw or not ops.layernorm_can_pack(x.shape[1]) \
                or x.shape[1] % 32 != 0:
            return False, None
        return True, torch.empty(x.shape[0], dtype=torch.float32, device=x.device)

    def _patch_encoder(self, prefix, coords, feats, centers, knn_idx, radius=None, center_idx=None):
        """PatchEncoder.forward on gathered groups (common.py:499-506) -> [B*rep*G, Cout].  center_idx [B,G]: centralize_features."""
        w, eps = self.w, self.cfg.ln_eps
        K = knn_idx.shape[2]
        h0 = self.cfg.patch_hidden[0]
        rows = feats.shape[0] * knn_idx.shape[1] * K
        l1 = (w[prefix + ".conv1.0.weight"], w[prefix + ".conv1.0.bias"], w[prefix + ".conv1.1.weight"], w[prefix + ".conv1.1.bias"], eps)
        w2a = w[prefix + ".conv2.0.weight"]
        # group sizes 32 / 64: the GEMM epilogues pool whole groups; multiples of 64 (cfg #3: 256): they pool 64-row parts, a small pass pools the parts
        fused = (self.precision == "f16x3" and self.fuse_patch and (K in (32, 64) or K % 64 == 0) and ops.fuse_supported(rows, 128) and prefix in self.pe_bound
                 and all((prefix + n) in self.fw for n in (".conv1.3.weight", ".conv2.0.weight#x", ".conv2.3.weight")))
        if fused and self.c_blocks and prefix in getattr(self, "c_patch", {}) and ops.current_gemm_mode() == "f16x3":
            return self.c_patch[prefix].run(coords, feats, centers, knn_idx, radius=radius, center_idx=center_idx)      # psam_patch_encoder: the same six launches
        if fused:
            # every hand-over stays in the GEMMs' packed form and both max-pools happen in GEMM epilogues: the [rows, 128] / [rows, 512]
            # activations are written once (packed) or not at all (conv2.3 only leaves its pooled [groups, Cout] rows)
            groups = rows // K
            s1 = torch.empty(rows, dtype=torch.float32, device=coords.device)
            h1 = ops.patch_l1(coords, feats, centers, knn_idx, *l1, radius=radius, center_idx=center_idx, scale_out=s1)
            h2 = torch.empty(rows, h0, dtype=torch.float32, device=coords.device)          # g8-packed container
            s2 = torch.empty(rows, dtype=torch.float32, device=coords.device)
            Kp = K if K <= 64 else 64             # rows pooled per epilogue group
            parts = K // Kp
            y1 = torch.empty(groups * parts, h0, dtype=torch.float32, device=coords.device)
            k1, k2 = self.pe_bound[prefix]
            self._lin(prefix + ".conv1.3", h1, x_scale=s1, x_packed=True, out=h2, pack_out=(s2, k1, k2), group_max_out=y1, group_max_k=Kp)
            if parts > 1:
                y1 = ops.group_max(y1, parts)
            del h1
            g1 = ops.linear(y1, self.fw.get(prefix + ".conv2.0.weight#max", w2a[:, :h0]), w[prefix + ".conv2.0.bias"])
            hd1 = w2a.shape[0]
            if ops.fused_row_ln(hd1) and hd1 == 512 and (prefix + ".conv2.3.weight") in self.fw and rows % 128 == 0:
                # conv2.0 -> LayerNorm -> GELU as ONE GEMM on full-row tiles: the [rows, 512] fp32 activation and the LayerNorm pass over it never exist
                h3 = torch.empty(rows, hd1, dtype=torch.float32, device=coords.device)     # g8-packed container
                rs, pk = torch.empty(rows, dtype=torch.float32, device=coords.device), True
                bound = self.ln_bound.get(prefix)
                if bound is None:
                    bound = self.ln_bound.setdefault(prefix, 1.001 * ops.row_ln_bound(w[prefix + ".conv2.1.weight"], w[prefix + ".conv2.1.bias"]) + 1e-30)
                ops.linear(h2, self.fw[prefix + ".conv2.0.weight#x"], None, act=ACT_GELU, rowbias=g1, rowgroup=K, x_scale=s2, x_packed=True, out=h3,
                           row_ln=(w[prefix + ".conv2.1.weight"], w[prefix + ".conv2.1.bias"], eps), pack_out=(rs, 0.0, bound))
                del h2
            else:
                h3 = ops.linear(h2, self.fw[prefix + ".conv2.0.weight#x"], None, rowbias=g1, rowgroup=K, x_scale=s2, x_packed=True)
                del h2
                pk, rs = self._ln_feeds_gemm(h3, prefix + ".conv2.3")
                self._ln(prefix + ".conv2.1", h3, eps, act=ACT_GELU, out=h3, scale_out=rs, pack=pk)
            emb = torch.empty(groups * parts, self.w[prefix + ".conv2.3.weight"].shape[0], dtype=torch.float32, device=coords.device)
            self._lin(prefix + ".conv2.3", h3, x_scale=rs, x_packed=pk, group_max_out=emb, group_max_k=Kp, no_store=True)
            return ops.group_max(emb, parts) if parts > 1 else emb
        h1 = ops.patch_l1(coords, feats, centers, knn_idx, *l1, radius=radius, center_idx=center_idx)
        h2 = self._lin(prefix + ".conv1.3", h1)
        del h1
        y1 = ops.group_max(h2, K)
        # cat([max, x]) @ W^T = max @ W[:, :h0]^T (one row per group) + x @ W[:, h0:]^T
        g1 = ops.linear(y1, self.fw.get(prefix + ".conv2.0.weight#max", w2a[:, :h0]), w[prefix + ".conv2.0.bias"])
        h3 = ops.linear(h2, self.fw.get(prefix + ".conv2.0.weight#x", w2a[:, h0:]), None, rowbias=g1, rowgroup=K)
        del h2
        pk, rs = self._ln_feeds_gemm(h3, prefix + ".conv2.3")
        self._ln(prefix + ".conv2.1", h3, eps, act=ACT_GELU, out=h3, scale_out=rs, pack=pk)
        h4 = self._lin(prefix + ".conv2.3", h3, x_scale=rs, x_packed=pk)
        del h3
        return ops.group_max(h4, K)

    def _block(self, blk, x, B, L):
        vit = self.cfg.vit
        D, H, hd = vit.dim, vit.heads, vit.head_dim
        p = blk.p
        # "f16x3" GEMMs need a power-of-two scale per operand row and stage hi/lo fp16 planes: the LayerNorms that feed them emit
        # the scale and (when the float4 LN path applies) the packed planes directly, so the GEMM does no split arithmetic for A
        f16 = self.precision == "f16x3"
        if (f16 and self.c_blocks and hasattr(blk, "c_block") and self.fuse_mlp and self.fuse_attn_pack and self.fuse_attn_operands
                and self.row_bounds and ops.current_gemm_mode() == "f16x3" and x.is_contiguous()
                and (x.shape[0] % 256 == 0 if vit.swiglu else x.shape[0] >= ops.SPLIT_MIN_M)):
            return blk.c_block.run(x, B, L)
        pk = f16 and x.shape[0] >= ops.SPLIT_MIN_M and isinstance(blk.wqkv, ops.F16Weight) and ops.layernorm_can_pack(D)
        rs = torch.empty(x.shape[0], dtype=torch.float32, device=x.device) if pk else None
        h = self._ln(p + ".norm1", x, vit.ln_eps, scale_out=rs, pack=pk)
        o = torch.empty_like(x)
        if (pk and self.fuse_attn_operands and self.fuse_attn_pack and ops.attention_packed_supported(hd, x.shape[0], D) and D % 32 == 0
                and (p + ".attn.proj.weight") in self.fw):
            qkvp = torch.empty(x.shape[0], 3 * D, dtype=torch.float32, device=x.device)      # g8-packed containers
            sq = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
            ops.linear(h, blk.wqkv, blk.bqkv, x_scale=rs, x_packed=True, out=qkvp, pack_out=(sq, 0.0, blk.qkv_bound))
            so = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
            ops.attention_packed(qkvp, sq, o, so, B, H, L, hd, hd ** -0.5, blk.v_bound)
            self._lin(p + ".attn.proj", o, residual=x, out=x, x_scale=so, x_packed=True)
            qkv = None
        else:
            qkv = ops.linear(h, blk.wqkv, blk.bqkv, x_scale=rs, x_packed=pk)
        if qkv is None:
            pass
        elif pk and self.fuse_attn_pack and ops.attention_can_pack(hd) and D % 32 == 0 and (p + ".attn.proj.weight") in self.fw:
            so = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)      # the attention kernel leaves its rows packed for proj
            ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], o, B, H, L, L, hd, hd ** -0.5, pack=(rs, blk.vk1, blk.vk2, so))
            self._lin(p + ".attn.proj", o, residual=x, out=x, x_scale=so, x_packed=True)
        else:
            ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], o, B, H, L, L, hd, hd ** -0.5)
            self._lin(p + ".attn.proj", o, residual=x, out=x)
        fused_swiglu = (vit.swiglu and pk and self.fuse_mlp and hasattr(blk, "w2g") and ops.fuse_supported(x.shape[0], 2 * blk.hp)
                        and ops.fuse_supported(x.shape[0], D))
        fused_gelu = (not vit.swiglu and pk and self.fuse_mlp and isinstance(blk.w2, ops.F16Weight) and ops.fuse_supported(x.shape[0], vit.mlp_hidden)
                      and vit.mlp_hidden % 32 == 0 and ops.splitk_factor(x.shape[0], vit.mlp_hidden, blk.w1.Kp, ACT_GELU) == 1)
        ub = torch.empty(x.shape[0], dtype=torch.float32, device=x.device) if (fused_swiglu or fused_gelu) and self.row_bounds else None
        self._ln(p + ".norm2", x, vit.ln_eps, out=h, scale_out=rs, pack=pk, bound_out=None if ub is None else (ub,) + blk.u_bound)
        if vit.swiglu:
            # fc1 with the SiLU gate fused in the GEMM epilogue -> u [M, Hp] (pad columns exactly 0), inner LayerNorm over
            # the first H columns in place, then fc2 over K = Hp (zero-padded weight columns)
            Hh = vit.mlp_hidden
            M = x.shape[0]
            if fused_swiglu:
                # two GEMMs, nothing in between: fc1 writes the gated rows g8-packed (bound-derived scales su) and their LayerNorm
                # partials; fc2 runs on them with the LayerNorm folded in (mean / rstd per row from the partials)
                u = torch.empty(M, blk.hp, dtype=torch.float32, device=x.device)
                su = torch.empty(M, dtype=torch.float32, device=x.device)
                st = torch.empty(M, ops.stat_segs(2 * blk.hp), 2, dtype=torch.float32, device=x.device)
                ops.linear(h, blk.w1, blk.b1, act=ops.ACT_SWIGLU, x_scale=rs, x_packed=True, out=u, stats=(st, Hh),
                           pack_out=(su, blk.k1, blk.k2) if ub is None else (su, ub))
                mean, rstd = ops.ln_stats_finalize(st, Hh, vit.ln_eps)
                ops.linear(u, blk.w2g, blk.ln_d, residual=x, out=x, x_scale=su, x_packed=True, ln_fold=(mean, rstd, blk.ln_c))
                return x
            u = ops.linear(h, blk.w1, blk.b1, act=ops.ACT_SWIGLU, x_scale=rs, x_packed=pk)
            pk2 = pk and isinstance(blk.w2, ops.F16Weight) and ops.layernorm_can_pack(Hh)
            ops.layernorm(u[:, :Hh], self.w[p + ".mlp.norm.weight"], self.w[p + ".mlp.norm.bias"], vit.ln_eps, out=u[:, :Hh], scale_out=rs if pk2 else None,
                          pack=pk2)
            ops.linear(u, blk.w2, self.w[p + ".mlp.fc2.bias"], residual=x, out=x, x_scale=rs if pk2 else None, x_packed=pk2)
        else:
            M, Hh = x.shape[0], vit.mlp_hidden
            if fused_gelu:
                # fc1 leaves GELU(.) g8-packed for fc2 (one scale from the a-priori bound): no scale + pack pass over the [M, hidden] rows
                g = torch.empty(M, Hh, dtype=torch.float32, device=x.device)
                sg = torch.empty(M, dtype=torch.float32, device=x.device)
                ops.linear(h, blk.w1, blk.b1, act=ACT_GELU, x_scale=rs, x_packed=True, out=g, pack_out=(sg, 0.0, blk.fc1_bound) if ub is None else (sg, ub))
                ops.linear(g, blk.w2, self.w[p + ".mlp.fc2.bias"], residual=x, out=x, x_scale=sg, x_packed=True)
            else:
                g = ops.linear(h, blk.w1, blk.b1, act=ACT_GELU, x_scale=rs, x_packed=pk)
                ops.linear(g, blk.w2, self.w[p + ".mlp.fc2.bias"], residual=x, out=x)
        return x

    # ------------------------------------------------------------------------------------------ encoder
    def _prep_inputs(self, coords, features):
        return (coords.to(self.device, torch.float32).contiguous(), features.to(self.device, torch.float32).contiguous())

    @torch.no_grad()
    def tokenize(self, coords: torch.Tensor, with_interp: bool = True) -> TokenizerState:
        """FPS -> centers -> kNN groups (common.py:91-97) and, optionally ahead of time, the 3-NN interpolation
        indices/weights the decoder needs (common.py:238-255)."""
        coords = coords.to(self.device, torch.float32).contiguous()
        g = self.pc_encoder.patch_embed.grouper
        G, K = int(g.num_groups), int(g.group_size)
        fps_idx, centers = ops.fps(coords, G)
        knn_idx = ops.knn(centers, coords, K)
        ii, iw = ops.three_nn(coords, centers) if with_interp else (None, None)
        return TokenizerState(fps_idx, centers, knn_idx, ii, iw)

    @torch.no_grad()
    def encode(self, coords: torch.Tensor, features: torch.Tensor, tok: Optional[TokenizerState] = None) -> EncoderState:
        """PointCloudEncoder.forward (pc_encoder.py:118-145) + pe_layer(centers) (pc_sam.py:59)."""
        cfg, w = self.cfg, self.w
        coords, features = self._prep_inputs(coords, features)
        B, N, _ = coords.shape
        E = cfg.embed_dim
        if tok is None:
            tok = self.tokenize(coords, with_interp=False)
        with ops.gemm_mode(self.precision):
            return self._encode(coords, features, tok, B, N, E)

    def _encode(self, coords, features, tok, B, N, E):
        cfg, w = self.cfg, self.w
        fps_idx, centers, knn_idx = tok.fps_idx, tok.centers, tok.knn_idx
        G = centers.shape[1]
        emb = self._patch_tokens(coords, features, tok)
        x = self._lin("pc_encoder.patch_proj", emb)
        p1 = ops.pos_l1(centers, w["pc_encoder.pos_embed.0.weight"], w["pc_encoder.pos_embed.0.bias"])
        self._lin("pc_encoder.pos_embed.2", p1, residual=x, out=x)
        for blk in self.blocks:
            x = self._block(blk, x, B, G)
        h = self._ln("pc_encoder.transformer.fc_norm", x, cfg.vit.ln_eps)
        pc_emb = self._lin("pc_encoder.out_proj", h).view(B, G, E)
        pc_pe = torch.empty(B, G, E, device=self.device)
        ops.fourier_pe(centers, w["point_encoder.pe_layer.positional_encoding_gaussian_matrix"], pc_pe, G, G * E, flag=self._flag)
        return EncoderState(coords, features, pc_emb, pc_pe, centers, knn_idx, fps_idx, emb.view(B, G, -1), tok.interp_index, tok.interp_weight,
                            dict(tok.extra) if tok.extra is not None else None)      # the state owns its copy: a variant's encode() adds entries (level-1 embeddings)

    def _patch_tokens(self, coords, features, tok):
        """PatchEmbed.forward after the grouping (pc_encoder.py:36-41) -> [B*G, patch_out]; the model variants override this."""
        cfg = self.cfg
        return self._patch_encoder("pc_encoder.patch_embed.patch_encoder", coords, features, tok.centers, tok.knn_idx, radius=cfg.radius,
                                   center_idx=tok.fps_idx if cfg.centralize_features else None)

    # ------------------------------------------------------------------------------------------ decoder
    def _attn(self, prefix, q_in, k_in, v_in, Z, Lq, Lk):
        """Attention.forward up to (not including) out_proj (transformer.py:214-234)."""
        H = self.cfg.dec_heads
        q = self._lin(prefix + ".q_proj", q_in)
        k = self._lin(prefix + ".k_proj", k_in)
        v = self._lin(prefix + ".v_proj", v_in)
        inner = q.shape[1]
        hd = inner // H
        o = torch.empty(Z * Lq, inner, device=q.device)
        return ops.attention_small(q, k, v, o, Z, H, Lq, Lk, hd, 1.0 / math.sqrt(hd))

    def _two_way_fused(self, src, pos, tokens, Z, G, T, rep):
        """_two_way with the token side of every layer in one launch (csrc/experiments/twoway.hip): per layer the image-side k / v projections, the token
        kernel, then the image -> token attention and its projection + norm4 on the patch tokens."""
        cfg, E, eps, H = self.cfg, self.cfg.embed_dim, self.cfg.ln_eps, self.cfg.dec_heads
        P = "mask_decoder.transformer"
        if self._tw is None:
            self._tw = [ops.TwoWayLayerWeights(self.w, f"{P}.layers.{i}") for i in range(cfg.dec_depth)] + [ops.TwoWayLayerWeights(self.w, P, final=True)]
        R, IX = Z * T, self._tw[0].inner
        queries, keys = tokens.clone(), src
        k = torch.empty_like(keys)
        ws = ops.twoway_tokens_ws(max(lw.args.mlp for lw in self._tw), self.device)
        ktok, vtok = torch.empty(R, IX, device=self.device), torch.empty(R, IX, device=self.device)
        for i in range(cfg.dec_depth):
            L = f"{P}.layers.{i}"
            ops.add_bcast(pos, rep, keys, k, Z, G, E)
            kimg = self._lin(L + ".cross_attn_token_to_image.k_proj", k)
            vimg = self._lin(L + ".cross_attn_token_to_image.v_proj", keys)
            ops.twoway_tokens(self._tw[i], queries, tokens, kimg, vimg, Z, T, G, H, eps, ws, ktok, vtok, skip_pe=(i == 0))
            qimg = self._lin(L + ".cross_attn_image_to_token.q_proj", k)
            a = torch.empty(Z * G, IX, device=self.device)
            ops.attention_small(qimg, ktok, vtok, a, Z, H, G, T, IX // H, 1.0 / math.sqrt(IX // H))
            keys = self._ln(L + ".norm4", self._lin(L + ".cross_attn_image_to_token.out_proj", a), eps, residual=keys)
        ops.add_bcast(pos, rep, keys, k, Z, G, E)
        kimg = self._lin(P + ".final_attn_token_to_image.k_proj", k)
        vimg = self._lin(P + ".final_attn_token_to_image.v_proj", keys)
        ops.twoway_tokens(self._tw[-1], queries, tokens, kimg, vimg, Z, T, G, H, eps, ws)
        return queries, keys

    def _two_way(self, src, pos, tokens, Z, G, T, rep):
        """TwoWayTransformer.forward (transformer.py:61-100).  src [Z*G,E] (overwritten), pos [B,G,E], tokens [Z*T,E]."""
        cfg, E, eps = self.cfg, self.cfg.embed_dim, self.cfg.ln_eps
        P = "mask_decoder.transformer"
        if self.c_blocks and getattr(self, "c_twoway", None) is not None and ops.current_gemm_mode() == "f16x3" and not self.fuse_tokens:
            # psam_twoway_decoder: the library's REGROUPED launch sequence (fused Linear + LayerNorm launches, merged projections, exact-fp32 row kernels on the
            # patch side; csrc/blocks.hip) -- the same operators as the Python sequence below with other launch boundaries: equal to fp32 round-off (<= 1.5e-5,
            # test_twoway_decoder_regrouped_sequence_matches_operator_sequence), not bit for bit
            return self.c_twoway.run(tokens, src, pos, rep, Z, T, G)
        if self.fuse_tokens and ops.TwoWayLayerWeights.supported(E, self.w[P + ".layers.0.cross_attn_token_to_image.q_proj.weight"].shape[0], cfg.dec_heads, Z, T, G):
            return self._two_way_fused(src, pos, tokens, Z, G, T, rep)
        queries, keys = tokens, src
        q = torch.empty_like(tokens)
        k = torch.empty_like(keys)
        for i in range(cfg.dec_depth):
            L = f"{P}.layers.{i}"
            if i == 0:
                a = self._attn(L + ".self_attn", queries, queries, queries, Z, T, T)
                queries = self._ln(L + ".norm1", self._lin(L + ".self_attn.out_proj", a), eps)
            else:
                ops.add_bcast(queries, 1, tokens, q, Z, T, E)
                a = self._attn(L + ".self_attn", q, q, queries, Z, T, T)
                queries = self._ln(L + ".norm1", self._lin(L + ".self_attn.out_proj", a), eps, residual=queries)
            ops.add_bcast(queries, 1, tokens, q, Z, T, E)
            ops.add_bcast(pos, rep, keys, k, Z, G, E)
            a = self._attn(L + ".cross_attn_token_to_image", q, k, keys, Z, T, G)
            queries = self._ln(L + ".norm2", self._lin(L + ".cross_attn_token_to_image.out_proj", a), eps, residual=queries)
            m = self._lin(L + ".mlp.lin2", self._lin(L + ".mlp.lin1", queries, act=ACT_RELU))
            queries = self._ln(L + ".norm3", m, eps, residual=queries)
            ops.add_bcast(queries, 1, tokens, q, Z, T, E)
            a = self._attn(L + ".cross_attn_image_to_token", k, q, queries, Z, G, T)
            keys = self._ln(L + ".norm4", self._lin(L + ".cross_attn_image_to_token.out_proj", a), eps, residual=keys)
        ops.add_bcast(queries, 1, tokens, q, Z, T, E)
        ops.add_bcast(pos, rep, keys, k, Z, G, E)
        a = self._attn(P + ".final_attn_token_to_image", q, k, keys, Z, T, G)
        queries = self._ln(P + ".norm_final_attn", self._lin(P + ".final_attn_token_to_image.out_proj", a), eps, residual=queries)
        return queries, keys

    def _dense_prompt(self, st, pm, Z, N, use_center_idx):
        """MaskEncoder.forward on a given mask prompt pm [Z, N] (prompt_encoder.py:122-133) -> [Z*G, E]."""
        cfg = self.cfg
        if cfg.mask_centralize_features and not use_center_idx:
            raise ValueError("MaskEncoder.centralize_features needs center_idx: only PointCloudSAM.forward passes it (pc_sam.py:151-157); "
                             "predict_masks would fail in the reference too (prompt_encoder.py:122-131 with center_idx=None)")
        return self._patch_encoder("mask_encoder.patch_encoder", st.coords, pm.view(Z, N, 1), st.centers, st.knn_idx, radius=cfg.mask_encoder_radius,
                                   center_idx=st.fps_idx if cfg.mask_centralize_features else None)

    def _masks_from_keys(self, st, keys, hs, Z, T, rep, multimask_output, hyper=None):
        """Upscaling + hyper-network products (mask_decoder.py:146-176): keys [Z*G, E] after the transformer, hs [Z, T, E] -> (masks [Z, C, N],
        the selected mask-token indices).  hyper [Z, C, E]: the hyper-network outputs when the caller already has them (_decode: one launch with the
        IoU head)."""
        cfg, w, E = self.cfg, self.w, self.cfg.embed_dim
        N, G, nmt = st.coords.shape[1], st.centers.shape[1], cfg.num_mask_tokens
        # upscale: 3-NN interpolation G -> N, MLP, hyper-network dot product    (mask_decoder.py:146-176)
        if st.interp_index is None:
            st.interp_index, st.interp_weight = ops.three_nn(st.coords, st.centers)
        sel = list(range(1, nmt)) if multimask_output else [0]
        C = len(sel)
        if hyper is None:
            hyper = torch.empty(Z, C, E, device=self.device)
            ops.mlp3(hs[:, 1 + sel[0], :], T * E, E, self.hyper_mw[bool(multimask_output)], hyper, C * E, E, Z)      # mask token i -> MLP i
        masks = torch.empty(Z, C, N, device=self.device)
        up = torch.empty(Z * N, E, device=self.device)
        U0, U3 = "mask_decoder.output_upscaling.0", "mask_decoder.output_upscaling.3"
        packed_interp = self.precision == "f16x3" and self.fuse_upscale and E == 256 and Z * N >= ops.SPLIT_MIN_M and (U0 + ".weight") in self.fw
        if (self.upscale_linear_first and E == 256 and self.c_blocks and getattr(self, "c_upscale", None) is not None and packed_interp and self.fuse_hyper
                and (U3 + ".weight") in self.fw and (Z * N) % 256 == 0 and N % 32 == 0 and C <= 4 and ops.current_gemm_mode() == "f16x3" and keys.is_contiguous()):
            self.c_upscale.run(keys.view(Z * G, E), st.interp_index, st.interp_weight, hyper, masks, rep, Z, N, G, C)      # psam_upscale_masks: the same launches
            return masks, sel
        if self.upscale_linear_first and E == 256:
            k1 = self._lin(U0, keys.view(Z * G, E))
            pk = packed_interp and (U3 + ".weight") in self.fw
            s1 = torch.empty(Z * N, dtype=torch.float32, device=self.device) if pk else None
            ops.interp3(k1.view(Z, G, E), st.interp_index, st.interp_weight, up, rep, scale_out=s1,
                        ln=(w["mask_decoder.output_upscaling.1.weight"], w["mask_decoder.output_upscaling.1.bias"], cfg.ln_eps), act=ACT_GELU)
            if pk and self.fuse_hyper and (Z * N) % 256 == 0 and N % 32 == 0 and C <= 4:
                # second Linear + GELU + hyper-network products in one GEMM: each 64-column wave tile adds its part of the C dot products
                # per row, the [N, 256] activation is never written (mask_decoder.py:171-176)
                self._lin(U3, up, act=ACT_GELU, x_scale=s1, x_packed=True, hyper=(hyper, masks, N), no_store=True)
            else:
                u2 = self._lin(U3, up, act=ACT_GELU, x_scale=s1, x_packed=pk)
                ops.gemm_batched(hyper, u2, masks, C, N, E, E, E, N, C * E, N * E, C * N, Z)
        elif packed_interp and self.fuse_upscale_rows and (Z * N) % 128 == 0 and N % 32 == 0 and C <= 4 and (U3 + ".weight") in self.fw:
            # interpolation -> Linear -> LayerNorm -> GELU -> Linear -> GELU -> hyper-network products as THREE kernels: the interpolated
            # rows leave packed, the first GEMM's epilogue normalises / activates / re-packs whole rows (a wave owns a 256-column row;
            # packed against the LayerNorm's a-priori bound),
            # the second one's takes the C dot products per row: only the [Z, C, N] logits are written after the first GEMM.
            s_up = torch.empty(Z * N, dtype=torch.float32, device=self.device)
            ops.interp3(keys.view(Z, G, E), st.interp_index, st.interp_weight, up, rep, scale_out=s_up)
            u1 = torch.empty(Z * N, E, device=self.device)
            s1 = torch.empty(Z * N, dtype=torch.float32, device=self.device)
            self._lin(U0, up, x_scale=s_up, x_packed=True, out=u1, act=ACT_GELU, pack_out=(s1, 0.0, self.up_ln_bound),
                      row_ln=(w["mask_decoder.output_upscaling.1.weight"], w["mask_decoder.output_upscaling.1.bias"], cfg.ln_eps))
            self._lin(U3, u1, x_scale=s1, x_packed=True, act=ACT_GELU, hyper=(hyper, masks, N), no_store=True)
        else:
            if packed_interp:
                s_up = torch.empty(Z * N, dtype=torch.float32, device=self.device)
                ops.interp3(keys.view(Z, G, E), st.interp_index, st.interp_weight, up, rep, scale_out=s_up)
                u1 = self._lin(U0, up, x_scale=s_up, x_packed=True)
            else:
                ops.interp3(keys.view(Z, G, E), st.interp_index, st.interp_weight, up, rep)
                u1 = self._lin(U0, up)
            pk, rs = self._ln_feeds_gemm(u1, U3)
            self._ln("mask_decoder.output_upscaling.1", u1, cfg.ln_eps, act=ACT_GELU, out=u1, scale_out=rs, pack=pk)
            self._lin(U3, u1, act=ACT_GELU, out=up, x_scale=rs, x_packed=pk)
            ops.gemm_batched(hyper, up, masks, C, N, E, E, E, N, C * E, N * E, C * N, Z)
        return masks, sel

    @torch.no_grad()
    def decode(self, st: EncoderState, prompt_coords, prompt_labels, prompt_masks=None, multimask_output=True, use_center_idx=False):
        """Prompt encoders + MaskDecoder (pc_sam.py:62-87, mask_decoder.py:65-184) on a cached EncoderState.  use_center_idx: the mask
        encoder receives the groups' FPS indices (what PointCloudSAM.forward does, pc_sam.py:156; predict_masks does not, :65-70)."""
        with ops.gemm_mode(self.precision):
            return self._decode(st, prompt_coords, prompt_labels, prompt_masks, multimask_output, use_center_idx)

    def _decode(self, st, prompt_coords, prompt_labels, prompt_masks, multimask_output, use_center_idx=False):
        cfg, w, E = self.cfg, self.w, self.cfg.embed_dim
        B, N, _ = st.coords.shape
        G = st.centers.shape[1]
        prompt_coords = prompt_coords.to(self.device, torch.float32).contiguous()
        if prompt_coords.shape[:-1] != prompt_labels.shape:  # prompt_encoder.py:73
            raise AssertionError((tuple(prompt_coords.shape), tuple(prompt_labels.shape)))
        prompt_labels = prompt_labels.to(self.device, torch.int64).contiguous()
        Z, Pn, _ = prompt_coords.shape
        if Z % B != 0:
            raise ValueError(f"prompt batch {Z} is not a multiple of the cloud batch {B}")
        rep = Z // B
        nmt = cfg.num_mask_tokens
        T = 1 + nmt + Pn
        # tokens = [iou_token, mask_tokens, sparse prompt embeddings]   (mask_decoder.py:126-133)
        tokens = torch.empty(Z, T, E, device=self.device)
        ops.add_bcast(self.out_tokens, Z, None, tokens, Z, 1 + nmt, E, sa=0, so=T * E)
        ops.fourier_pe(prompt_coords, w["point_encoder.pe_layer.positional_encoding_gaussian_matrix"], tokens.view(-1)[(1 + nmt) * E:], Pn,
                       T * E, labels=prompt_labels, emb0=w["point_encoder.point_embeddings.0.weight"],
                       emb1=w["point_encoder.point_embeddings.1.weight"], flag=self._flag)
        # src = repeat(pc_embeddings) + dense prompt embedding        (prompt_encoder.py:118-133, mask_decoder.py:136-139)
        src = torch.empty(Z, G, E, device=self.device)
        if prompt_masks is None:
            ops.add_bcast(st.pc_embeddings, rep, w["mask_encoder.no_mask_embed.weight"], src, Z, G, E, sb=0, ldb=0)
        else:
            pm = prompt_masks.to(self.device, torch.float32).contiguous()
            if pm.shape != (Z, N):
                raise AssertionError((tuple(pm.shape), (Z, N)))
            dense = self._dense_prompt(st, pm, Z, N, use_center_idx)
            ops.add_bcast(st.pc_embeddings, rep, dense, src, Z, G, E)
        hs, keys = self._two_way(src.view(Z * G, E), st.pc_pe, tokens.view(Z * T, E), Z, G, T, rep)
        hs = hs.view(Z, T, E)
        assert hs.is_contiguous()
        # the hyper-networks (mask tokens) and the IoU head (token 0) read the same rows: one launch (mask_decoder.py:167-182)
        sel0, C = (1, nmt - 1) if multimask_output else (0, 1)
        hmw = self.hyper_mw[bool(multimask_output)]      # output width E (E // 2 in the hierarchical decoder)
        hyper, iou = torch.empty(Z, C, hmw.dout, device=self.device), torch.empty(Z, nmt, device=self.device)
        ops.mlp3_pair(hs[:, 1 + sel0, :], T * E, E, hmw, hyper, C * hmw.dout, hmw.dout, hs, T * E, 0, self.iou_mw, iou, nmt, 0, Z)
        masks, sel = self._masks_from_keys(st, keys, hs, Z, T, rep, multimask_output, hyper=hyper)
        return masks, iou[:, sel[0]:sel[-1] + 1]

    # ------------------------------------------------------------------------------------------ reference API
    def check_coordinate_range(self):
        """Raises the reference's ValueError (prompt_encoder.py:44-46) if any encoded coordinate left [-1, 1].
        The kernels record the violation in a device flag; reading it is the one host sync of a call."""
        if int(self._flag.item()) != 0:
            self._flag.zero_()
            raise ValueError("Input coordinates must be normalized to [-1, 1].")

    @torch.no_grad()
    def predict_masks(self, coords, features, prompt_coords, prompt_labels, prompt_masks=None, multimask_output=True, validate=True):
        """PointCloudSAM.predict_masks (pc_sam.py:37-88): (masks [B*M,C,N] logits, iou_preds [B*M,C])."""
        st = self.encode(coords, features)
        masks, iou = self.decode(st, prompt_coords, prompt_labels, prompt_masks, multimask_output)
        if validate:
            self.check_coordinate_range()
        return masks, iou

    @torch.no_grad()
    def click_session(self, st: "EncoderState", clicks, labels):
        """The interactive loop on a cached encoder state (pc_sam.py:139-194 with given clicks; BASELINE config #5): click t decodes with clicks
        0 .. t and the previous step's best mask as the mask prompt -- multimask output and the highest-IoU candidate after the first click,
        single-mask output afterwards.  clicks [B, T, 3], labels [B, T].  Returns the list of (masks, iou) per click.  No host
        synchronisation (the best candidate is selected on the device): the loop can be captured in a HIP graph."""
        N = st.coords.shape[1]
        outs, best = [], None
        for t in range(clicks.shape[1]):
            masks, iou = self.decode(st, clicks[:, : t + 1].contiguous(), labels[:, : t + 1].contiguous(), best, best is None)
            outs.append((masks, iou))
            best = torch.gather(masks, 1, iou.argmax(1).view(-1, 1, 1).expand(-1, 1, N))[:, 0] if t == 0 else masks[:, 0]
        return outs

    # ------------------------------------------------------------------------------------------ evaluation protocol
    @torch.no_grad()
    def sample_prompts(self, coords, gt_masks, pred_logits=None, is_eval=True):
        """sample_prompts_adapter (common.py:287-316): one simulated click per (cloud, mask).  Fixed sampler
        (sample_fixed_points, common.py:368-441) = the point of the error region farthest from its border; the random
        sampler (common.py:319-365) only when not evaluating and the batch IoU is exactly 1.
        coords [B,N,3], gt_masks [B,M,N] bool, pred_logits None | [B*M,N] -> (coords [B*M,1,3], labels [B*M,1] bool)."""
        B, M, N = gt_masks.shape
        Z = B * M
        gt = gt_masks.reshape(Z, N).to(self.device).to(torch.uint8).contiguous()
        if pred_logits is not None:
            pred_logits = pred_logits.to(self.device, torch.float32).contiguous()
            if not is_eval:
                pred = pred_logits > 0
                g = gt.bool()
                if float((g & pred).sum()) / max(float((g | pred).sum()), 1.0) >= 1.0:  # common.py:310-315
                    idx = torch.stack([(m.nonzero()[:, 0])[torch.randint(0, int(m.sum()), (1,), device=m.device)] for m in g]).view(Z)
                    return self._gather_clicks(coords, gt, idx, M)
        fn, fp = ops.error_regions(gt, pred_logits)
        pi, pd = ops.border_farthest(coords, fn)
        if pred_logits is None:  # from_error_region=True: mask = fn | fp = gt
            idx = pi
        else:
            # per mask: the false-negative candidate if it lies deeper inside its region than the false-positive one, else the false-positive
            # one, else (no error region at all: distance -1) a point of the ground truth (common.py:424-437) -- selected on the device, so the
            # only host synchronisation of an iteration is the emptiness check below
            ni, nd = ops.border_farthest(coords, fp)
            gi, _ = ops.border_farthest(coords, gt)
            take_p = pd > nd
            idx = torch.where(take_p, pi, torch.where(nd == -1, gi, ni))
        if bool((idx < 0).any()):
            raise ValueError("empty ground-truth / error region: no click can be sampled (the reference fails in torch.stack, common.py:439)")
        return self._gather_clicks(coords, gt, idx, M)

    @staticmethod
    def _gather_clicks(coords, gt, idx, M):
        Z = gt.shape[0]
        pts = coords.repeat_interleave(M, 0) if M > 1 else coords
        ar = torch.arange(Z, device=gt.device)
        return pts[ar, idx][:, None, :].contiguous(), gt[ar, idx].bool()[:, None]

    @torch.no_grad()
    def forward(self, coords, features, gt_masks, is_eval=False):
        """PointCloudSAM.forward in inference mode (pc_sam.py:90-196): encoder once, then ``prompt_iters`` iterations of
        {simulate a click from the current error region, decode with all clicks so far and the previous best mask}.
        Returns the reference's list of dicts.  (The two training-only mask-refinement iterations, pc_sam.py:128-134,
        apply only when ``self.training``; this object is inference-only.)"""
        coords, features = self._prep_inputs(coords, features)
        B, M, N = gt_masks.shape
        gt_masks = gt_masks.to(self.device)
        st = self.encode(coords, features)
        prompt_coords = coords.new_empty((B * M, 0, 3))
        prompt_labels = torch.empty((B * M, 0), dtype=torch.bool, device=self.device)
        prompt_masks, outputs = None, []
        for i in range(self.prompt_iters):
            nc, nl = self.sample_prompts(coords, gt_masks, prompt_masks, is_eval)
            prompt_coords = torch.cat([prompt_coords, nc], dim=1)
            prompt_labels = torch.cat([prompt_labels, nl], dim=1)
            masks, iou_preds = self.decode(st, prompt_coords, prompt_labels, prompt_masks, multimask_output=(i == 0), use_center_idx=True)
            if i == 0:  # pc_sam.py:176-180
                max_iou_pred_ind = torch.argmax(iou_preds, dim=1)
                prompt_masks = torch.gather(masks, 1, max_iou_pred_ind.view(-1, 1, 1).expand(-1, 1, N))[:, 0].contiguous()
            else:       # pc_sam.py:181-183
                max_iou_pred_ind = 0
                prompt_masks = masks[:, 0].contiguous()
            outputs.append(dict(prompt_coords=prompt_coords, prompt_labels=prompt_labels, masks=masks, iou_preds=iou_preds,
                                max_iou_pred_ind=max_iou_pred_ind, prompt_masks=prompt_masks))
        self.check_coordinate_range()
        return outputs

    def __call__(self, *args, **kwargs):
        """``model(coords, features, gt_masks, is_eval=True)`` like the reference's nn.Module (evaluation/eval_kitti.py:363)."""
        return self.forward(*args, **kwargs)

    def eval(self):
        return self

    def cuda(self):
        return self


class BatchPipeline:
    """Software pipeline over a stream of independent batches.

    Stage 1, the coordinate-only tokenizer (FPS, kNN, 3-NN; a few latency-bound workgroups), of batch i+1 runs on its own
    high-priority HIP stream while stage 2, the dense work (mini-PointNet, ViT, decoder; every CU), of batch i runs.
    ``dense_streams = 1``: ``submit`` enqueues stage 1, ``next`` enqueues stage 2 of the oldest submitted batch on the caller's
    stream.  ``dense_streams = S > 1``: ``submit`` enqueues BOTH stages, stage 2 on dense stream ``i % S``, so that S batches are
    in flight and one batch's kernel tails (a 768-tile GEMM on 512 workgroup slots, single-workgroup epilogues, the small decoder
    kernels) are filled by the other's kernels; ``next`` only makes the caller's stream wait for the batch.  Results are identical
    to ``predict_masks`` (same kernels, same order per batch); only the interleaving on the device changes."""

    def __init__(self, model: PointCloudSAM, dense_streams: int = 1):
        from collections import deque
        self.model = model
        self.tok_stream, self.dense = pipeline_streams(model.device, dense_streams if dense_streams > 1 else 0)
        self.count = 0
        self.queue = deque()

    @property
    def depth(self) -> int:
        """How many batches to keep submitted ahead of ``next``."""
        return max(1, len(self.dense))

    @torch.no_grad()
    def submit(self, coords, features, prompt_coords, prompt_labels, prompt_masks=None, multimask_output=True):
        m = self.model
        coords, features = m._prep_inputs(coords, features)
        main = torch.cuda.current_stream(m.device)
        self.tok_stream.wait_stream(main)  # inputs may have been produced on the caller's stream
        with torch.cuda.stream(self.tok_stream):
            tok = m.tokenize(coords, with_interp=True)
            ready = torch.cuda.Event()
            ready.record(self.tok_stream)
        if not self.dense:
            for t in tok.tensors():
                t.record_stream(main)  # allocated on tok_stream, consumed on the caller's stream
            self.queue.append((tok, ready, coords, features, prompt_coords, prompt_labels, prompt_masks, multimask_output))
            return
        ds = self.dense[self.count % len(self.dense)]
        self.count += 1
        for t in tok.tensors():
            t.record_stream(ds)
        ds.wait_stream(main)
        ds.wait_event(ready)
        with torch.cuda.stream(ds), ops.attention_keysplit(1):      # several batches in flight: the other streams fill idle CUs, no key split
            st = m.encode(coords, features, tok)
            out = m.decode(st, prompt_coords, prompt_labels, prompt_masks, multimask_output)
            done = torch.cuda.Event()
            done.record(ds)
        for t in out:
            t.record_stream(main)
        # The inputs were allocated on the caller's stream but are read on tok_stream and ds: hold references until next() has made
        # the caller's stream wait for `done`, so the caching allocator cannot hand their memory to a later allocation on the
        # caller's stream while these kernels still read it (and mark them, for callers that drop them right after submit()).
        keep = tuple(t for t in (coords, features, prompt_coords, prompt_labels, prompt_masks) if isinstance(t, torch.Tensor) and t.is_cuda)
        for t in keep:
            t.record_stream(self.tok_stream)
            t.record_stream(ds)
        self.queue.append((out, done, keep))

    @torch.no_grad()
    def next(self):
        if self.dense:
            out, done, keep = self.queue.popleft()
            torch.cuda.current_stream(self.model.device).wait_event(done)
            del keep
            return out
        tok, ready, coords, features, pc, pl, pm, mm = self.queue.popleft()
        torch.cuda.current_stream(self.model.device).wait_event(ready)
        st = self.model.encode(coords, features, tok)
        return self.model.decode(st, pc, pl, pm, mm)

    def __len__(self):
        return len(self.queue)


class GraphPipeline:
    """BatchPipeline with both stages of a batch captured in HIP graphs (static shapes): per step the host copies the inputs into the
    slot's static buffers and replays two graphs -- the tokenizer graph on the high-priority side stream, the dense graph (encode +
    decode, ~380 kernel launches) on the slot's dense stream -- instead of issuing every launch (0.1 ms instead of ~4.5 ms of host time
    per step).  Up to `slots` batches are in flight, each in its own set of static buffers, their dense stages alternating over
    `dense_streams` streams; results are bit-identical to the eager path (same kernels, same order per batch).  Protocol: next() before
    the slot is submitted again -- i.e. `for k: out = next(); submit(batch k + slots)` after `slots` initial submits (submit() raises
    if the slot's previous batch has not been handed out).  The outputs returned by next() are the slot's static tensors: consume or
    copy them before `slots` further submits.  Coordinates outside [-1, 1] are still recorded in the model's device flag."""

    def __init__(self, model: PointCloudSAM, coords, features, prompt_coords, prompt_labels, prompt_masks=None, multimask_output=True, slots: int = 3,
                 dense_streams: int = 2, session: bool = False, streams=None):
        """session=True: prompt_coords [B, T, 3] / prompt_labels [B, T] are the T clicks of an interactive session; the dense graph is encode +
        PointCloudSAM.click_session (T decodes on the cached state) and next() returns the LAST click's (masks, iou)."""
        from collections import deque
        # Every slot owns an arrival-counter block (ops.new_counters) and captures its graphs with it (ops.use_counters): kernels with in-kernel fix-ups
        # (split-K GEMMs, key-split attention, the decoder's fused Linear + LayerNorm) count in through the SLOT's block, so any two slots' graphs may be in
        # flight together, on whatever streams.  (Until round 5 the counters belonged to the (device, stream) inside the library: with 3 slots on 2 streams
        # every third batch replayed beside a graph holding the same counters -- garbage partial sums; round 5 bound slots to their capture streams, round 6
        # moved the state to the caller.)  Slot s still runs on stream s % streams and the number of slots is a multiple of the number of streams, so
        # that consecutive batches alternate streams.
        nstreams = max(1, min(dense_streams, max(1, slots)))
        self.model, self.depth, self.count = model, (max(1, slots) + nstreams - 1) // nstreams * nstreams, 0
        self.session = bool(session)
        if session and prompt_masks is not None:
            raise ValueError("GraphPipeline(session=True) feeds each click's best mask forward itself: prompt_masks must be None")
        dense_body = (lambda st_, tok_: model.click_session(model.encode(st_.coords, st_.features, tok_), st_.pc, st_.pl)[-1]) if session else \
                     (lambda st_, tok_: model.decode(model.encode(st_.coords, st_.features, tok_), st_.pc, st_.pl, st_.pm, multimask_output))
        self.queue = deque()
        dev = model.device
        self.tok_stream, self.dense = pipeline_streams(dev, nstreams) if streams is None else (streams[0], list(streams[1])[:nstreams])      # streams=(tok, [dense..]): the caller's own
        self.multimask = multimask_output
        conv = lambda t, dt: None if t is None else t.to(dev, dt).contiguous().clone()
        self.slots = []
        main = torch.cuda.current_stream(dev)
        for s in range(self.depth):
            st = SimpleNamespace(coords=conv(coords, torch.float32), features=conv(features, torch.float32), pc=conv(prompt_coords, torch.float32),
                                 pl=conv(prompt_labels, torch.int64), pm=conv(prompt_masks, torch.float32))
            ds = st.ds = self.dense[s % len(self.dense)]
            st.busy = False
            st.counters = ops.new_counters(dev)
            # one eager pass first: every kernel's one-time set-up (LDS attributes, library loading) must not happen inside a capture
            self.tok_stream.wait_stream(main); ds.wait_stream(main)
            with torch.cuda.stream(self.tok_stream), ops.use_counters(st.counters):
                tok = model.tokenize(st.coords, with_interp=True)
            ds.wait_stream(self.tok_stream)
            # the key split of the single-cloud attention serves the latency of ONE stream; with several dense streams the graphs are captured without it
            keysplit = ops.attention_keysplit(1 if len(self.dense) > 1 else 4)
            with torch.cuda.stream(ds), keysplit, ops.use_counters(st.counters):
                dense_body(st, tok)
            torch.cuda.synchronize(dev)
            st.g_tok = torch.cuda.CUDAGraph()
            with torch.cuda.graph(st.g_tok, stream=self.tok_stream), ops.use_counters(st.counters):
                st.tok = model.tokenize(st.coords, with_interp=True)
            st.g_dense = torch.cuda.CUDAGraph()
            with torch.cuda.graph(st.g_dense, stream=ds), keysplit, ops.use_counters(st.counters):
                st.out = dense_body(st, st.tok)
            st.tok_done, st.done = torch.cuda.Event(), torch.cuda.Event()
            self.slots.append(st)
        # what the graphs were captured for: submit() refuses anything else (a replay would silently compute on stale or mis-shaped buffers)
        g = model.pc_encoder.patch_embed.grouper
        self.captured = dict(num_groups=int(g.num_groups), group_size=int(g.group_size), precision=model.precision)
        torch.cuda.synchronize(dev)

    @torch.no_grad()
    def submit(self, coords, features, prompt_coords, prompt_labels, prompt_masks=None, multimask_output=True):
        if multimask_output != self.multimask:
            raise ValueError("GraphPipeline was captured for a fixed multimask_output")
        g = self.model.pc_encoder.patch_embed.grouper
        now = dict(num_groups=int(g.num_groups), group_size=int(g.group_size), precision=self.model.precision)
        if now != self.captured:
            raise ValueError(f"GraphPipeline was captured for {self.captured}, the model is now set to {now}: build a new pipeline")
        st = self.slots[self.count % self.depth]
        for name, dst, src in (("coords", st.coords, coords), ("features", st.features, features), ("prompt_coords", st.pc, prompt_coords),
                               ("prompt_labels", st.pl, prompt_labels), ("prompt_masks", st.pm, prompt_masks)):
            if (dst is None) != (src is None):
                raise ValueError(f"GraphPipeline: {name} was {'absent' if dst is None else 'present'} at capture and is {'absent' if src is None else 'present'} now "
                                 "(the captured graph has a fixed set of inputs)")
            if dst is not None and tuple(src.shape) != tuple(dst.shape):
                raise ValueError(f"GraphPipeline: {name} has shape {tuple(src.shape)}, the graphs were captured for {tuple(dst.shape)}")
        if st.busy:
            raise RuntimeError("GraphPipeline: the slot's previous batch has not been taken with next() yet (at most `slots` batches in flight)")
        ds = st.ds      # the stream the slot's dense graph was captured on
        self.count += 1
        st.busy = True
        main = torch.cuda.current_stream(self.model.device)
        # the slot's previous results must have been consumed (next() made `main` wait for them); its static inputs are rewritten on
        # `main`, ordered after that wait and before the replays below
        for dst, src in ((st.coords, coords), (st.features, features), (st.pc, prompt_coords), (st.pl, prompt_labels), (st.pm, prompt_masks)):
            if dst is not None and src is not None and dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.tok_stream.wait_stream(main)
        with torch.cuda.stream(self.tok_stream):
            st.g_tok.replay()
            st.tok_done.record(self.tok_stream)
        ds.wait_stream(main)
        ds.wait_event(st.tok_done)
        with torch.cuda.stream(ds):
            st.g_dense.replay()
            st.done.record(ds)
        self.queue.append(st)

    @torch.no_grad()
    def next(self):
        st = self.queue.popleft()
        torch.cuda.current_stream(self.model.device).wait_event(st.done)
        st.busy = False
        return st.out

    def __len__(self):
        return len(self.queue)
