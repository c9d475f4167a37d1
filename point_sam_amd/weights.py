"""Weight container keyed by the reference state-dict names (the checkpoint ABI).

The reference loads weights with ``safetensors.torch.load_model(model, path)`` (evaluation/inference.py:46,
demo/app.py:67), so the *names* of ``PointCloudSAM.state_dict()`` are the contract.  Names for our own
modules follow pc_sam/model/{pc_sam,pc_encoder,prompt_encoder,mask_decoder,transformer,common}.py; names under
``pc_encoder.transformer.`` follow timm's Eva (not vendored in the reference; our spec).

No network here => no pretrained checkpoint: ``random_state_dict`` creates a seeded random model that the
oracle, the golden-fixture generator (which loads it *into the reference's own modules*, strict=True) and
the HIP path all share bit-for-bit.
"""
from collections import OrderedDict
import math

import torch

from .config import ModelConfig

# timm parameters that exist in real checkpoints but are never touched by the hot path
# (PointCloudEncoder.forward only uses .blocks/.norm/.fc_norm: pc_encoder.py:136-142).
UNUSED_TIMM_PREFIXES = (
    "pc_encoder.transformer.cls_token",
    "pc_encoder.transformer.pos_embed",
    "pc_encoder.transformer.patch_embed.",
    "pc_encoder.transformer.head.",
    "pc_encoder.transformer.rope.",
    "pc_encoder.transformer.norm.",  # Identity when fc_norm is used; tolerate if present
)


def expected_shapes(cfg: ModelConfig) -> "OrderedDict[str, tuple]":
    """name -> shape for every tensor the hot path reads, in a fixed order."""
    s = OrderedDict()
    D = cfg.vit.dim
    E = cfg.embed_dim
    h0, h1 = cfg.patch_hidden

    def linear(name, fin, fout, bias=True):
        s[name + ".weight"] = (fout, fin)
        if bias:
            s[name + ".bias"] = (fout,)

    def norm(name, n):
        s[name + ".weight"] = (n,)
        s[name + ".bias"] = (n,)

    def patch_encoder(prefix, cin, cout, hidden=None):  # common.py:477-497
        a, b = hidden or (h0, h1)
        linear(prefix + ".conv1.0", cin, a)
        norm(prefix + ".conv1.1", a)
        linear(prefix + ".conv1.3", a, a)
        linear(prefix + ".conv2.0", 2 * a, b)
        norm(prefix + ".conv2.1", b)
        linear(prefix + ".conv2.3", b, cout)

    # --- pc_encoder (pc_encoder.py:84-116)
    if cfg.variant == "voronoi":      # PatchEmbedNN (pc_encoder.py:165-198): in_proj, 3 + 3 pre-LN residual MLP blocks (:147-162), norm, out_proj
        Hn = cfg.nn_hidden
        linear("pc_encoder.patch_embed.in_proj", cfg.in_channels, Hn)
        for grp in ("blocks1", "blocks2"):
            for i in range(3):
                q = f"pc_encoder.patch_embed.{grp}.{i}"
                linear(q + ".mlp.0", Hn, Hn)
                norm(q + ".mlp.2", Hn)
                linear(q + ".mlp.3", Hn, Hn)
                norm(q + ".norm", Hn)
        norm("pc_encoder.patch_embed.norm", Hn)
        linear("pc_encoder.patch_embed.out_proj", Hn, cfg.patch_out)
    elif cfg.variant == "hier":       # PatchEmbedHier (pc_encoder.py:201-239)
        patch_encoder("pc_encoder.patch_embed.patch_encoder1", cfg.in_channels, cfg.hier_dim1, (64, 128))
        patch_encoder("pc_encoder.patch_embed.patch_encoder2", cfg.hier_dim1 + 3, cfg.patch_out, (128, 256))
    else:
        patch_encoder("pc_encoder.patch_embed.patch_encoder", cfg.patch_in_channels, cfg.patch_out)
    linear("pc_encoder.patch_proj", cfg.patch_out, D)
    linear("pc_encoder.pos_embed.0", 3, 128)
    linear("pc_encoder.pos_embed.2", 128, D)
    for i in range(cfg.vit.depth):
        p = f"pc_encoder.transformer.blocks.{i}"
        norm(p + ".norm1", D)
        if cfg.vit.swiglu:
            linear(p + ".attn.q_proj", D, D)
            linear(p + ".attn.k_proj", D, D, bias=False)
            linear(p + ".attn.v_proj", D, D)
        else:
            s[p + ".attn.qkv.weight"] = (3 * D, D)
            s[p + ".attn.q_bias"] = (D,)
            s[p + ".attn.v_bias"] = (D,)
        linear(p + ".attn.proj", D, D)
        norm(p + ".norm2", D)
        H = cfg.vit.mlp_hidden
        if cfg.vit.swiglu:
            linear(p + ".mlp.fc1_g", D, H)
            linear(p + ".mlp.fc1_x", D, H)
            norm(p + ".mlp.norm", H)
            linear(p + ".mlp.fc2", H, D)
        else:
            linear(p + ".mlp.fc1", D, H)
            linear(p + ".mlp.fc2", H, D)
    norm("pc_encoder.transformer.fc_norm", D)
    linear("pc_encoder.out_proj", D, E)
    # --- prompt encoders (prompt_encoder.py:13-95)
    s["point_encoder.pe_layer.positional_encoding_gaussian_matrix"] = (3, E // 2)
    s["point_encoder.point_embeddings.0.weight"] = (1, E)
    s["point_encoder.point_embeddings.1.weight"] = (1, E)
    if cfg.variant == "voronoi":      # MaskEncoderNN (prompt_encoder.py:255-300) with ResMlp (:186-211)
        Hm = cfg.nn_mask_hidden
        linear("mask_encoder.first_nn", 5, Hm)
        linear("mask_encoder.second_nn.mlp.0", Hm, Hm)
        norm("mask_encoder.second_nn.mlp.1", Hm)
        for i in (3, 4, 5):
            linear(f"mask_encoder.second_nn.mlp.{i}.mlp.0", Hm, Hm)
            norm(f"mask_encoder.second_nn.mlp.{i}.mlp.1", Hm)
        linear("mask_encoder.second_nn.mlp.6", Hm, E)
    elif cfg.variant == "hier":       # MaskEncoderHier (prompt_encoder.py:136-183)
        patch_encoder("mask_encoder.patch_encoder1", 4, cfg.hier_dim1, (64, 128))
        patch_encoder("mask_encoder.patch_encoder2", cfg.hier_dim1 + 3, E, (128, 256))
    else:
        patch_encoder("mask_encoder.patch_encoder", cfg.mask_in_channels, E)
    s["mask_encoder.no_mask_embed.weight"] = (1, E)
    # --- mask decoder (mask_decoder.py:21-63, transformer.py:15-59,103-142,179-202,240-249)
    s["mask_decoder.iou_token.weight"] = (1, E)
    s["mask_decoder.mask_tokens.weight"] = (cfg.num_mask_tokens, E)

    def attention(prefix, downsample):
        inner = E // downsample
        linear(prefix + ".q_proj", E, inner)
        linear(prefix + ".k_proj", E, inner)
        linear(prefix + ".v_proj", E, inner)
        linear(prefix + ".out_proj", inner, E)

    for i in range(cfg.dec_depth):
        p = f"mask_decoder.transformer.layers.{i}"
        attention(p + ".self_attn", 1)
        norm(p + ".norm1", E)
        attention(p + ".cross_attn_token_to_image", cfg.dec_downsample)
        norm(p + ".norm2", E)
        linear(p + ".mlp.lin1", E, cfg.dec_mlp)
        linear(p + ".mlp.lin2", cfg.dec_mlp, E)
        norm(p + ".norm3", E)
        norm(p + ".norm4", E)
        attention(p + ".cross_attn_image_to_token", cfg.dec_downsample)
    attention("mask_decoder.transformer.final_attn_token_to_image", cfg.dec_downsample)
    norm("mask_decoder.transformer.norm_final_attn", E)
    hier = cfg.variant == "hier"
    for i in range(cfg.num_mask_tokens):
        for j in range(3):      # MaskDecoderHier: MLP(E, E, E // 2, 3) (mask_decoder.py:240-245)
            linear(f"mask_decoder.output_hypernetworks_mlps.{i}.layers.{j}", E, E // 2 if (hier and j == 2) else E)
    if hier:                      # mask_decoder.py:246-258
        linear("mask_decoder.output_upscaling2.0", E + cfg.hier_dim1, E)
        norm("mask_decoder.output_upscaling2.1", E)
        linear("mask_decoder.output_upscaling2.3", E, E)
        linear("mask_decoder.output_upscaling1.0", E, E // 2)
        norm("mask_decoder.output_upscaling1.1", E // 2)
        linear("mask_decoder.output_upscaling1.3", E // 2, E // 2)
    else:
        linear("mask_decoder.output_upscaling.0", E, E)
        norm("mask_decoder.output_upscaling.1", E)
        linear("mask_decoder.output_upscaling.3", E, E)
    linear("mask_decoder.iou_prediction_head.layers.0", E, E)
    linear("mask_decoder.iou_prediction_head.layers.1", E, E)
    linear("mask_decoder.iou_prediction_head.layers.2", E, cfg.num_mask_tokens)
    return s


_EMBEDDING_SUFFIXES = (
    "positional_encoding_gaussian_matrix",
    "point_embeddings.0.weight",
    "point_embeddings.1.weight",
    "no_mask_embed.weight",
    "iou_token.weight",
    "mask_tokens.weight",
)


def _is_norm_key(name: str) -> bool:
    parts = name.split(".")
    leaf_parent = parts[-2]
    return (
        leaf_parent.startswith("norm")
        or leaf_parent == "fc_norm"
        or name.endswith(("conv1.1.weight", "conv1.1.bias", "conv2.1.weight", "conv2.1.bias"))
        or ".output_upscaling.1." in name or ".output_upscaling1.1." in name or ".output_upscaling2.1." in name
        or (".patch_embed.blocks" in name and ".mlp.2." in name)                       # Block: Linear, GELU, LayerNorm, Linear (pc_encoder.py:152-157)
        or (name.startswith("mask_encoder.second_nn.mlp.") and _resmlp_norm(name))
    )


def _resmlp_norm(name: str) -> bool:
    """LayerNorms of MaskEncoderNN.second_nn (ResMlp, prompt_encoder.py:186-211): mlp.1 and mlp.{3,4,5}.mlp.1."""
    parts = name.split(".")
    return (len(parts) == 5 and parts[3] == "1") or (len(parts) == 7 and parts[5] == "1")


def random_state_dict(cfg: ModelConfig, seed: int = 42) -> "OrderedDict[str, torch.Tensor]":
    """Seeded random fp32 weights.  Linear: N(0, 1/fan_in); biases N(0, 0.02^2); LayerNorm weight
    1+0.1*N, bias 0.1*N (non-trivial affine so tests see it); embeddings / PE matrix N(0,1)."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for name, shape in expected_shapes(cfg).items():
        if _is_norm_key(name):
            t = torch.randn(shape, generator=g) * 0.1
            if name.endswith(".weight"):
                t = t + 1.0
        elif name.endswith(_EMBEDDING_SUFFIXES):
            t = torch.randn(shape, generator=g)  # nn.Embedding tables / Gaussian PE matrix
        elif name.endswith(".weight") and len(shape) == 2:
            t = torch.randn(shape, generator=g) / math.sqrt(shape[1])
        else:  # biases (incl. q_bias / v_bias)
            t = torch.randn(shape, generator=g) * 0.02
        sd[name] = t.contiguous()
    return sd


def check_state_dict(cfg: ModelConfig, sd) -> None:
    """Strict check of names and shapes; unknown timm leftovers are tolerated (listed above)."""
    exp = expected_shapes(cfg)
    missing = [k for k in exp if k not in sd]
    if missing:
        raise KeyError(f"state dict is missing {len(missing)} tensors, e.g. {missing[:5]}")
    for k, shp in exp.items():
        if tuple(sd[k].shape) != tuple(shp):
            raise ValueError(f"{k}: expected shape {shp}, got {tuple(sd[k].shape)}")
    extra = [k for k in sd if k not in exp and not k.startswith(UNUSED_TIMM_PREFIXES)]
    if extra:
        raise KeyError(f"state dict has {len(extra)} unexpected tensors, e.g. {extra[:5]}")


def load_safetensors(cfg: ModelConfig, path: str):
    """Reads a reference checkpoint (``model.safetensors``) into a name->fp32 tensor dict."""
    from safetensors.torch import load_file

    sd = {k: v.float() for k, v in load_file(path).items()}
    check_state_dict(cfg, sd)
    return sd


def state_dict_checksum(sd) -> float:
    """Cheap order-dependent fingerprint used by golden fixtures to detect RNG drift."""
    acc = 0.0
    for i, (k, v) in enumerate(sd.items()):
        acc += float(v.double().sum()) * (1.0 + (i % 7)) + float(v.double().abs().sum())
    return acc
