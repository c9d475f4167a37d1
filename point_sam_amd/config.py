"""Architecture hyper-parameters of the Point-SAM hot path, restated as plain dataclasses.

Source of truth in the reference (hydra YAML, not importable here because hydra is absent):
  configs/model/base.yaml:1-26, configs/model/default.yaml:1-27 (= "large"), configs/model/giant.yaml:1-26.
The ViT dimensions come from timm's model registry (eva02_base_patch14_448, eva02_large_patch14_448,
eva_giant_patch14_560); timm is not vendored in the reference, so these numbers are *our spec* of that
dependency.
"""
from dataclasses import dataclass, field, replace


@dataclass(frozen=True)
class ViTConfig:
    name: str
    dim: int            # transformer width D
    depth: int
    heads: int
    mlp_hidden: int     # SwiGLU hidden (eva02) or GELU-MLP hidden (eva_giant)
    swiglu: bool        # eva02: separate q/k/v proj + SwiGLU with inner LayerNorm; eva_giant: fused qkv + GELU MLP
    ln_eps: float = 1e-6

    @property
    def head_dim(self) -> int:
        return self.dim // self.heads


@dataclass(frozen=True)
class ModelConfig:
    vit: ViTConfig
    num_groups: int = 512        # G, KNNGrouper.num_groups  (pc_encoder.py:26-31)
    group_size: int = 64         # K, KNNGrouper.group_size
    in_channels: int = 6         # xyz-relative + rgb        (configs/model/default.yaml:6)
    patch_out: int = 512         # PatchEmbed.out_channels   (configs/model/default.yaml:7)
    patch_hidden: tuple = (128, 512)  # pc_encoder.py:34
    embed_dim: int = 256         # SAM embed dim             (configs/model/default.yaml:14)
    dec_depth: int = 2           # TwoWayTransformer.depth   (configs/model/default.yaml:23)
    dec_heads: int = 8
    dec_mlp: int = 2048
    dec_downsample: int = 2      # transformer.py:23
    num_multimask: int = 3       # mask_decoder.py:26
    prompt_iters: int = 5
    ln_eps: float = 1e-5         # torch.nn.LayerNorm default used by every non-timm LayerNorm
    radius: float = None         # KNNGrouper.radius (configs/model/enc_with_radius.yaml: 0.1); None = off
    mask_radius: float = "same"  # MaskEncoder.radius (prompt_encoder.py:78-93): an independent option in the reference; "same" = radius
    centralize_features: bool = False   # KNNGrouper.centralize_features (common.py:116-118): + (neighbour - centre) features, in_channels 3 + 2*3
    mask_centralize_features: bool = False   # MaskEncoder.centralize_features (common.py:183-186): needs center_idx, i.e. only the
                                             # forward() protocol passes it (pc_sam.py:151-157); in_channels 3 + 2*1
    # model variants (configs/model/voronoi.yaml, hier.yaml): "knn" = PointCloudSAM; "voronoi" = PointCloudSAMNN (every point belongs to its
    # nearest FPS centre: NNGrouper / PatchEmbedNN / MaskEncoderNN, pc_encoder.py:165-198, prompt_encoder.py:255-300, pc_sam.py:199-374);
    # "hier" = PointCloudSAMHier (two kNN levels: PatchEmbedHier / MaskEncoderHier / MaskDecoderHier, pc_encoder.py:201-239,
    # prompt_encoder.py:136-183, mask_decoder.py:214-370, pc_sam.py:377-496)
    variant: str = "knn"
    nn_hidden: int = 256             # PatchEmbedNN.hidden_dim   (voronoi.yaml:7)
    nn_mask_hidden: int = 1024       # MaskEncoderNN first_nn / ResMlp width: fixed in the reference (prompt_encoder.py:259-260)
    nn_mask_scatter_fixed: bool = False   # MaskEncoderNN scatters every mask set into the FIRST set's cells (no batch offset, prompt_encoder.py:291-297):
                                          # False = as the reference does it (results must equal the reference's); True = each set into its own cells
    hier_groups: tuple = (2048, 512)     # PatchEmbedHier.num_patches (hier.yaml:8)
    hier_sizes: tuple = (32, 32)         # PatchEmbedHier.patch_size  (hier.yaml:9)
    hier_radius: tuple = (0.05, 0.1)     # hier.yaml:10 (also MaskEncoderHier.radius, :18); None = no normalisation
    hier_dim1: int = 128                 # level-1 embedding width (pc_encoder.py:222, mask_decoder.py:226 encoder_dim)

    @property
    def mask_encoder_radius(self):
        return self.radius if self.mask_radius == "same" else self.mask_radius

    @property
    def patch_in_channels(self) -> int:
        return 3 + (self.in_channels - 3) * (2 if self.centralize_features else 1)

    @property
    def mask_in_channels(self) -> int:
        return 3 + (2 if self.mask_centralize_features else 1)

    @property
    def num_mask_tokens(self) -> int:
        return self.num_multimask + 1

    @property
    def num_tokens(self) -> int:
        """Patch tokens the transformer sees."""
        return self.hier_groups[1] if self.variant == "hier" else self.num_groups

    def with_groups(self, num_groups: int, group_size: int) -> "ModelConfig":
        return replace(self, num_groups=num_groups, group_size=group_size)


VIT_BASE = ViTConfig("eva02_base_patch14_448", 768, 12, 12, 2048, True)
VIT_LARGE = ViTConfig("eva02_large_patch14_448", 1024, 24, 16, 2730, True)
VIT_GIANT = ViTConfig("eva_giant_patch14_560", 1408, 40, 16, 6144, False)
# Tiny transformers used only by tests / golden fixtures (same block arithmetic, small dims).
VIT_TINY_SWIGLU = ViTConfig("tiny_eva02", 64, 2, 2, int(64 * 8 / 3), True)       # hidden 170: exercises padding
VIT_TINY_GELU = ViTConfig("tiny_eva_giant", 96, 2, 4, 256, False)                # head_dim 24

CONFIGS = {
    # configs/model/base.yaml: num_patches 512, patch_size 64, prompt_iters 10
    "base": ModelConfig(VIT_BASE, 512, 64, prompt_iters=10),
    # configs/model/default.yaml (selected by configs/large.yaml:2): num_patches 1024, patch_size 256
    "large": ModelConfig(VIT_LARGE, 1024, 256, prompt_iters=5),
    # configs/model/giant.yaml
    "giant": ModelConfig(VIT_GIANT, 512, 64, prompt_iters=10),
    "tiny": ModelConfig(VIT_TINY_SWIGLU, 32, 16, prompt_iters=3),
    "tiny_gelu": ModelConfig(VIT_TINY_GELU, 32, 16, prompt_iters=3),
    # configs/model/enc_with_radius.yaml on top of default.yaml / on the tiny test transformer
    "large_radius": ModelConfig(VIT_LARGE, 1024, 256, prompt_iters=5, radius=0.1),
    "tiny_radius": ModelConfig(VIT_TINY_SWIGLU, 32, 16, prompt_iters=3, radius=0.1),
    # centralised group features (KNNGrouper.centralize_features) + different grouper / mask-encoder radii on the tiny test transformer
    "tiny_central": ModelConfig(VIT_TINY_SWIGLU, 32, 16, prompt_iters=3, radius=0.2, mask_radius=0.1, centralize_features=True),
    # configs/model/voronoi.yaml (PointCloudSAMNN, 7 input channels: unit offset 3 + distance 1 + rgb 3) and hier.yaml (PointCloudSAMHier)
    "voronoi": ModelConfig(VIT_LARGE, 1024, 1, in_channels=7, prompt_iters=5, variant="voronoi"),
    "hier": ModelConfig(VIT_LARGE, 512, 32, prompt_iters=8, variant="hier"),
    "tiny_voronoi": ModelConfig(VIT_TINY_SWIGLU, 32, 1, in_channels=7, prompt_iters=3, variant="voronoi", nn_hidden=64),
    "tiny_hier": ModelConfig(VIT_TINY_SWIGLU, 16, 8, prompt_iters=3, variant="hier", hier_groups=(64, 16), hier_sizes=(8, 8), hier_radius=(0.2, 0.4)),
}


def get_config(name: str, num_groups: int = None, group_size: int = None) -> ModelConfig:
    cfg = CONFIGS[name]
    if num_groups is not None or group_size is not None:
        cfg = cfg.with_groups(num_groups or cfg.num_groups, group_size or cfg.group_size)
    return cfg
