"""The predictor the reference's demo calls but never defines (demo/app.py:198-205; README.md:41 points at a
sibling repo): ``set_pointcloud`` / ``set_prompts`` / ``predict_masks``.

Contract inferred from the call sites:
    sam.set_pointcloud(pc_xyz[B,N,3], pc_rgb[B,N,3])
    mask, scores, logits = sam.predict_masks(prompt_points[B,P,3], prompt_labels[B,P], prompt_mask|None, multimask: bool)
    prompt_mask = logits[0][argmax(scores[0])][None]; segment = mask[0][argmax(scores[0])] > 0

The encoder state is cached per cloud (the demo calls set_pointcloud on EVERY click, app.py:199), so clicks after
the first run the decoder only -- BASELINE config #5's "encoder cached, decoder-only loop".
"""
from typing import Optional

import torch

from .config import ModelConfig, get_config
from .model import EncoderState, PointCloudSAM
from .weights import load_safetensors, random_state_dict


class PointSAMPredictor:
    def __init__(self, model: PointCloudSAM):
        self.model = model
        self._state: Optional[EncoderState] = None
        self._key = None
        self._prompts = None

    @classmethod
    def from_config(cls, name: str, ckpt_path: str = None, num_groups: int = None, group_size: int = None, seed: int = 42,
                    device="cuda", precision: str = "f16x3") -> "PointSAMPredictor":
        cfg: ModelConfig = get_config(name, num_groups, group_size)
        sd = load_safetensors(cfg, ckpt_path) if ckpt_path else random_state_dict(cfg, seed)
        from .variants import build_model      # cfg.variant: PointCloudSAM | PointCloudSAMNN (voronoi) | PointCloudSAMHier
        return cls(build_model(cfg, sd, device, precision=precision))

    # -- state ---------------------------------------------------------------------------------------------
    @staticmethod
    def _cloud_key(xyz, rgb):
        return (xyz.data_ptr(), rgb.data_ptr(), tuple(xyz.shape), xyz._version, rgb._version)

    @torch.no_grad()
    def set_pointcloud(self, xyz: torch.Tensor, rgb: torch.Tensor) -> None:
        """Runs the encoder unless this exact cloud tensor is already cached."""
        if xyz.dim() == 2:
            xyz, rgb = xyz[None], rgb[None]
        g = self.model.pc_encoder.patch_embed.grouper
        key = self._cloud_key(xyz, rgb) + (g.num_groups, g.group_size)
        if key != self._key:
            self._state = self.model.encode(xyz, rgb)
            self._key = key
            self._keepalive = (xyz, rgb)  # the cache key uses data_ptr: keep the tensors alive

    def set_prompts(self, prompt_points, prompt_labels, prompt_mask=None) -> None:
        self._prompts = (prompt_points, prompt_labels, prompt_mask)

    @torch.no_grad()
    def predict_masks(self, prompt_points=None, prompt_labels=None, prompt_mask=None, multimask_output: bool = True):
        """-> (masks [BM,C,N] logits, scores [BM,C], logits [BM,C,N]); masks and logits are the same tensor, the
        caller thresholds at 0 (demo/app.py:203-205)."""
        if self._state is None:
            raise RuntimeError("call set_pointcloud() first")
        if prompt_points is None:
            if self._prompts is None:
                raise RuntimeError("no prompts: pass them or call set_prompts() first")
            prompt_points, prompt_labels, prompt_mask = self._prompts
        logits, scores = self.model.decode(self._state, prompt_points, prompt_labels, prompt_mask, multimask_output)
        self.model.check_coordinate_range()
        return logits, scores, logits
