"""Writes the flat-binary input of examples/predict_masks_from_c.c: the seeded state dict of a small model on which every coarse C-ABI entry
applies (ViT of 2 blocks x 256 x 4 heads of 64, SwiGLU hidden 682; 256 groups of 32; the reference's decoder geometry), one synthetic cloud of 4096
points with one positive click, and what the CPU ORACLE (oracle/pointsam_oracle.py, test infrastructure) computes for it: FPS indices, mask logits,
IoU predictions.  Format: "PSAMBLOB", u32 count, then per tensor u32 name length, name, u32 dtype (0 = f32, 1 = i64), u32 ndim, u32 dims, raw data.

    python examples/make_c_demo_blob.py out.blob
"""
import os
import struct
import sys
from dataclasses import replace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pointsam_oracle as O  # noqa: E402
from point_sam_amd.config import ViTConfig, get_config  # noqa: E402
from point_sam_amd.weights import random_state_dict  # noqa: E402


def demo_config():
    return replace(get_config("base", 256, 32), vit=ViTConfig("c_demo_eva02", 256, 2, 4, 682, True))


def main(path):
    cfg = demo_config()
    sd = random_state_dict(cfg, seed=21)
    xyz, rgb, prompt, labels = O.synthetic_batch(1, 4096, seed=22)
    masks, iou, mid = O.predict_masks(sd, cfg, xyz, rgb, prompt, labels, None, True, mode="exact", return_intermediates=True)
    tensors = {k: v for k, v in sd.items()}
    tensors.update({"in.xyz": xyz, "in.rgb": rgb, "in.prompt": prompt, "in.labels": labels.to(torch.int64),
                    "want.fps_idx": mid["patches"]["fps_idx"].to(torch.int64), "want.masks": masks, "want.iou": iou})
    with open(path, "wb") as f:
        f.write(b"PSAMBLOB" + struct.pack("<I", len(tensors)))
        for name, t in tensors.items():
            t = t.detach().cpu().contiguous()
            dt = 1 if t.dtype == torch.int64 else 0
            if dt == 0:
                t = t.to(torch.float32)
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)) + nb + struct.pack("<II", dt, t.dim()) + struct.pack("<%dI" % t.dim(), *t.shape))
            f.write(t.numpy().tobytes())
    print(f"{path}: {len(tensors)} tensors, logits in [{masks.min():.3f}, {masks.max():.3f}], iou {iou.flatten().tolist()}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "c_demo.blob")
