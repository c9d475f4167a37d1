"""Seeded synthetic inputs of the benchmark workload: uniform clouds normalised into the unit ball like
evaluation/inference.py:58-59 (per cloud), rgb in [-1, 1] (the trained convention), one positive point prompt taken from the cloud."""
import torch


def synthetic_batch(B: int, N: int, seed: int = 42, num_prompts: int = 1):
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(B, N, 3, generator=g) * 2 - 1
    xyz = xyz - xyz.mean(dim=1, keepdim=True)
    xyz = xyz / xyz.norm(dim=2).max(dim=1).values.view(B, 1, 1)
    rgb = torch.rand(B, N, 3, generator=g) * 2 - 1
    pidx = torch.randint(0, N, (B, num_prompts), generator=g)
    prompt = torch.gather(xyz, 1, pidx.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    labels = torch.ones(B, num_prompts, dtype=torch.int64)
    return xyz.contiguous(), rgb.contiguous(), prompt, labels
