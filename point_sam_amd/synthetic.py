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


def ply_batch(B: int, N: int, seed: int = 42, num_prompts: int = 1, path: str = None):
    """The second input distribution of SURVEY.md 8(d): the reference's six demo point clouds (demo/static/models/*.ply, as its own loader
    normalises them: tests/golden/ref_demo_ply.npz holds those arrays) tiled and jittered to N points -- surface samples with clusters,
    empty space and near-duplicates instead of a uniform ball.  Cloud b is demo cloud b mod (number of clouds): its points repeated to cover N
    (a random permutation per copy), every copy after the first displaced by a Gaussian jitter of 2e-3, re-normalised into the unit ball."""
    import os
    import numpy as np
    if path is None:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_demo_ply.npz")
    z = np.load(path)
    names = sorted(k[:-5] for k in z.files if k.endswith("__xyz"))
    g = torch.Generator().manual_seed(seed)
    xyzs, rgbs = [], []
    for b in range(B):
        name = names[b % len(names)]
        x = torch.from_numpy(z[name + "__xyz"]).float()
        c = torch.from_numpy(z[name + "__rgb_u8"]).float() / 255.0 * 2 - 1
        n0, parts_x, parts_c, have = x.shape[0], [], [], 0
        while have < N:
            perm = torch.randperm(n0, generator=g)[: min(n0, N - have)]
            jit = torch.randn(perm.numel(), 3, generator=g) * 2e-3 if have else torch.zeros(perm.numel(), 3)
            parts_x.append(x[perm] + jit); parts_c.append(c[perm]); have += perm.numel()
        x, c = torch.cat(parts_x), torch.cat(parts_c)
        x = x - x.mean(0, keepdim=True)
        x = x / x.norm(dim=1).max()
        xyzs.append(x); rgbs.append(c)
    xyz, rgb = torch.stack(xyzs).contiguous(), torch.stack(rgbs).contiguous()
    pidx = torch.randint(0, N, (B, num_prompts), generator=g)
    prompt = torch.gather(xyz, 1, pidx.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    return xyz, rgb, prompt, torch.ones(B, num_prompts, dtype=torch.int64)
