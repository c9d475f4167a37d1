"""Cooperative FPS (N = 131072, G = 2048, B = 1: cfg #3): time per iteration for the points-per-thread choices and placements."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1:
    import torch
    from point_sam_amd import ops
    mode = int(sys.argv[1])
    ops._lib.load().psam_fps_set_cooperative(mode)
    shapes = ((8, 32768, 512), (1, 32768, 512)) if os.environ.get("FPS_SWEEP") == "cfg2" else ((1, 131072, 2048), (2, 131072, 2048), (4, 65536, 1024))
    for B, N, G in shapes:
        xyz = torch.rand(B, N, 3, device="cuda")
        for _ in range(2): ops.fps(xyz, G)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops.fps(xyz, G)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"  B={B} N={N} G={G}: {ms:.3f} ms, {ms * 1000 / G:.3f} us/iter", flush=True)
else:
    sweep = [(4, 1, 8), (4, 1, 9), (1, 1, 8), (2, 1, 9)] if os.environ.get("FPS_SWEEP") == "cfg2" else [(p, m, 9) for p in (4, 2, 1) for m in (1, 2)]
    for ppt4, mode, ming in sweep:
        print(f"PSAM_FPS_COOP_PPT4={ppt4} PSAM_FPS_COOP_MIN_GROUPS={ming} placement={'one XCD per cloud' if mode == 1 else 'spread'}", flush=True)
        subprocess.run([sys.executable, __file__, str(mode)], env=dict(os.environ, PSAM_FPS_COOP_PPT4=str(ppt4), PSAM_FPS_COOP_MIN_GROUPS=str(ming)))
