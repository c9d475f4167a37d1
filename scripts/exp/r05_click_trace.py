"""One additional click of a cfg5 session (ViT-giant, N=32768, 512x64, encoder cached), eager, under `rocprofv3 --kernel-trace`: the ordered kernel
list of one decode with its durations.  A torch.cumsum launch separates the decodes in the trace.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/click -o t -- python scripts/exp/r05_click_trace.py run
    python scripts/exp/r05_click_trace.py report gpurun_out/click/**/t_kernel_trace.csv > profiles/r05/r05_click_kernels.txt"""
import sys, os, csv, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

if sys.argv[1] == "run":
    import torch
    from point_sam_amd.config import get_config
    from point_sam_amd.model import PointCloudSAM
    from point_sam_amd.weights import random_state_dict
    from point_sam_amd.synthetic import synthetic_batch
    cfg = get_config("giant", 512, 64)
    model = PointCloudSAM(cfg, random_state_dict(cfg, 42), "cuda")
    xyz, rgb, prompt, labels = (t.cuda() for t in synthetic_batch(1, 32768, seed=42))
    st = model.encode(xyz, rgb, model.tokenize(xyz))
    masks, iou = model.decode(st, prompt, labels, None, True)
    pc = torch.cat([prompt, xyz[:, :2]], 1); pl = torch.ones(1, 3, dtype=labels.dtype, device="cuda")
    best = masks[:, 0].contiguous()
    mark = torch.ones(64, device="cuda")
    for _ in range(12):
        torch.cumsum(mark, 0)
        model.decode(st, pc, pl, best, False)
    torch.cumsum(mark, 0)
    torch.cuda.synchronize()
else:
    files = [f for a in sys.argv[2:] for f in glob.glob(a, recursive=True)]
    rows = sorted(csv.DictReader(open(files[0])), key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "cumsum" in r["Kernel_Name"].lower() or "scan" in r["Kernel_Name"].lower()]
    a, b = marks[-2], marks[-1]
    seg = rows[a + 1:b]
    t0 = int(seg[0]["Start_Timestamp"])
    tot = 0
    print(f"# one additional click (3 point prompts + mask prompt), eager: {len(seg)} launches")
    print(f"# {'start_us':>9s} {'dur_us':>7s}  kernel")
    for r in seg:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot += d
        print(f"  {(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {d:7.1f}  {r['Kernel_Name'][:110]}")
    print(f"# sum of kernel durations {tot:.1f} us; first start -> last end {(int(seg[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
