"""Round 6: layernorm_v4_kernel<4> at the ViT-L block shape (4096 x 1024 fp32 in, g8-packed out + row scale + row bound) against plain device copies of the same
bytes: how far is the 9.5 us launch from what the memory system gives a one-round kernel of this size?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from point_sam_amd import ops


def time_us(fn, n=50):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(n):
            fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


L = ops._lib.load()
for M, D in ((4096, 1024), (2048, 1024), (512, 1408), (262144, 512)):
    x = torch.randn(M, D, device="cuda"); w = torch.randn(D, device="cuda"); b = torch.randn(D, device="cuda")
    Dp = (D + 31) // 32 * 32
    y = torch.empty(M, Dp, device="cuda"); rs = torch.empty(M, device="cuda"); rb = torch.empty(M, device="cuda")
    st = lambda: torch.cuda.current_stream().cuda_stream
    def ln():
        ops.check(L.psam_layernorm_ex2(x.data_ptr(), D, None, 0, w.data_ptr(), b.data_ptr(), y.data_ptr(), Dp, M, D, 1e-6, 0, rs.data_ptr(), 1, rb.data_ptr(), 0.0, 1.0, 0.0, st()), "ln")
    z = torch.empty_like(x)
    t_ln, t_cp = time_us(ln), time_us(lambda: z.copy_(x))
    nb = M * D * 4 + M * Dp * 4
    print(f"{M} x {D}: layernorm (packed out) {t_ln:6.1f} us = {nb / t_ln / 1e6:5.2f} TB/s; torch copy of the same bytes {t_cp:6.1f} us = {2 * M * D * 4 / t_cp / 1e6:5.2f} TB/s", flush=True)
