"""Is the GEMM bound by L2-miss operand traffic?  Time each kernel normally and with lda = ldw = 0 (every tile re-reads ONE row:
all operand loads hit L1/L2; results meaningless, instruction stream identical)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from point_sam_amd import ops
L = ops._lib.load()
st = lambda: torch.cuda.current_stream().cuda_stream
def timeit(fn):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / 10 * 1e3
for name, M, N, K in [("qkv", 4096, 3072, 1024), ("fc1", 4096, 5504, 1024), ("fc2", 4096, 1024, 2752), ("big", 32768, 4096, 1024)]:
    x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5; y = torch.empty(M, N, device="cuda")
    sa, sw = ops.row_scale_f16(x), ops.row_scale_f16(W)
    line = f"{name:4s} {M}x{N}x{K} |"
    for cfg in (0, 1):
        L.psam_gemm_f16x3_force_config(cfg); L.psam_gemm_bf16x6_force_config(cfg)
        for ld in (K, 0):
            t = timeit(lambda: L.psam_gemm_f16x3(x.data_ptr(), ld, sa.data_ptr(), W.data_ptr(), ld, sw.data_ptr(), y.data_ptr(), N, 0, 0, 0, 0, 0, 0, M, N, K, 1.0, 0, st()))
            line += f" f16x3:{cfg} ld={ld}: {t:6.1f}us {2*M*N*K/t/1e6:5.0f}TF |"
        for ld in (K, 0):
            t = timeit(lambda: L.psam_gemm_bf16x6(x.data_ptr(), ld, 0, 0, W.data_ptr(), ld, 0, 0, y.data_ptr(), N, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, M, N, K, 1, 1, 1.0, 0, st()))
            line += f" bf16x6:{cfg} ld={ld}: {t:6.1f}us {2*M*N*K/t/1e6:5.0f}TF |"
    print(line, flush=True)
