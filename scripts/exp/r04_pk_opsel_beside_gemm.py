"""Runs every form of scripts/exp/r04_pk_opsel.hip alone and while the library's f16x3 GEMM kernel (LDS-DMA ring + MFMA) runs on another stream.  GPU box."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from point_sam_amd import ops
P = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libr04_pk_opsel.so"))
P.pk_probe_name.restype = ctypes.c_char_p
torch.manual_seed(0)
M, D = 4096, 1024
x = torch.randn(M, D, device="cuda"); wq = ops.F16Weight(torch.randn(3 * D, D, device="cuda") / 32); bq = torch.zeros(3 * D, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
ITERS, BLOCKS, REPS = 20000, 1024, 6
with ops.gemm_mode("f16x3"):
    xp, sx = ops.scale_pack_rows_g8(x)
    out = torch.empty(M, 3 * D, device="cuda")
    torch.cuda.synchronize()
    form = 0
    while P.pk_probe_name(form) and not os.environ.get("SKIP_FORMS"):
        for gemm in (0, 1):
            tot = [0] * 8
            for rep in range(REPS):
                if gemm:
                    with torch.cuda.stream(s2):
                        for _ in range(400): ops.linear(xp, wq, bq, x_scale=sx, x_packed=True, out=out)
                buf = (ctypes.c_uint * 8)()
                assert P.pk_probe_run(form, ITERS, BLOCKS, ctypes.c_void_p(s1.cuda_stream), buf) == 0
                tot = [a + b for a, b in zip(tot, buf)]
                torch.cuda.synchronize()
            n = ITERS * BLOCKS * 8 * 16 * REPS
            print(f"{P.pk_probe_name(form).decode().splitlines()[0]:72s} {'beside GEMM' if gemm else 'alone':12s} wrong LOW per lane quarter {tot[0::2]}  wrong HIGH {tot[1::2]}  (of {n:.1e} each)", flush=True)
        form += 1
    # which property of the neighbour matters?  form 0 beside synthetic MFMA aggressors
    sink = torch.zeros(64, dtype=torch.int32, device="cuda")
    for kind, what, n_it, n_blk in ((0, "short-lived waves (2e6 workgroups x 20 MFMAs), small footprint", 10, 2000000), (3, "short-lived waves (2e6 workgroups x 20 MFMAs), 64 KiB LDS", 10, 2000000),
                                     (2, "short-lived waves (2e6 workgroups x 20 MFMAs), 296 registers", 10, 2000000)):
        tot = [0] * 8
        for rep in range(REPS):
            assert P.pk_probe_aggressor(kind, n_it, n_blk, ctypes.c_void_p(s2.cuda_stream), ctypes.c_void_p(sink.data_ptr())) == 0
            buf = (ctypes.c_uint * 8)()
            assert P.pk_probe_run(0, ITERS, BLOCKS, ctypes.c_void_p(s1.cuda_stream), buf) == 0
            tot = [a + b for a, b in zip(tot, buf)]
            torch.cuda.synchronize()
        print(f"op_sel:[0,1] beside {what:66s} wrong LOW per lane quarter {tot[0::2]}  wrong HIGH {tot[1::2]}", flush=True)
    for kind, what in ((0, "MFMA loop, small register footprint"), (1, "MFMA loop, 251 VGPRs, no AGPRs"), (2, "MFMA loop, 231 VGPRs + 64 AGPRs (> 256 registers)"), (3, "MFMA loop, small footprint, 64 KiB LDS")):
        tot = [0] * 8
        for rep in range(REPS):
            assert P.pk_probe_aggressor(kind, 400000, 2048, ctypes.c_void_p(s2.cuda_stream), ctypes.c_void_p(sink.data_ptr())) == 0
            buf = (ctypes.c_uint * 8)()
            assert P.pk_probe_run(0, ITERS, BLOCKS, ctypes.c_void_p(s1.cuda_stream), buf) == 0
            tot = [a + b for a, b in zip(tot, buf)]
            torch.cuda.synchronize()
        print(f"op_sel:[0,1] beside {what:55s} wrong LOW per lane quarter {tot[0::2]}  wrong HIGH {tot[1::2]}", flush=True)
