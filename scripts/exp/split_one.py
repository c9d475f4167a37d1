import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from point_sam_amd import ops
M,N,K = 32768,4096,1024
x=torch.randn(M,K,device="cuda"); W=torch.randn(N,K,device="cuda"); b=torch.randn(N,device="cuda"); y=torch.empty(M,N,device="cuda")
with ops.gemm_mode("bf16x6"):
    for _ in range(3): ops.linear(x,W,b,out=y)
torch.cuda.synchronize()
