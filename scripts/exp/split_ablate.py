import ctypes, os, sys, torch
here=os.path.dirname(os.path.abspath(__file__))
L=ctypes.CDLL(os.path.join(here,"libgemm_split_ablate.so"))
L.gemm_split_ablate.argtypes=[ctypes.c_void_p]*4+[ctypes.c_int]*4+[ctypes.c_void_p]
names={0:"full",1:"cheap-split(hi only)",2:"hh term only (1/6 MFMA)",3:"no gload in loop",4:"no ds_write/barrier",5:"no gload/ds_write/barrier",6:"ds_read+mfma only",7:"mfma only"}
for (M,N,K) in [(4096,3072,1024),(4096,5504,1024),(32768,4096,1024)]:
    x=torch.randn(M,K,device="cuda"); W=torch.randn(N,K,device="cuda"); b=torch.randn(N,device="cuda"); y=torch.empty(M,N,device="cuda")
    line=f"{M}x{N}x{K}: "
    for abl in range(8):
        st=torch.cuda.current_stream().cuda_stream
        for _ in range(2): L.gemm_split_ablate(x.data_ptr(),W.data_ptr(),y.data_ptr(),b.data_ptr(),M,N,K,abl,st)
        s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for _ in range(10): L.gemm_split_ablate(x.data_ptr(),W.data_ptr(),y.data_ptr(),b.data_ptr(),M,N,K,abl,st)
        e.record(); torch.cuda.synchronize(); ms=s.elapsed_time(e)/10
        line+=f"{names[abl]}: {ms*1e3:.0f}us {2*M*N*K/ms/1e9:.0f}TF | "
    print(line, flush=True)
