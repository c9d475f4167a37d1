import ctypes, os, torch
here=os.path.dirname(os.path.abspath(__file__))
names={0:"full",1:"no MFMA",2:"cheap split",3:"no gload in loop",4:"no ds_write",5:"no barrier",6:"mfma+dsread+split only",7:"loads+split+barrier (no mfma, no dswrite)",8:"loads+split only"}
st=torch.cuda.current_stream().cuda_stream
libs={}
for n in names:
    L=ctypes.CDLL(os.path.join(here,f"libgemm_f16x3_abl{n}.so"))
    L.psam_gemm_f16x3.argtypes=[ctypes.c_void_p,ctypes.c_int64,ctypes.c_void_p,ctypes.c_void_p,ctypes.c_int64,ctypes.c_void_p,ctypes.c_void_p,ctypes.c_int64,ctypes.c_void_p,ctypes.c_void_p,ctypes.c_int64,ctypes.c_void_p,ctypes.c_int64]+[ctypes.c_int32]*4+[ctypes.c_float,ctypes.c_int32,ctypes.c_void_p]
    L.psam_gemm_f16x3_force_config.argtypes=[ctypes.c_int32]; L.psam_gemm_f16x3_force_config(3); libs[n]=L
for (M,N,K) in [(4096,1024,2752),(4096,3072,1024),(4096,5504,1024)]:
    x=torch.randn(M,K,device="cuda"); W=torch.randn(N,K,device="cuda"); y=torch.empty(M,N,device="cuda")
    sa=torch.ones(M,device="cuda")*1024; sw=torch.ones(N,device="cuda")*1024
    line=f"{M}x{N}x{K}: "
    for n,L in libs.items():
        f=lambda: L.psam_gemm_f16x3(x.data_ptr(),K,sa.data_ptr(),W.data_ptr(),K,sw.data_ptr(),y.data_ptr(),N,0,0,0,0,0,0,M,N,K,1.0,0,st)
        for _ in range(3): f()
        s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for _ in range(10): f()
        e.record(); torch.cuda.synchronize(); us=s.elapsed_time(e)*100
        line+=f"{names[n]}: {us:.0f}us | "
    print(line,flush=True)
    line="   ld=0 (all operand loads hit L1/L2): "
    for n in (0,6,7,8):
        L=libs[n]
        f=lambda: L.psam_gemm_f16x3(x.data_ptr(),0,sa.data_ptr(),W.data_ptr(),0,sw.data_ptr(),y.data_ptr(),N,0,0,0,0,0,0,M,N,K,1.0,0,st)
        for _ in range(3): f()
        s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for _ in range(10): f()
        e.record(); torch.cuda.synchronize(); us=s.elapsed_time(e)*100
        line+=f"{names[n]}: {us:.0f}us | "
    print(line,flush=True)
