import ctypes, os, sys, torch, collections
import numpy as np
here=os.path.dirname(os.path.abspath(__file__))
L=ctypes.CDLL(os.path.join(here,"libgemm_f16x3_dbg.so"))
L.psam_gemm_f16x3.argtypes=[ctypes.c_void_p,ctypes.c_int64,ctypes.c_void_p,ctypes.c_void_p,ctypes.c_int64,ctypes.c_void_p,ctypes.c_void_p,ctypes.c_int64,ctypes.c_void_p,ctypes.c_void_p,ctypes.c_int64,ctypes.c_void_p,ctypes.c_int64]+[ctypes.c_int32]*4+[ctypes.c_float,ctypes.c_int32,ctypes.c_void_p]
L.psam_row_scale_f16.argtypes=[ctypes.c_void_p,ctypes.c_int64,ctypes.c_int32,ctypes.c_int32,ctypes.c_void_p,ctypes.c_void_p]
L.dbg_set_buffer.argtypes=[ctypes.c_void_p]; L.psam_gemm_f16x3_force_config.argtypes=[ctypes.c_int32]
st=torch.cuda.current_stream().cuda_stream
for (M,N,K) in [(4096,3072,1024),(4096,5504,1024),(4096,1024,2752)]:
  for cfg in (0,1,3):
    x=torch.randn(M,K,device="cuda"); W=torch.randn(N,K,device="cuda"); y=torch.empty(M,N,device="cuda")
    sa=torch.empty(M,device="cuda"); sw=torch.empty(N,device="cuda")
    L.psam_row_scale_f16(x.data_ptr(),K,M,K,sa.data_ptr(),st); L.psam_row_scale_f16(W.data_ptr(),K,N,K,sw.data_ptr(),st)
    bn=128 if cfg!=1 else 64
    nt=((M+127)//128)*((N+bn-1)//bn)
    dbg=torch.zeros(nt,8,dtype=torch.int64,device="cuda"); L.dbg_set_buffer(dbg.data_ptr()); L.psam_gemm_f16x3_force_config(cfg)
    for _ in range(3): L.psam_gemm_f16x3(x.data_ptr(),K,sa.data_ptr(),W.data_ptr(),K,sw.data_ptr(),y.data_ptr(),N,0,0,0,0,0,0,M,N,K,1.0,0,st)
    torch.cuda.synchronize()
    d=dbg.cpu().numpy()
    t0=d[:,0].min(); start=(d[:,0]-t0)/100.0; end=(d[:,1]-t0)/100.0   # us (100 MHz)
    hw=d[:,5]; xcc=d[:,6]&0xf; cu=(hw>>8)&0xf; se=(hw>>13)&0x7; sh=(hw>>12)&1
    cuid=xcc*1000+se*100+sh*20+cu
    per=collections.Counter(cuid.tolist())
    print(f"== {M}x{N}x{K} cfg{cfg}: tiles {nt}  span {end.max():.1f} us  distinct CUs {len(per)}  tiles/CU min {min(per.values())} max {max(per.values())}")
    print(f"   cycles/WG: prologue {d[:,2].mean():.0f}  loop {d[:,3].mean():.0f} ({d[:,3].mean()/((K+31)//32):.0f}/slab)  epilogue {d[:,4].mean():.0f};  WG lifetime mean {np.mean(end-start):.1f} us (min {np.min(end-start):.1f} max {np.max(end-start):.1f});  clock {np.mean((d[:,2]+d[:,3]+d[:,4])/((d[:,1]-d[:,0])*10.0)):.2f} GHz")
    qs=[0,5,20,40,60,80,100,150,200]
    print("   starts by time bucket(us):", {f"<{q}":int((start<q).sum()) for q in qs[1:]})
    ts=np.linspace(0,end.max(),41)
    conc=[((start<=t)&(end>t)).sum()/len(per) for t in ts]
    print("   resident WGs/CU at 40 time points:", " ".join(f"{c:.1f}" for c in conc))
    first=start<5
    print(f"   first-round WGs: n={first.sum()} lifetime {np.mean((end-start)[first]):.1f} us; later WGs: n={(~first).sum()} lifetime {np.mean((end-start)[~first]) if (~first).any() else 0:.1f} us, loop cyc first {d[first,3].mean():.0f} later {d[~first,3].mean() if (~first).any() else 0:.0f}")
