import ctypes, os, torch, collections
here=os.path.dirname(os.path.abspath(__file__))
L=ctypes.CDLL(os.path.join(here,"libgemm_split_dbg.so"))
L.gemm_split_dbg.argtypes=[ctypes.c_void_p]*4+[ctypes.c_int]*3+[ctypes.c_void_p]*2
M,N,K=8192,4096,1024
x=torch.randn(M,K,device="cuda"); W=torch.randn(N,K,device="cuda"); b=torch.randn(N,device="cuda"); y=torch.empty(M,N,device="cuda")
nt=(M//128)*(N//128)
dbg=torch.zeros(nt,4,8,dtype=torch.int64,device="cuda")
st=torch.cuda.current_stream().cuda_stream
for _ in range(2): L.gemm_split_dbg(x.data_ptr(),W.data_ptr(),y.data_ptr(),b.data_ptr(),M,N,K,dbg.data_ptr(),st)
torch.cuda.synchronize()
d=dbg.cpu()
w0=d[:,0,:]
barA=(w0[:,1]-w0[:,0]).double(); stage=(w0[:,2]-w0[:,1]).double(); slab=(w0[:,3]-w0[:,0]).double(); s16=((w0[:,4]-w0[:,3]).double())/15
print("tiles",nt,"per-slab cycles (wave 0): wait@barrierA mean %.0f  store+barrierB mean %.0f  whole slab (t=8) mean %.0f  avg slab over t=9..24: %.0f (min %.0f max %.0f)"%(barA.mean(),stage.mean(),slab.mean(),s16.mean(),s16.min(),s16.max()))
print("cycles: prologue mean %.0f  loop mean %.0f  epilogue mean %.0f (max %.0f)"%(w0[:,5].double().mean(), w0[:,6].double().mean(), w0[:,7].double().mean(), w0[:,7].double().max()))
