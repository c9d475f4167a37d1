import ctypes, os, torch
L=ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)),"libmfma_peak.so"))
L.run.argtypes=[ctypes.c_void_p]+[ctypes.c_int]*4+[ctypes.c_void_p]
out=torch.empty(1<<22,device="cuda")
st=torch.cuda.current_stream().cuda_stream
for grid,block,nacc in [(256,256,4),(512,256,4),(768,256,2),(1024,256,1),(256,256,1),(256,512,2),(2048,256,4)]:
    iters=2000
    for rep in range(2):
        s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record(); L.run(out.data_ptr(),grid,block,iters,nacc,st); e.record(); torch.cuda.synchronize()
    ms=s.elapsed_time(e); n=grid*(block//64)*iters*8*nacc
    print(f"grid {grid} block {block} nacc {nacc}: {ms:.3f} ms {n*4096/ms/1e9:.1f} TF  ({n*64/ (256*4) / (ms*1e-3)/1e9:.2f} GHz-equivalent if pipe-bound)")
