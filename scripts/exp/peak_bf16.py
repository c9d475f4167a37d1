import ctypes, os, torch
L=ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)),"libmfma_peak_bf16.so"))
L.run.argtypes=[ctypes.c_void_p, ctypes.c_void_p]+[ctypes.c_int]*3+[ctypes.c_void_p]
out=torch.empty(1<<22,device="cuda")
st=torch.cuda.current_stream().cuda_stream
for fill in ("zeros","randn"):
    inp=(torch.zeros(512,8) if fill=="zeros" else torch.randn(512,8)).bfloat16().cuda()
    for grid,nacc,iters in [(256,4,2000),(512,4,2000),(512,4,40000),(768,2,40000)]:
        for rep in range(2):
            s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); s.record(); L.run(out.data_ptr(),inp.data_ptr(),grid,iters,nacc,st); e.record(); torch.cuda.synchronize()
        ms=s.elapsed_time(e); n=grid*4*iters*8*nacc
        print(f"{fill}: grid {grid} nacc {nacc} iters {iters}: {ms:.3f} ms {n*32768/ms/1e9:.0f} TF")
