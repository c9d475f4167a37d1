import ctypes, os, sys, torch, collections
here=os.path.dirname(os.path.abspath(__file__))
L=ctypes.CDLL(os.path.join(here,"libgemm_dbg.so"))
L.gemm_dbg.argtypes=[ctypes.c_void_p]*4+[ctypes.c_int]*4+[ctypes.c_void_p]*2
for (M,N,K,cfg) in [(4096,3072,1024,0),(4096,3072,1024,1),(4096,5504,1024,0),(4096,1024,1024,1),(4096,5504,1024,1)]:
    x=torch.randn(M,K,device="cuda"); W=torch.randn(N,K,device="cuda"); b=torch.randn(N,device="cuda"); y=torch.empty(M,N,device="cuda")
    bm=64 if cfg==2 else 128; bn=128 if cfg==0 else 64
    nt=((M+bm-1)//bm)*((N+bn-1)//bn)
    dbg=torch.zeros(nt,4,dtype=torch.int64,device="cuda")
    st=torch.cuda.current_stream().cuda_stream
    for _ in range(3): L.gemm_dbg(x.data_ptr(),W.data_ptr(),y.data_ptr(),b.data_ptr(),M,N,K,cfg,dbg.data_ptr(),st)
    torch.cuda.synchronize()
    d=dbg.cpu()
    t0=d[:,0].min().item()
    start=(d[:,0]-t0).double()/100.0  # us (100 MHz)
    end=(d[:,1]-t0).double()/100.0
    hw=d[:,2]; xcc=d[:,3]&0xf
    cu=(hw>>8)&0xf; sh=(hw>>12)&1; se=(hw>>13)&0x7
    cuid=xcc*128+se*32+sh*16+cu
    dur=end-start
    print(f"== {M}x{N}x{K} cfg{cfg}: tiles {nt} total {end.max():.1f} us; WG dur mean {dur.mean():.1f} min {dur.min():.1f} max {dur.max():.1f}")
    per=collections.Counter(cuid.tolist())
    print("   distinct CUs", len(per), "WGs/CU histogram", sorted(collections.Counter(per.values()).items()))
    # how many WGs start late (after 10us) and their CU co-location
    late=start>10
    print(f"   late-start WGs: {int(late.sum())}; start times quantiles {[round(float(start[late].quantile(q)),1) for q in (0,.25,.5,.75,1)] if late.any() else []}")
    if late.any():
        latecu=collections.Counter(cuid[late].tolist())
        print("   late WGs per CU histogram", sorted(collections.Counter(latecu.values()).items()), " dur of late WGs mean %.1f"%dur[late].mean(), " dur of early mean %.1f"%dur[~late].mean())
    # concurrency per CU over time: for each WG, count of overlapping WGs on same CU at its midpoint
    import itertools
    byc=collections.defaultdict(list)
    for i,c in enumerate(cuid.tolist()): byc[c].append((start[i].item(),end[i].item()))
    busy=[]; 
    for c,l in byc.items():
        l.sort(); busy.append(max(e for s,e in l))
    print("   per-CU finish time: min %.1f mean %.1f max %.1f"%(min(busy),sum(busy)/len(busy),max(busy)))
