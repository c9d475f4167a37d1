import ctypes, os, sys, collections
import torch
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, "libcumask.so"))
def mk(mask_bits):
    words = (ctypes.c_uint32 * 8)(*[sum(((mask_bits >> (32 * w + b)) & 1) << b for b in range(32)) for w in range(8)])
    s = ctypes.c_void_p()
    rc = L.cumask_stream_create(ctypes.byref(s), words, 8)
    assert rc == 0, rc
    return s
def probe(s, blocks=4096):
    out = torch.zeros(2 * blocks, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    rc = L.cumask_probe(s, ctypes.c_void_p(out.data_ptr()), blocks, 20000)
    assert rc == 0
    torch.cuda.synchronize()
    o = out.cpu().view(-1, 2).tolist()
    cnt = collections.Counter()
    for hw, xcc in o:
        hw &= 0xffffffff
        cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 0x7
        cnt[(xcc & 0xf, se, sh, cu)] += 1
    return cnt
pats = {"all": (1 << 256) - 1, "low128": (1 << 128) - 1, "even": sum(1 << i for i in range(0, 256, 2)), "bit0-7": 0xff, "bits 0,8,16..": sum(1 << i for i in range(0, 256, 8)),
        "low32": (1 << 32) - 1, "bits32-63": ((1 << 32) - 1) << 32}
for name, m in pats.items():
    c = probe(mk(m))
    xccs = collections.Counter(k[0] for k in c)
    print(f"{name:14s}: {len(c)} distinct (xcc,se,sh,cu); per XCC: {dict(sorted(xccs.items()))}")
    if len(c) <= 40:
        print("     ", sorted(c.keys()))
