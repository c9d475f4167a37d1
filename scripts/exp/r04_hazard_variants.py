"""Variants of the PRE-FIX GEMM epilogue (commit 1179a69, the pass loop that produced 1e-2 wrong fc2 outputs beside another stream) for the
hazard hunt of VERDICT r03 item 2.  Run HERE (needs .git): extracts the old tree into scripts/exp/hazard_tree/ (git-ignored, travels with
gpurun), patches the fetch of the pass loop behind -DPSAM_HAZ=n and builds one library per variant:

  0  as committed                                   (control: must fail)
  1  32 wait states between the VALU that computes the bpermute address and the ds_bpermute (VALU -> DS hazard on the aliased register?)
  2  ds_bpermute through inline asm with early-clobber destinations (destination != address register), same lgkmcnt(1) before the use
  3  s_waitcnt lgkmcnt(0) before the prefetched operands are taken (the known cure; control: must pass)
  4  as 0, plus the whole LDS allocation filled with NaN at kernel entry (stale-LDS read?)
  5  as 0, ds_read issued BEFORE the three ds_bpermute (does the stale value follow the last-issued bpermute?)

To re-run on the GPU box take the hazard_tree lines out of .gpurunignore first (the tree + its variant libraries are ~110 MB per push).
scripts/exp/r04_hazard_run.sh runs the reproducer (graph pipeline at the bench configuration vs eager) on each of them on the GPU box.
"""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
TREE = os.path.join(ROOT, "scripts", "exp", "hazard_tree")
CSRC = os.path.join(TREE, "point_sam_amd", "csrc")
if not os.path.exists(os.path.join(TREE, "bench.py")):
    os.makedirs(TREE, exist_ok=True)
    subprocess.run(f"git -C {ROOT} archive 1179a69 | tar -x -C {TREE}", shell=True, check=True)

ep = os.path.join(CSRC, "gemm_epilogue.h")
s = open(ep).read()
if "PSAM_HAZ" not in s:
    old_fetch = """                    o.rsq = __shfl(rows.rs[i >> 1], srcl, 64);
                    o.lmean = 0.f; o.lrstd = 1.f;
                    if (o_lnc) { o.lmean = __shfl(rows.mean[i >> 1], srcl, 64); o.lrstd = __shfl(rows.rstd[i >> 1], srcl, 64); }
                    o.v = ep_f32x4{0.f, 0.f, 0.f, 0.f}; o.x = o.v;
                    if (ALL_ON || lane_on) {
                        o.v = ep_load4(lw + rl * LD + scol);
                        if (o_swiglu) o.x = ep_load4(lw + rl * LD + scol + 32);
                    }
                    return o;"""
    new_fetch = """                    auto hz_shfl = [&](float val, int sl) -> float {
#if PSAM_HAZ == 1
                        int a = sl << 2;
                        asm volatile("s_nop 15\\n\\ts_nop 15" : "+v"(a));
                        return __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(val)));
#elif PSAM_HAZ == 2
                        int a = sl << 2, out;
                        asm volatile("ds_bpermute_b32 %0, %1, %2" : "=&v"(out) : "v"(a), "v"(__float_as_int(val)) : "memory");
                        return __int_as_float(out);
#else
                        return __shfl(val, sl, 64);
#endif
                    };
                    o.v = ep_f32x4{0.f, 0.f, 0.f, 0.f}; o.x = o.v;
#if PSAM_HAZ == 5
                    if (ALL_ON || lane_on) {
                        o.v = ep_load4(lw + rl * LD + scol);
                        if (o_swiglu) o.x = ep_load4(lw + rl * LD + scol + 32);
                    }
                    asm volatile("" ::: "memory");
#endif
                    o.rsq = hz_shfl(rows.rs[i >> 1], srcl);
                    o.lmean = 0.f; o.lrstd = 1.f;
                    if (o_lnc) { o.lmean = hz_shfl(rows.mean[i >> 1], srcl); o.lrstd = hz_shfl(rows.rstd[i >> 1], srcl); }
#if PSAM_HAZ != 5
                    if (ALL_ON || lane_on) {
                        o.v = ep_load4(lw + rl * LD + scol);
                        if (o_swiglu) o.x = ep_load4(lw + rl * LD + scol + 32);
                    }
#endif
                    return o;"""
    assert old_fetch in s
    s = s.replace(old_fetch, new_fetch)
    old_top = """                    const PassIn cur = nxt;
                    if (q + 1 < np) nxt = fetch(q + 1);"""
    new_top = """#if PSAM_HAZ == 2
                    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(nxt.rsq), "+v"(nxt.lmean), "+v"(nxt.lrstd) :: "memory");    // the compiler does not count the asm bpermutes
#elif PSAM_HAZ == 3
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nxt.rsq), "+v"(nxt.lmean), "+v"(nxt.lrstd) :: "memory");
#endif
                    const PassIn cur = nxt;
                    if (q + 1 < np) nxt = fetch(q + 1);"""
    assert old_top in s
    s = s.replace(old_top, new_top)
    s = s.replace("#pragma once", "#pragma once\n#ifndef PSAM_HAZ\n#define PSAM_HAZ 0\n#endif", 1)
    assert "define PSAM_HAZ" in s
    open(ep, "w").write(s)
kp = os.path.join(CSRC, "gemm_f16x3p.hip")
k = open(kp).read()
if "PSAM_HAZ" not in k:
    old = "    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];\n"
    new = old + """#if PSAM_HAZ == 4
    for (int w_ = threadIdx.x; w_ < S * STAGE / 4; w_ += 64 * NW) reinterpret_cast<unsigned*>(smem)[w_] = 0x7fc00000u;
    __syncthreads();
#endif
"""
    assert k.count(old) == 1
    open(kp, "w").write(k.replace(old, new))

HIPCC = "/opt/rocm/bin/hipcc"
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-unused-function"]
base = [("tokenizer.hip", ["-ffp-contract=off"]), ("gemm.hip", []), ("gemm_split.hip", []), ("attention.hip", []), ("rowops.hip", []), ("error.cpp", ["-x", "hip"])]
def run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode: raise RuntimeError(" ".join(cmd) + "\n" + r.stderr[-4000:])
jobs = []
for src, extra in base:
    o = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
    if not os.path.exists(o): jobs.append([HIPCC] + COMMON + extra + ["-c", os.path.join(CSRC, src), "-o", o])
variants = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 4, 5]
for n in variants:
    for src in ("gemm_f16x3p.hip", "gemm_f16x3pp.hip"):
        o = os.path.join(CSRC, f"{os.path.splitext(src)[0]}_haz{n}.o")
        jobs.append([HIPCC] + COMMON + [f"-DPSAM_HAZ={n}", "-c", os.path.join(CSRC, src), "-o", o])
with ThreadPoolExecutor(max_workers=6) as ex: list(ex.map(run, jobs))
for n in variants:
    objs = [os.path.join(CSRC, os.path.splitext(s_)[0] + ".o") for s_, _ in base] + [os.path.join(CSRC, f"gemm_f16x3p_haz{n}.o"), os.path.join(CSRC, f"gemm_f16x3pp_haz{n}.o")]
    run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(CSRC, f"libpointsam_hip_haz{n}.so")] + objs)
    print("built variant", n, flush=True)
