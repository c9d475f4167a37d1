"""Assembly-level variants of the PRE-FIX kernel (see r04_hazard_variants.py): the compiler's own code for variant 0, with ONLY the instructions
named below inserted after its counted waits -- source-level probes (variants 6-9) made the compiler re-place its waits (it emitted lgkmcnt(0)).
  20  after every `s_waitcnt lgkmcnt(k>0)` that follows a ds_write_b128 within 3 instructions (the pass loops): s_nop 15 x 2 (32 wait states)
  21  same sites: s_nop 3 (4 wait states)
  22  same sites: the wait itself becomes lgkmcnt(0) (control = the cure, expressed at ISA level)
  23  same sites: s_nop 0 (1 wait state)
  24  same sites: s_nop 15 x 8 (128 wait states)
Run HERE after r04_hazard_variants.py; needs /tmp scratch."""
import os, re, shlex, subprocess, sys, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "scripts", "exp", "hazard_tree", "point_sam_amd", "csrc")
W = "/tmp/hazasm"
os.makedirs(W, exist_ok=True)
DEV_S = "gemm_f16x3p-hip-amdgcn-amd-amdhsa-gfx950.s"
if not os.path.exists(os.path.join(W, "cmds.txt")) or not os.path.exists(os.path.join(W, DEV_S + ".orig")):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-unused-function", "-DPSAM_HAZ=0",
           "-save-temps", "-v", "-c", os.path.join(CSRC, "gemm_f16x3p.hip"), "-o", "gemm_f16x3p_asm.o"]
    r = subprocess.run(cmd, cwd=W, capture_output=True, text=True); assert r.returncode == 0, r.stderr[-3000:]
    open(os.path.join(W, "cmds.txt"), "w").write(r.stderr)
    shutil.copyfile(os.path.join(W, DEV_S), os.path.join(W, DEV_S + ".orig"))
cmds = [l.strip() for l in open(os.path.join(W, "cmds.txt")) if l.startswith(' "')]
dev_as = next(c for c in cmds if "-cc1as" in c and "amdgcn" in c)
after = cmds[cmds.index(dev_as):]
# stages to re-run: device assembler, lld, bundler, then the host stages that embed the bundle (emit-llvm-bc, -S, cc1as)
rerun = [c for c in after if ("-cc1as" in c or "lld" in c.split()[0] or "clang-offload-bundler" in c.split()[0] or "-emit-llvm-bc" in c or " -S " in c)]
orig = open(os.path.join(W, DEV_S + ".orig")).read().split("\n")
def patch(kind):
    out, sites = [], 0
    if 30 <= kind < 40:
        # 30: every counted lgkm wait -> lgkmcnt(0); 31: every counted vm wait -> vmcnt(0); 32: both (every s_waitcnt drains both counters);
        # 33: as 32, restricted to the kernel instance that fails (gemm_f16x3p_kernel<2,2,2,2,2,0,0,2>)
        name, inside = "_Z18gemm_f16x3p_kernelILi2ELi2ELi2ELi2ELi2ELi0ELi0ELi2EEv8F16PArgs:", False
        for l in orig:
            if re.match(r"^_Z\w+:", l): inside = l.startswith(name)
            m = re.match(r"\s*s_waitcnt (.*)$", l)
            if m and (kind != 33 or inside):
                a = m.group(1)
                lg = re.search(r"lgkmcnt\((\d+)\)", a); vm = re.search(r"vmcnt\((\d+)\)", a)
                parts = []
                if vm: parts.append("vmcnt(0)" if kind in (31, 32, 33) else vm.group(0))
                if lg: parts.append("lgkmcnt(0)" if kind in (30, 32, 33) else lg.group(0))
                if kind in (32, 33): parts = ["vmcnt(0)", "lgkmcnt(0)"]
                new = "\ts_waitcnt " + " ".join(parts)
                if new.split() != l.split(): sites += 1
                out.append(new); continue
            out.append(l)
        return "\n".join(out), sites
    if kind >= 40:
        # ISA-level probes inside the failing kernel instance only (gemm_f16x3p_kernel<2,2,2,2,2,0,0,2> = cfg 21):
        # 40: s_nop 4 after every v_pk_*      41: s_nop 4 before every v_pk_*      42: s_nop 4 after every v_mov_b64
        # 43: every v_mov_b64 -> two v_mov_b32  44: s_nop 4 after every v_mfma       45: s_nop 15 x 4 + full drain at the epilogue entry (after the last s_barrier of the K loop)
        # 46: s_nop 4 after every VALU instruction (v_*) that is not an MFMA
        name, inside = "_Z18gemm_f16x3p_kernelILi2ELi2ELi2ELi2ELi2ELi0ELi0ELi2EEv8F16PArgs:", False
        nbar = 0
        for l in orig:
            if re.match(r"^_Z\w+:", l): inside = l.startswith(name); nbar = 0
            t = l.strip()
            if inside:
                if kind == 41 and t.startswith("v_pk_"): out.append("\ts_nop 4"); sites += 1
                if kind == 43 and t.startswith("v_mov_b64"):
                    m = re.match(r"v_mov_b64(?:_e32)? v\[(\d+):(\d+)\], v\[(\d+):(\d+)\]$", t)
                    if m:
                        a, b, c, d = map(int, m.groups())
                        # the compiler's own order (low half first) is only safe when the pairs do not overlap the wrong way round
                        assert not (a == d), t
                        out.append(f"\tv_mov_b32_e32 v{a}, v{c}"); out.append(f"\tv_mov_b32_e32 v{b}, v{d}"); sites += 1
                        continue
                if kind in (60, 61, 62) and re.match(r"v_pk_mul_f32 .* op_sel:\[0,1\]$", t):
                    # 60: every DS operation drained before the instruction   61: 64 wait states before it   62: drained before AND 8 wait states after
                    out += {60: ["\ts_waitcnt lgkmcnt(0)"], 61: ["\ts_nop 15"] * 4, 62: ["\ts_waitcnt lgkmcnt(0)"]}[kind]; sites += 1
                    out.append(l)
                    if kind == 62: out.append("\ts_nop 7")
                    continue
                if kind in (50, 51, 52):
                    # 50: every v_pk_mul_f32 whose LOW result reads a HIGH source register (op_sel:[0,1] / op_sel:[1,0]) -> two v_mul_f32 with the same operands
                    # 51: only those with op_sel:[0,1]   52: only those with op_sel:[1,0]
                    m = re.match(r"v_pk_mul_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] op_sel:\[(\d),(\d)\]$", t)
                    if m and ((kind == 50) or (kind == 51 and m.group(7) == "0") or (kind == 52 and m.group(7) == "1")):
                        d0, d1, a0, a1, b0, b1 = map(int, m.groups()[:6]); sa, sb = int(m.group(7)), int(m.group(8))
                        alo, blo = (a1 if sa else a0), (b1 if sb else b0)          # low result; the high result takes the high registers (op_sel_hi default)
                        assert d0 not in (a1, b1), t                               # low result written first: it must not clobber a source of the high result
                        out.append(f"\tv_mul_f32_e32 v{d0}, v{alo}, v{blo}"); out.append(f"\tv_mul_f32_e32 v{d1}, v{a1}, v{b1}"); sites += 1
                        continue
                out.append(l)
                if kind == 40 and t.startswith("v_pk_"): out.append("\ts_nop 4"); sites += 1
                if kind == 42 and t.startswith("v_mov_b64"): out.append("\ts_nop 4"); sites += 1
                if kind == 44 and t.startswith("v_mfma"): out.append("\ts_nop 4"); sites += 1
                if kind == 46 and t.startswith("v_") and not t.startswith("v_mfma"): out.append("\ts_nop 4"); sites += 1
                if kind == 45 and t == "s_barrier":
                    nbar += 1
                    if nbar == 7: out += ["\ts_waitcnt vmcnt(0) lgkmcnt(0)"] + ["\ts_nop 15"] * 4; sites += 1
                continue
            out.append(l)
        return "\n".join(out), sites
    for i, l in enumerate(orig):
        m = re.match(r"\s*s_waitcnt lgkmcnt\((\d+)\)\s*$", l)
        recent = [x for x in orig[max(0, i - 8):i] if x.strip() and not x.strip().startswith((";", "."))][-3:]
        if m and int(m.group(1)) > 0 and any("ds_write_b128" in x for x in recent):
            sites += 1
            if kind == 22: out.append("\ts_waitcnt lgkmcnt(0)"); continue
            out.append(l)
            out += {20: ["\ts_nop 15"] * 2, 21: ["\ts_nop 3"], 23: ["\ts_nop 0"], 24: ["\ts_nop 15"] * 8}[kind]
            continue
        out.append(l)
    return "\n".join(out), sites
base = ["tokenizer", "gemm", "gemm_split", "attention", "rowops", "error"]
for kind in [int(a) for a in sys.argv[1:]] or [20, 21, 22, 23, 24, 30, 31, 32, 33]:
    s, sites = patch(kind)
    open(os.path.join(W, DEV_S), "w").write(s)
    for c in rerun:
        r = subprocess.run(shlex.split(c), cwd=W, capture_output=True, text=True)
        assert r.returncode == 0, c[:200] + "\n" + r.stderr[-2000:]
    o = os.path.join(CSRC, f"gemm_f16x3p_haz{kind}.o")
    shutil.copyfile(os.path.join(W, "gemm_f16x3p_asm.o"), o)
    objs = [os.path.join(CSRC, b + ".o") for b in base] + [o, os.path.join(CSRC, "gemm_f16x3pp_haz0.o")]
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(CSRC, f"libpointsam_hip_haz{kind}.so")] + objs, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    print(f"asm variant {kind}: {sites} sites patched", flush=True)
