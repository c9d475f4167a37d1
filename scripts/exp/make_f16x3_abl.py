"""Ablation builds of the pipelined f16x3 kernel (timing only; results are wrong by construction)."""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(here, "../../point-sam_amd/csrc/gemm_f16x3.hip")).read()
V = {
 0: [],
 1: [("        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AF[i][PA], WF[j][PW], acc[i][j], 0, 0, 0);\n#define HP_STEP", None)],  # placeholder (handled below)
}
def variant(n):
    s = src
    if n == 1:   # no MFMA in the pipelined kernel (keep one dependency so fragments are still read)
        s = s.replace("#define HP_TERM(AF, WF, PA, PW)                                                                                 \\\n    _Pragma(\"unroll\") for (int i = 0; i < TM; ++i) _Pragma(\"unroll\") for (int j = 0; j < TN; ++j)               \\\n        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AF[i][PA], WF[j][PW], acc[i][j], 0, 0, 0);",
                      "#define HP_TERM(AF, WF, PA, PW)                                                                                 \\\n    _Pragma(\"unroll\") for (int i = 0; i < TM; ++i) _Pragma(\"unroll\") for (int j = 0; j < TN; ++j)               \\\n        acc[i][j][0] += (float)AF[i][PA][0] + (float)WF[j][PW][0];")
        assert s != src
    if n == 2:   # cheap split
        s = s.replace("    const f32x2 xs = x * s;\n    const f16x2 h = __builtin_convertvector(xs, f16x2);\n    const f32x2 r = xs - __builtin_convertvector(h, f16x2 == f16x2 ? f32x2 : f32x2);", "")
        s = s.replace("    const f16x2 h = __builtin_convertvector(xs, f16x2);\n    const f32x2 r = xs - __builtin_convertvector(h, f32x2);\n    const f16x2 l = __builtin_convertvector(r, f16x2);\n    hi = __builtin_bit_cast(unsigned, h);\n    lo = __builtin_bit_cast(unsigned, l);",
                      "    hi = __builtin_bit_cast(unsigned, xs[0]); lo = __builtin_bit_cast(unsigned, xs[1]);")
        assert s != src
    if n == 3:   # no global loads in the loop
        s = s.replace("        load_slab((t + 3) * HG_BK, a_next, w_next);            // registers free again: slab t+3\n", "")
        assert s != src
    if n == 4:   # no LDS writes (split is dead code then)
        s = s.replace("        split_store(buf ^ 1, a_next, w_next);                  // slab t+1\n", "")
        assert s != src
    if n == 5:   # no barrier
        s = s.replace("        __builtin_amdgcn_sched_group_barrier(0x20, 8, 0);\n        __syncthreads();\n", "        __builtin_amdgcn_sched_group_barrier(0x20, 8, 0);\n")
        assert s != src
    if n == 6:   # MFMA only: no loads, no LDS writes, no barrier, no fragment re-reads
        s = s.replace("        load_slab((t + 3) * HG_BK, a_next, w_next);            // registers free again: slab t+3\n", "")
        s = s.replace("        split_store(buf ^ 1, a_next, w_next);                  // slab t+1\n", "")
        s = s.replace("        __builtin_amdgcn_sched_group_barrier(0x20, 8, 0);\n        __syncthreads();\n", "        __builtin_amdgcn_sched_group_barrier(0x20, 8, 0);\n")
    if n in (7, 8):   # loads only: no MFMA, no split, no LDS writes; loads kept alive through a dummy accumulation
        s = s.replace("#define HP_TERM(AF, WF, PA, PW)                                                                                 \\\n    _Pragma(\"unroll\") for (int i = 0; i < TM; ++i) _Pragma(\"unroll\") for (int j = 0; j < TN; ++j)               \\\n        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AF[i][PA], WF[j][PW], acc[i][j], 0, 0, 0);",
                      "#define HP_TERM(AF, WF, PA, PW)")
        s = s.replace("        split_store(buf ^ 1, a_next, w_next);                  // slab t+1\n", "        _Pragma(\"unroll\") for (int i = 0; i < NF4; ++i) acc[0][0][i] += a_next[i][0] + w_next[i][0];\n")
        s = s.replace("        load_frags(buf, 1, af1, wf1);\n", "").replace("        load_frags(buf ^ 1, 0, af0, wf0);                      // (after the last slab: the all-zero slab, unused)\n", "")
        assert "HP_TERM(AF, WF, PA, PW)\n" in s
    if n == 8:        # ... and no barrier
        s = s.replace("        __builtin_amdgcn_sched_group_barrier(0x20, 8, 0);\n        __syncthreads();\n", "        __builtin_amdgcn_sched_group_barrier(0x20, 8, 0);\n")
    s += "\nvoid psam_set_error(const char*) {}\n"
    return s
for n in range(9):
    f = os.path.join(here, f"gemm_f16x3_abl{n}.hip")
    open(f, "w").write(variant(n))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(here, "../../point-sam_amd/csrc"),
                           "-I" + os.path.join(here, "../../include"), f, "-o", os.path.join(here, f"libgemm_f16x3_abl{n}.so")], stderr=subprocess.DEVNULL)
    os.remove(f)
print("ok")
