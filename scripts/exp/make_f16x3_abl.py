"""Ablation builds of the pipelined f16x3 kernel (timing only; results are wrong by construction)."""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(here, "../../point_sam_amd/csrc/gemm_f16x3.hip")).read()
MFMA = "        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][PA[term]], wf[j][PW[term]], acc[i][j], 0, 0, 0);\n    };"
RELOAD_W = "            if (isw) w[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, kok ? offW[i] : OOB, knext * 4, 0));\n"
RELOAD_A = "            else a[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, kok ? offA[i] : OOB, knext * 4, 0));\n"
STORES = "            *reinterpret_cast<u32x2*>(st) = u32x2{ph0, h1};\n            *reinterpret_cast<u32x2*>(st + PLANE) = u32x2{pl0, l1};\n"
BAR = "        __syncthreads();\n        // ---- region B"
for x in (MFMA, RELOAD_W, RELOAD_A, STORES, BAR): assert x in src, x
def variant(n):
    s = src
    if n in (1, 7, 8):   # no MFMA
        s = s.replace(MFMA, "        acc[i][j][0] += (float)af[i][PA[term]][0] + (float)wf[j][PW[term]][0];\n    };")
    if n == 2:           # cheap split
        s = s.replace("    const f16x2 h = __builtin_convertvector(xs, f16x2);\n    const f32x2 r = xs - __builtin_convertvector(h, f32x2);\n    const f16x2 l = __builtin_convertvector(r, f16x2);\n    hi = __builtin_bit_cast(unsigned, h);\n    lo = __builtin_bit_cast(unsigned, l);",
                      "    hi = __builtin_bit_cast(unsigned, xs[0]); lo = __builtin_bit_cast(unsigned, xs[1]);")
        assert s != src
    if n in (3, 6):      # no global loads in the loop
        s = s.replace(RELOAD_W, "").replace(RELOAD_A, "")
    if n in (4, 6, 7, 8):   # no LDS writes
        s = s.replace(STORES, "            acc[0][0][1] += __builtin_bit_cast(float, ph0 ^ h1 ^ pl0 ^ l1);\n" if n in (7, 8) else "")
    if n in (5, 6, 8):   # no barrier
        s = s.replace(BAR, "        // ---- region B")
    s += "\nvoid psam_set_error(const char*) {}\n"
    return s
for n in range(9):
    f = os.path.join(here, f"gemm_f16x3_abl{n}.hip")
    open(f, "w").write(variant(n))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(here, "../../point_sam_amd/csrc"),
                           "-I" + os.path.join(here, "../../include"), f, "-o", os.path.join(here, f"libgemm_f16x3_abl{n}.so")], stderr=subprocess.DEVNULL)
    os.remove(f)
print("ok")
