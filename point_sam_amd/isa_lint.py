"""ISA lint of the built library: no packed-FP32 arithmetic with a non-uniform op_sel.

On gfx950 `v_pk_mul_f32` / `v_pk_add_f32` / `v_pk_fma_f32` whose op_sel bits differ between the sources (the LOW result takes the HIGH register of one
source pair and the low register of another, e.g. `v_pk_mul_f32 v[52:53], v[74:75], v[42:43] op_sel:[0,1]`) return a result computed from a wrong (zero)
operand in lanes 48-63 while workgroups with a large LDS / register allocation (the LDS-DMA GEMM kernels of another stream) are being launched on the
same CU: 1e7 wrong results of 1.6e10 beside the f16x3 GEMM, 0 alone, 0 for the uniform forms, for `v_pk_mov_b32` and for the scalar `v_mul_f32`
(scripts/exp/r04_pk_opsel.hip, profiles/r04/r04_hazard.txt).  That was round 3's "cross-lane hazard" (1e-2 wrong fc2 outputs in multi-stream runs).  hipcc
selects these forms by itself when a packed operation broadcasts element 1 of a 64-bit value, so the library is checked after every build:

    python -m point_sam_amd.isa_lint [path/to/libpointsam_hip.so]

prints the offending instructions (kernel, instruction) and exits non-zero if there are any.  The cure at the source is to detach the scalar from its
pair (`asm volatile("" : "+v"(w))`), which makes the compiler broadcast the LOW register (`op_sel_hi`, a safe form).
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = os.environ.get("PSAM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
_PK = re.compile(r"^\s*(v_pk_(?:mul|add|fma)_f32)\s+(.*?)(?:\s*//.*)?$")
_SEL = re.compile(r"\bop_sel:\[([01](?:,[01])+)\]")


def hazardous(instruction: str) -> bool:
    """True for a packed-FP32 arithmetic instruction whose op_sel (LOW-result selectors) is not the same bit for every source."""
    m = _PK.match(instruction)
    if not m:
        return False
    s = _SEL.search(m.group(2))
    return bool(s) and len(set(s.group(1).split(","))) > 1


def device_disassembly(lib: str):
    """Yields (kernel, instruction) for every instruction of every gfx950 code object bundled in `lib`."""
    objdump = os.path.join(LLVM, "llvm-objdump")
    with tempfile.TemporaryDirectory() as tmp:
        copy = os.path.join(tmp, os.path.basename(lib))
        shutil.copyfile(lib, copy)
        subprocess.run([objdump, "--offloading", copy], check=True, capture_output=True)      # writes <copy>.N.<triple> next to the input
        objs = sorted(f for f in os.listdir(tmp) if "amdgcn" in f)
        if not objs:
            raise RuntimeError(f"{lib}: no device code objects found")
        for f in objs:
            r = subprocess.run([objdump, "-d", os.path.join(tmp, f)], check=True, capture_output=True, text=True)
            kernel = "?"
            for line in r.stdout.splitlines():
                k = re.match(r"^[0-9a-f]+ <(.*)>:", line)
                if k:
                    kernel = k.group(1)
                elif line.startswith(("\t", " ")):
                    yield kernel, line.strip()


def lint(lib: str):
    return [(k, i.split("//")[0].strip()) for k, i in device_disassembly(lib) if hazardous(i)]


def main(argv):
    lib = argv[1] if len(argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libpointsam_hip.so")
    bad = lint(lib)
    for k, i in bad:
        print(f"{k}: {i}")
    print(f"{lib}: {len(bad)} packed-FP32 instruction(s) with a non-uniform op_sel")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
