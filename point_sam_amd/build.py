"""Builds csrc/libpointsam_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m point_sam_amd.build [--force]

The library is built IN-TREE so that it travels with the repository snapshot to the GPU box.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libpointsam_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
# (source, extra flags).  The tokenizer must not contract a*b+c into fma: FPS / kNN indices are bit-exact
# against the oracle's fp32 arithmetic (oracle/tokenizer_oracle.c is built with -ffp-contract=off too).
# PSAM_BUILD_EXPERIMENTS=1: also build the measured-and-rejected paths under csrc/experiments/ (the unit-ring GEMM gemm_f16x3q.hip, the persistent GEMMs gemm_f16x3s.hip (stream-K) and gemm_f16x3c.hip (continuous), the one-launch token side twoway.hip, the
# forked two-way decoder, the 128x512 row-LayerNorm GEMM tile) -- off by default: they are not on the product path, and their objects, exports, ISA
# lint and tests cost every build and every test run (tests that need them skip unless the library was built with them).
EXPERIMENTS = os.environ.get("PSAM_BUILD_EXPERIMENTS", "0") == "1"
SOURCES = [
    ("tokenizer.hip", ["-ffp-contract=off"]),
    ("gemm.hip", []),
    ("gemm_split.hip", []),
    ("gemm_f16x3p.hip", []),
    ("gemm_f16x3pp.hip", []),
    ("attention.hip", []),
    ("rowops.hip", []),
    ("blocks.hip", []),
    ("error.cpp", ["-x", "hip"]),
] + ([("experiments/gemm_f16x3q.hip", ["-I" + CSRC]), ("experiments/gemm_f16x3s.hip", ["-I" + CSRC]), ("experiments/gemm_f16x3c.hip", ["-I" + CSRC]),
        ("experiments/twoway.hip", ["-I" + CSRC])] if EXPERIMENTS else [])
if EXPERIMENTS:
    COMMON = COMMON + ["-DPSAM_BUILD_EXPERIMENTS"]
FLAGS_STAMP = os.path.join(CSRC, ".build_flags")
# measurement aid of bench.py (matrix-pipe ceiling probe): its own small library, not part of the product ABI
PROBE_SRC = os.path.join(CSRC, "probe", "mfma_probe.hip")
PROBE_LIB = os.path.join(CSRC, "probe", "libpsam_probe.so")


def _stale(out, deps):
    return not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    hdrs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]   # every header: an edit to any rebuilds all objects
    hdrs.append(os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "pointsam_hip.h"))   # the public ABI header is part of every translation unit (common.h)
    flags = " ".join(COMMON)
    # another flag set than the objects were built with (experiments on / off): every object is stale.  No stamp = a tree built before the stamp
    # existed, or a snapshot that dropped it: the default flags are assumed (the GPU box must not spend minutes rebuilding a library that travelled).
    if (open(FLAGS_STAMP).read() != flags) if os.path.exists(FLAGS_STAMP) else EXPERIMENTS:
        force = True
    jobs = []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        if force or _stale(o, [s, os.path.abspath(__file__)] + hdrs):
            jobs.append([HIPCC] + COMMON + extra + ["-c", s, "-o", o])
    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stderr)
        return r.stderr
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(CSRC, os.path.splitext(s)[0] + ".o") for s, _ in SOURCES]
    if force or jobs or _stale(LIB, objs):
        # Link under a temporary name and move it into place only once the ISA lint has passed: a library that could not be linted (llvm-objdump
        # missing, lint crashed) or that failed it never sits at LIB, where the next build_library() would find it "fresh" and return it unchecked.
        tmp = LIB + ".unlinted"
        run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", tmp] + objs)
        # no packed-FP32 instruction with a non-uniform op_sel may ship: wrong lanes 48-63 beside another stream's GEMM workgroups (isa_lint.py)
        try:
            from . import isa_lint
        except ImportError:      # run as a script
            sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
            import isa_lint
        bad = isa_lint.lint(tmp)
        if bad:
            os.replace(tmp, LIB + ".rejected")
            raise RuntimeError("libpointsam_hip.so contains packed-FP32 instructions with a non-uniform op_sel (see point_sam_amd/isa_lint.py):\n" +
                               "\n".join(f"  {k}: {i}" for k, i in bad))
        os.replace(tmp, LIB)
        with open(FLAGS_STAMP, "w") as f:
            f.write(flags)
    if force or _stale(PROBE_LIB, [PROBE_SRC]):
        run([HIPCC, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared", "-o", PROBE_LIB, PROBE_SRC])
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
