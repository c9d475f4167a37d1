"""Compiles one csrc/*.hip for gfx950 with -Rpass-analysis=kernel-resource-usage and prints a table per kernel: registers, spills, scratch, occupancy.
    python scripts/kernel_resources.py gemm_f16x3pp.hip [name-filter] [-D...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = [a for a in sys.argv[2:] if not a.startswith("-")]
extra = [a for a in sys.argv[2:] if a.startswith("-")]
csrc = os.path.join(ROOT, "point_sam_amd", "csrc")
with tempfile.TemporaryDirectory() as d:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-I" + csrc, "-I" + os.path.join(ROOT, "include"),
                        "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(csrc, src), "-o", os.path.join(d, "o.o")] + extra, capture_output=True, text=True)
if r.returncode:
    print(r.stderr[-4000:]); sys.exit(1)
for b in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
    name = subprocess.run(["c++filt", b.split()[0]], capture_output=True, text=True).stdout.strip()
    if flt and not any(f in name for f in flt):
        continue
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    scr, occ = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")
    print(f"{name[:100]:100s} vgpr {g('VGPRs'):>3} agpr {g('AGPRs'):>3} sgpr {g('TotalSGPRs'):>3} scratch {scr:>3} vspill {g('VGPRs Spill'):>3} sspill {g('SGPRs Spill'):>3} occ {occ}")
