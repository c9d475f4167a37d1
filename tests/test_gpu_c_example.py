"""A caller that is not Python: examples/eva_block_from_c.c (plain C99 + the HIP runtime) drives one transformer block through the C ABI --
prepare (the library packs the weights), run twice, bitwise repeatable.  Built with gcc against include/pointsam_hip.h and the in-tree library."""
import os
import shutil
import subprocess

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_caller_runs_a_transformer_block(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("needs gcc and the ROCm headers")
    from point_sam_amd import _lib
    _lib.load()      # fails loudly if the library is not built
    libdir = os.path.join(ROOT, "point_sam_amd", "csrc")
    exe = str(tmp_path / "eva_block_from_c")
    cmd = ["gcc", "-std=c99", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "eva_block_from_c.c"), "-L" + libdir, "-lpointsam_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print("\n" + r.stdout + r.stderr)
    assert r.returncode == 0 and "ok" in r.stdout and "non-finite 0" in r.stdout


def test_c_caller_runs_the_whole_path_against_the_oracle(tmp_path):
    """examples/predict_masks_from_c.c: the whole path -- tokenizer, patch embedding, transformer blocks, prompt encodings, two-way decoder,
    hyper-networks, upscaling -- from plain C through the C ABI on a flat-binary dump (examples/make_c_demo_blob.py: seeded weights, one cloud,
    and what the CPU oracle computes for them).  The C program itself asserts FPS indices identical and logits / IoU within 1e-3."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("needs gcc and the ROCm headers")
    import sys
    from point_sam_amd import _lib
    _lib.load()
    blob = str(tmp_path / "c_demo.blob")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "make_c_demo_blob.py"), blob], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HIP_VISIBLE_DEVICES=""))      # the dump (oracle) runs on the CPU
    assert r.returncode == 0, r.stderr[-2000:]
    libdir = os.path.join(ROOT, "point_sam_amd", "csrc")
    exe = str(tmp_path / "predict_masks_from_c")
    cmd = ["gcc", "-std=c99", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "predict_masks_from_c.c"), "-L" + libdir, "-lpointsam_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    r = subprocess.run([exe, blob], capture_output=True, text=True, timeout=300)
    print("\n" + r.stdout + r.stderr)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok") and "indices differing from the oracle: 0 of" in r.stdout
