#!/bin/bash
# Builds an ablation variant of the library (gemm_f16x3p.hip with -DPSAM_GEMM_ABLATE) next to the production one.  Run here (no GPU needed).
set -e
cd "$(dirname "$0")/../.."
C=point_sam_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DPSAM_GEMM_ABLATE -I$C -Iinclude -c $C/gemm_f16x3p.hip -o scripts/exp/gemm_f16x3p_abl.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DPSAM_GEMM_ABLATE -I$C -Iinclude -c $C/gemm_f16x3pp.hip -o scripts/exp/gemm_f16x3pp_abl.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DPSAM_GEMM_ABLATE -I$C -Iinclude -c $C/experiments/gemm_f16x3c.hip -o scripts/exp/gemm_f16x3c_abl.o
OBJS=$(ls $C/*.o | grep -v "gemm_f16x3p.o\|gemm_f16x3pp.o\|gemm_f16x3c.o\|gemm_f16x3s.o\|gemm_f16x3q.o\|twoway.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/exp/libpointsam_abl.so $OBJS scripts/exp/gemm_f16x3p_abl.o scripts/exp/gemm_f16x3pp_abl.o scripts/exp/gemm_f16x3c_abl.o
ls -la scripts/exp/libpointsam_abl.so
