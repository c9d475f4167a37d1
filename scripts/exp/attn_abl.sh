#!/bin/bash
# Builds an ablation variant of the library (attention.hip with -DPSAM_ATTN_ABLATE) next to the production one.  Run here (no GPU needed).
set -e
cd "$(dirname "$0")/../.."
C=point_sam_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DPSAM_ATTN_ABLATE -Iinclude -I$C -c $C/attention.hip -o scripts/exp/attention_abl.o
OBJS=$(ls $C/*.o | grep -v attention.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/exp/libpointsam_attn_abl.so $OBJS scripts/exp/attention_abl.o
ls -la scripts/exp/libpointsam_attn_abl.so
