#!/bin/bash
# Builds the experiments variant of the library (PSAM_BUILD_EXPERIMENTS: the measured-and-rejected kernels too) NEXT TO the production one, without touching
# the in-tree objects: scripts/exp/libpointsam_experiments.so, loaded with PSAM_LIB_PATH.  Run here (no GPU needed); ~10 minutes.
set -e
cd "$(dirname "$0")/../.."
C=point_sam_amd/csrc; O=/tmp/psam_exp_objs; mkdir -p $O
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -DPSAM_BUILD_EXPERIMENTS"
build() { src=$1; shift; b=$(basename ${src%.*}); /opt/rocm/bin/hipcc $F -I$C "$@" -c $C/$src -o $O/$b.o; }
build tokenizer.hip -ffp-contract=off & build gemm.hip & build gemm_split.hip & build rowops.hip & wait
build gemm_f16x3p.hip & build gemm_f16x3pp.hip & build attention.hip & build blocks.hip & wait
build experiments/gemm_f16x3q.hip & build experiments/gemm_f16x3s.hip & build experiments/gemm_f16x3c.hip & build experiments/twoway.hip & wait
/opt/rocm/bin/hipcc $F -x hip -c $C/error.cpp -o $O/error.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/exp/libpointsam_experiments.so $O/*.o
python -c "
from point_sam_amd import isa_lint
bad = isa_lint.lint('scripts/exp/libpointsam_experiments.so'); print('isa lint:', len(bad), 'hazardous instructions'); assert not bad"
ls -la scripts/exp/libpointsam_experiments.so
