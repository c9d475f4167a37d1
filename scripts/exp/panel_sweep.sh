#!/bin/bash
# Tile-order sweep of the f16x3p GEMM (column-panel width, PSAM_GEMM_PANEL) on the encoder shapes.  GPU box.
cd "${GRAFT_REPO_ROOT:-.}"
for P in 0 auto 1 2 3 6; do
  if [ "$P" = auto ]; then unset PSAM_GEMM_PANEL; else export PSAM_GEMM_PANEL=$P; fi
  echo "== PSAM_GEMM_PANEL=$P"
  timeout 200 python scripts/gemm_p_bench.py 21,14 2>&1 | grep -v "^check\|amdgpu.ids" | head -12
done
