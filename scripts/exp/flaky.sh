#!/bin/bash
# Failure rate of test_graph_pipeline_matches_eager after the BatchPipeline tests (debugging aid, GPU box): 10 fresh processes.
cd "${GRAFT_REPO_ROOT:-.}"
fails=0
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x --tb=line -p no:cacheprovider -k "batch_pipeline or graph_pipeline" > /tmp/f.log 2>&1 || fails=$((fails+1))
done
echo "$fails / 10 failed"
