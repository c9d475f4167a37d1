#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu 2>&1 | tail -25 > gpurun_out/r06_gpu_tests_v2.txt
cat gpurun_out/r06_gpu_tests_v2.txt
