#!/bin/bash
cd $GRAFT_REPO_ROOT
python scripts/exp/r06_pipe_probe.py > gpurun_out/r06_pipe_probe.txt 2>&1
cat gpurun_out/r06_pipe_probe.txt
