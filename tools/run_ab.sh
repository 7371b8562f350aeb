#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{ timeout 2400 python -m pytest tests/test_gpu_parity.py -q 2>&1 | tail -12; } > gpurun_out/gputests.log 2>&1
cat gpurun_out/gputests.log
