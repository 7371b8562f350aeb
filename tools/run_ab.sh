#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
 timeout 2400 python -m pytest tests/test_render_image_gpu.py "tests/test_gpu_parity.py::test_nerf_ds_graph_tiny" tests/test_training.py -k "top_level or tiny or multi_tile" -q -m gpu --tb=short 2>&1 | grep -v "^  " | tail -80
} > gpurun_out/gputests.log 2>&1
tail -90 gpurun_out/gputests.log
