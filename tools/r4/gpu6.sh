#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_6; mkdir -p $O
timeout 1500 python tools/precision_study_gpu.py --rays 480000 --chunk 8192 --modes bf16x3,f16x3,f16mx8,bf16mx8,i8x2,i8x2u,f16w2,f16 --out $O/precision_modes_480000.json > $O/precision_modes.log 2>&1
timeout 1200 python tools/precision_study_gpu.py --rays 65536 --chunk 8192 --modes bf16x3 --layers --out $O/precision_layers_65536.json > $O/precision_layers.log 2>&1
tail -12 $O/precision_modes.log; tail -20 $O/precision_layers.log | cut -c1-260
