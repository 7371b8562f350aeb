#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 300 python -m pytest tests/test_training.py -m gpu -q -s -k "multi_tile and False" 2>&1 | grep -E "multi-tile|failed" | cut -c1-330; done
echo HEAD_OFF; NERFDS_WGRAD_HEAD_OFF=1 timeout 300 python -m pytest tests/test_training.py -m gpu -q -s -k "multi_tile and False" 2>&1 | grep -E "multi-tile|failed" | cut -c1-330
echo TR_OFF+HEAD_OFF; NERFDS_WGRAD_TR_OFF=1 NERFDS_WGRAD_HEAD_OFF=1 timeout 300 python -m pytest tests/test_training.py -m gpu -q -s -k "multi_tile and False" 2>&1 | grep -E "multi-tile|failed" | cut -c1-330
echo UNMERGED; NERFDS_TRAIN_MERGED=0 timeout 300 python -m pytest tests/test_training.py -m gpu -q -s -k "multi_tile and False" 2>&1 | grep -E "multi-tile|failed" | cut -c1-330
echo G16_OFF; NERFDS_TRAIN_G16=0 timeout 300 python -m pytest tests/test_training.py -m gpu -q -s -k "multi_tile and False" 2>&1 | grep -E "multi-tile|failed" | cut -c1-330
