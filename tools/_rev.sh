for f in 1 0; do
  echo REVERSE_SIGMA=$f
  NERFDS_TRAIN_REVERSE_SIGMA=$f python -m pytest tests/test_training.py -m gpu -q -s -k "norm or multi_tile or sigma_gradient or elastic" 2>&1 | grep -E "worst|passed|failed|Error|assert|target_norm" | cut -c1-250
  NERFDS_TRAIN_REVERSE_SIGMA=$f python tools/objective_time.py 2>&1 | tail -6
done
