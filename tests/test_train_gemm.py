"""Kernel-level parity of the trainer's MFMA layers (csrc/train_gemm.hip): the forward / data-gradient layer in every
variant (segments, ReLU, mask, accumulate, two- and three-way split) and the weight gradient - fp32 operands, the f16 X / f16 dY operands
of the fused training step with the bias gradient as its by-product, and the heads (f16 X, narrow fp32 dY) -, against fp64 sums, at sample
counts the trainer tests do not reach - several tiles per persistent workgroup and a partial last tile.  The checker is the
small C++ program tools/bench_dense.hip (built next to the library); it is also the micro-benchmark quoted in DESIGN 8.1."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CHECK = os.path.join(HERE, '..', 'nerf-ds_amd', 'nerfds_amd', '_lib', 'train_gemm_check')


@pytest.mark.gpu
@pytest.mark.parametrize('rows', [40, 16416, 70001])
def test_mfma_layers_against_fp64(rows):
  assert os.path.exists(CHECK), 'train_gemm_check is not built (make -C nerf-ds_amd/csrc)'
  out = subprocess.run([CHECK, str(rows)], capture_output=True, text=True, timeout=600)
  print(out.stdout)
  assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
  assert out.stdout.count('max|err|') == 35 and 'FAIL' not in out.stdout and '!!' not in out.stdout
