"""Numpy emulation of how render_kernel.hip consumes the packed MFMA weight stream.

It restates, independently of the C++ packer, the conventions of the device code:
  * fragment = [32 rows m] x [16 k-slots (h, i)], lane = (h << 5) | m, fp32 stream: part 0 = slots i<4, part 1 = i>=4;
  * an output tile's accumulator register r of lane half h is row (r & 3) + 8 (r >> 2) + 4 h, and registers
    8c .. 8c+7 become k-slots (h, 0..7) of activation chunk 2*tile + c of the next layer;
  * linear input chunks: slot (c, h, i) <-> feature 16 c + 8 h + i;
  * heads: logical output j is accumulator register j (rows duplicated over both halves).
Used by the CPU tests to check the packer + stream order against the oracle's MLPs without a GPU.
"""
import numpy as np


class Stream:
  def __init__(self, wbytes: np.ndarray, bias: np.ndarray, prec: str):
    self.prec = prec
    self.fb = 1024 if prec == 'bf16' else 2048
    self.w = wbytes
    self.bias = bias
    self.fi = 0
    self.bt = 0

  def next_frag(self) -> np.ndarray:
    """Returns A[32 rows][2 halves][8 slots] as float64."""
    raw = self.w[self.fi * self.fb:(self.fi + 1) * self.fb]
    self.fi += 1
    A = np.zeros((32, 2, 8))
    if self.prec == 'f32':
      p0 = raw[:1024].view(np.float32).reshape(64, 4)
      p1 = raw[1024:].view(np.float32).reshape(64, 4)
      v = np.concatenate([p0, p1], axis=1)
    else:
      def bf(x):
        return (x.view(np.uint16).astype(np.uint32) << 16).view(np.float32).reshape(64, 8)
      v = bf(raw[:1024])
      if self.prec == 'bf16x3':
        v = v.astype(np.float64) + bf(raw[1024:])
    for lane in range(64):
      A[lane & 31, lane >> 5] = v[lane]
    return A

  def next_bias(self) -> np.ndarray:
    b = self.bias[self.bt * 32:(self.bt + 1) * 32]
    self.bt += 1
    return b.astype(np.float64)


def linear_chunks(feats: np.ndarray, n_chunks: int) -> np.ndarray:
  """feats [N, F] -> chunks [n_chunks, 2, 8, N]: slot (c, h, i) <-> feature 16c + 8h + i, zero padded."""
  n, f = feats.shape
  pad = np.zeros((n, 16 * n_chunks))
  pad[:, :f] = feats
  return pad.T.reshape(n_chunks, 2, 8, n)


def tile_to_chunks(acc: np.ndarray) -> np.ndarray:
  """acc [32 rows, N] -> 2 chunks [2, 2, 8, N] (register r of half h = row (r&3) + 8(r>>2) + 4h)."""
  out = np.zeros((2, 2, 8, acc.shape[1]))
  for c in range(2):
    for h in range(2):
      for i in range(8):
        r = 8 * c + i
        out[c, h, i] = acc[(r & 3) + 8 * (r >> 2) + 4 * h]
  return out


def mma_tile(stream: Stream, inputs) -> np.ndarray:
  """One output tile: sum over all chunks of all input arrays, in stream order; + bias.  Returns [32, N]."""
  acc = None
  bias = stream.next_bias()
  for chunks in inputs:
    for kc in range(chunks.shape[0]):
      A = stream.next_frag()
      part = np.einsum('mhi,hin->mn', A, chunks[kc])
      acc = part if acc is None else acc + part
  return acc + bias[:, None]


def dense(stream: Stream, inputs, n_tiles: int, relu: bool) -> np.ndarray:
  outs = []
  for _ in range(n_tiles):
    acc = mma_tile(stream, inputs)
    if relu:
      acc = np.maximum(acc, 0.0)
    outs.append(tile_to_chunks(acc))
  return np.concatenate(outs, axis=0)


def head(stream: Stream, inputs, n_out: int) -> np.ndarray:
  """Returns [n_out, N]: logical output j = accumulator register j of the lower half = row (j&3) + 8(j>>2);
  also checks the duplicate in the upper half."""
  acc = mma_tile(stream, inputs)
  out = np.stack([acc[(j & 3) + 8 * (j >> 2)] for j in range(n_out)])
  dup = np.stack([acc[(j & 3) + 8 * (j >> 2) + 4] for j in range(n_out)])
  assert np.array_equal(out, dup), 'head outputs must be duplicated in both lane halves'
  return out


def mlp(stream: Stream, feats: np.ndarray, depth: int, width: int, skip: int) -> np.ndarray:
  in0 = linear_chunks(feats, -(-feats.shape[1] // 16))
  x = None
  for l in range(depth):
    if l == 0:
      x = dense(stream, [in0], width // 32, True)
    elif l == skip:
      x = dense(stream, [x, in0], width // 32, True)
    else:
      x = dense(stream, [x], width // 32, True)
  return x
