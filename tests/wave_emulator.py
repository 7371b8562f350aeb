"""Numpy emulation of how field.h (the fused kernels) consumes the packed MFMA weight stream.

It restates, independently of the C++ packer, the conventions of the device code:
  * fragment = [32 rows m] x [16 k-slots (h, i)], lane = (h << 5) | m; the stream is a sequence of 1-KiB units: a bf16 / f16
    fragment is one unit, a split-bf16 fragment two (hi, lo), an fp32 fragment two (slots i<4, slots i>=4), back to
    back without alignment, so networks of different precision can follow each other in one stream (csrc/graphs.h Plan);
  * the output tiles of a hidden layer come TILE_PAIR = 2 at a time: within a pair the stream holds, segment by segment
    and chunk by chunk, one fragment of each tile (consecutive MFMAs of a wave go to different accumulators); heads
    are single tiles;
  * an output tile's accumulator register r of lane half h is row (r & 3) + 8 (r >> 2) + 4 h, and registers
    8c .. 8c+7 become k-slots (h, 0..7) of activation chunk 2*tile + c of the next layer;
  * linear input chunks: slot (c, h, i) <-> feature 16 c + 8 h + i;
  * heads: logical output j is accumulator register j (rows duplicated over both halves).
Used by the CPU tests to check the packer + stream order against the oracle's MLPs without a GPU.
"""
import numpy as np


PREC_NAMES = ('bf16', 'bf16x3', 'f32', 'f16')     # csrc/graphs.h enum Prec


class Stream:
  def __init__(self, wbytes: np.ndarray, bias: np.ndarray, prec: str):
    """prec: the precision of every fragment unless a call names another one (mixed plans)."""
    self.prec = prec
    self.w = wbytes
    self.bias = bias
    self.pos = 0          # stream position in 1-KiB units
    self.bt = 0

  def next_frag(self, prec=None) -> np.ndarray:
    """Returns A[32 rows][2 halves][8 slots] as float64."""
    prec = prec or self.prec
    prec = 'bf16x3' if prec == 'f16x3' else prec      # (the plan of 'f16x3' is the split plan: csrc/graphs.h plan_of(6))
    units = 1 if prec in ('bf16', 'f16') else 2
    raw = self.w[self.pos * 1024:(self.pos + units) * 1024]
    self.pos += units
    A = np.zeros((32, 2, 8))
    if prec == 'f32':
      p0 = raw[:1024].view(np.float32).reshape(64, 4)
      p1 = raw[1024:].view(np.float32).reshape(64, 4)
      v = np.concatenate([p0, p1], axis=1)
    elif prec == 'f16':
      v = raw.view(np.float16).astype(np.float64).reshape(64, 8)
    elif prec == 'bf16x3' and self.prec == 'f16x3':
      # NERFDS_PREC_F16X3: the split plan's fragments with f16 hi / lo parts (csrc/pack.h StreamWriter::x3_f16; the kernel: field.h NERFDS_X3_F16)
      v = raw[:1024].view(np.float16).astype(np.float64).reshape(64, 8) + raw[1024:].view(np.float16).astype(np.float64).reshape(64, 8)
    else:
      def bf(x):
        return (x.view(np.uint16).astype(np.uint32) << 16).view(np.float32).reshape(64, 8)
      v = bf(raw[:1024])
      if prec == 'bf16x3':
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


TILE_PAIR = 2


def mma_tiles(stream: Stream, inputs, precs=None, tp: int = 1):
  """tp output tiles at once: for every chunk of every input array (stream order), one fragment per tile; + bias.
  Returns tp arrays [32, N].  precs: per input array, the precision of its weight fragments (default: the stream's)."""
  acc = [None] * tp
  bias = [stream.next_bias() for _ in range(tp)]
  for j, chunks in enumerate(inputs):
    for kc in range(chunks.shape[0]):
      for t in range(tp):
        A = stream.next_frag(precs[j] if precs else None)
        part = np.einsum('mhi,hin->mn', A, chunks[kc])
        acc[t] = part if acc[t] is None else acc[t] + part
  return [a + b[:, None] for a, b in zip(acc, bias)]


def mma_tile(stream: Stream, inputs, precs=None) -> np.ndarray:
  return mma_tiles(stream, inputs, precs, 1)[0]


def dense(stream: Stream, inputs, n_tiles: int, relu: bool, precs=None) -> np.ndarray:
  outs = []
  assert n_tiles % TILE_PAIR == 0
  for _ in range(n_tiles // TILE_PAIR):
    for acc in mma_tiles(stream, inputs, precs, TILE_PAIR):
      if relu:
        acc = np.maximum(acc, 0.0)
      outs.append(tile_to_chunks(acc))
  return np.concatenate(outs, axis=0)


def head(stream: Stream, inputs, n_out: int, precs=None) -> np.ndarray:
  """Returns [n_out, N]: logical output j = accumulator register j of the lower half = row (j&3) + 8(j>>2);
  also checks the duplicate in the upper half."""
  acc = mma_tile(stream, inputs, precs)
  out = np.stack([acc[(j & 3) + 8 * (j >> 2)] for j in range(n_out)])
  dup = np.stack([acc[(j & 3) + 8 * (j >> 2) + 4] for j in range(n_out)])
  assert np.array_equal(out, dup), 'head outputs must be duplicated in both lane halves'
  return out


def mlp(stream: Stream, feats: np.ndarray, depth: int, width: int, skip: int, prec=None) -> np.ndarray:
  in0 = linear_chunks(feats, -(-feats.shape[1] // 16))
  x = None
  for l in range(depth):
    if l == 0:
      x = dense(stream, [in0], width // 32, True, [prec])
    elif l == skip:
      x = dense(stream, [x, in0], width // 32, True, [prec, prec])
    else:
      x = dense(stream, [x], width // 32, True, [prec])
  return x
