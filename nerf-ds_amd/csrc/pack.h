// Host-side packer: Flax parameter tree -> MFMA fragment stream (see field.h header).
//
// For every dense layer the stream holds, per output tile of 32 rows and per k16-chunk, one fragment
// laid out lane-linearly: lane l = (h << 5) | m supplies, for output row m of the tile, the weights of
// k-slots (h, i), i = 0..7.  What makes the in-register chaining work is the k-slot -> input-feature map:
//   * "linear" segments (raw network inputs built in-kernel): slot (chunk c, h, i) <-> feature 16c + 8h + i
//   * "tile" segments (a previous layer's output, straight from its accumulators):
//       slot (tile t, chunk c in {0,1}, h, i) <-> feature 32t + 16c + (i & 3) + 8 (i >> 2) + 4h
// and, for output heads, the row -> logical-output map j = (m & 3) + 4 (m >> 3) (each logical output is
// present in both lane halves so no cross-lane traffic is needed to read it).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <vector>

#include "graphs.h"

namespace nerfds {

struct DenseView {
  const float* kernel = nullptr;   // [in][out] row-major (Flax nn.Dense)
  const float* bias = nullptr;     // [out]
  int in_dim = 0, out_dim = 0;
  float W(int r, int c) const { return kernel[(size_t)r * out_dim + c]; }
};

inline uint16_t f32_to_bf16_rne(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
inline float bf16_to_f32(uint16_t b) {
  uint32_t u = (uint32_t)b << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
// fp32 -> IEEE binary16, round to nearest even (what v_cvt_pk_f16_f32 does on the device), subnormals kept.
inline uint16_t f32_to_f16_rne(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
  const uint32_t a = u & 0x7fffffffu;
  if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);                   // NaN
  if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                  // >= 65520 rounds to inf
  if (a < 0x33000001u) return sign;                                          // <= 2^-25 rounds to zero
  const int e = (int)(a >> 23) - 127;
  uint32_t m = (a & 0x7fffffu) | 0x800000u;                                  // 24-bit significand
  int shift = (e < -14) ? (13 + (-14 - e)) : 13;                             // bits dropped
  uint32_t q = m >> shift;
  const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
  if (rem > half || (rem == half && (q & 1u))) ++q;
  // normal: q has the hidden bit at position 10 -> adding (e + 14) << 10 yields the biased exponent; a carry out of
  // the significand bumps the exponent by construction.  subnormal: exponent field 0, q < 2^10 (or exactly 2^10 = min normal).
  const uint32_t h = (e < -14) ? q : (((uint32_t)(e + 14) << 10) + q);
  return (uint16_t)(sign | h);
}

// IEEE binary16 -> fp32, exact (subnormals included)
inline float f16_to_f32(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3ffu;
  uint32_t u;
  if (e == 0) {
    if (m == 0) u = sign;
    else {                                     // subnormal: m * 2^-24
      const float f = (float)m * 5.9604644775390625e-08f;
      std::memcpy(&u, &f, 4);
      u |= sign;
    }
  } else if (e == 31) u = sign | 0x7f800000u | (m << 13);
  else u = sign | ((e + 112u) << 23) | (m << 13);
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

// k-slot -> kernel-row maps -------------------------------------------------------------------------
inline void rows_linear(std::vector<int>& rm, int n_chunks, const std::function<int(int)>& feat_to_row) {
  for (int s = 0; s < 16 * n_chunks; ++s) rm.push_back(feat_to_row(s));
}
inline void rows_tile(std::vector<int>& rm, int width, int row0) {
  for (int t = 0; t < width / 32; ++t)
    for (int c = 0; c < 2; ++c)
      for (int h = 0; h < 2; ++h)
        for (int i = 0; i < 8; ++i) rm.push_back(row0 + 32 * t + 16 * c + (i & 3) + 8 * (i >> 2) + 4 * h);
}
inline int head_col(int m) { return (m & 3) + 4 * (m >> 3); }

struct Seg {
  std::vector<int> rows;                       // per k-slot (chunk*16 + h*8 + i): kernel row or -1
  int prec = P_BF16;                           // arithmetic of this input segment (graphs.h Plan)
};
struct Layer {
  std::vector<Seg> segs;                       // input segments in K order (e.g. [previous layer | raw input] on a skip layer)
  int n_tiles = 0;                             // output tiles of 32 rows
  bool is_head = false;
  int n_out = 0;                               // logical outputs
  std::function<float(int, int)> W;            // (row, col)
  std::function<float(int)> B;                 // (col)
};

struct StreamWriter {
  uint8_t* w;       // may be nullptr: count only
  float* b;
  size_t wbytes = 0, bfloats = 0;
  int tile_pair = TILE_PAIR;   // tiles per group in THIS stream (graphs.h stream_tile_pair: what the consuming kernel was built with)
  bool x3_f16 = false;         // the hi / lo parts of a split (P_BF16X3) fragment are f16, not bf16 (NERFDS_PREC_F16X3: field.h NERFDS_X3_F16)

  void frag_out(const Layer& L, const Seg& S, int ot, int kc) {
    const int prec = S.prec;
    if (w) {
      uint8_t* frag = w + wbytes;
      for (int lane = 0; lane < 64; ++lane) {
        const int m = lane & 31, h = lane >> 5;
        int col = L.is_head ? head_col(m) : 32 * ot + m;
        if (col >= L.n_out) col = -1;
        float v[8];
        for (int i = 0; i < 8; ++i) {
          const int r = S.rows[kc * 16 + h * 8 + i];
          v[i] = (r < 0 || col < 0) ? 0.f : L.W(r, col);
        }
        if (prec == P_BF16 || prec == P_F16) {
          uint16_t* d = reinterpret_cast<uint16_t*>(frag + lane * 16);
          for (int i = 0; i < 8; ++i) d[i] = prec == P_BF16 ? f32_to_bf16_rne(v[i]) : f32_to_f16_rne(v[i]);
        } else if (prec == P_BF16X3) {
          uint16_t* dh = reinterpret_cast<uint16_t*>(frag + lane * 16);
          uint16_t* dl = reinterpret_cast<uint16_t*>(frag + 1024 + lane * 16);
          for (int i = 0; i < 8; ++i) {
            if (x3_f16) {
              dh[i] = f32_to_f16_rne(v[i]);
              dl[i] = f32_to_f16_rne(v[i] - f16_to_f32(dh[i]));
            } else {
              dh[i] = f32_to_bf16_rne(v[i]);
              dl[i] = f32_to_bf16_rne(v[i] - bf16_to_f32(dh[i]));
            }
          }
        } else if (prec == P_BF16X6) {
          uint16_t* dh = reinterpret_cast<uint16_t*>(frag + lane * 16);
          uint16_t* dm = reinterpret_cast<uint16_t*>(frag + 1024 + lane * 16);
          uint16_t* dl = reinterpret_cast<uint16_t*>(frag + 2048 + lane * 16);
          for (int i = 0; i < 8; ++i) {
            dh[i] = f32_to_bf16_rne(v[i]);
            const float r = v[i] - bf16_to_f32(dh[i]);
            dm[i] = f32_to_bf16_rne(r);
            dl[i] = f32_to_bf16_rne(r - bf16_to_f32(dm[i]));
          }
        } else {
          float* da = reinterpret_cast<float*>(frag + lane * 16);
          float* db = reinterpret_cast<float*>(frag + 1024 + lane * 16);
          for (int i = 0; i < 4; ++i) { da[i] = v[i]; db[i] = v[4 + i]; }
        }
      }
    }
    wbytes += frag_bytes(prec);
  }

  // Stream order of a layer: tiles in groups of TILE_PAIR (heads: one tile); within a group, segment by segment and
  // chunk by chunk, one fragment per tile of the group - the order field.h accum consumes them in.
  void emit(const Layer& L) {
    const int tp = (L.n_tiles % tile_pair == 0) ? tile_pair : 1;
    for (int ot = 0; ot < L.n_tiles; ot += tp) {
      for (const Seg& S : L.segs) {
        const int kc_n = (int)S.rows.size() / 16;
        for (int kc = 0; kc < kc_n; ++kc)
          for (int t = 0; t < tp; ++t) frag_out(L, S, ot + t, kc);
      }
      if (b) {
        for (int t = 0; t < tp; ++t)
          for (int m = 0; m < 32; ++m) {
            int col = L.is_head ? head_col(m) : 32 * (ot + t) + m;
            b[bfloats + 32 * t + m] = (col < L.n_out) ? L.B(col) : 0.f;
          }
      }
      bfloats += 32 * tp;
    }
  }
};

inline Layer plain_layer(const DenseView& d, std::vector<Seg> segs, int width) {
  Layer L;
  L.segs = std::move(segs);
  L.n_tiles = width / 32;
  L.n_out = d.out_dim;
  L.W = [d](int r, int c) { return d.W(r, c); };
  L.B = [d](int c) { return d.bias ? d.bias[c] : 0.f; };
  return L;
}
inline Layer head_layer(const DenseView& d, std::vector<int> rows, int prec) {
  Layer L = plain_layer(d, {Seg{std::move(rows), prec}}, 32);
  L.is_head = true;
  return L;
}

// A reference modules.MLP (modules.py:57-83) with `depth` hidden layers of `width`, raw input of `in_dim`
// features in `in_chunks` linear chunks, skip re-concatenation [x, inputs] before layer `skip`.
// Split f16 only (StreamWriter::x3_f16): f16's normal range ends at 6.1e-5, and the output heads of the level-independent networks are initialised
// far below it (hyper sheet N(0, 1e-5), modules.py:367-392; SE3 branches 1e-4, warping.py:156-157): their weights would keep 7 - 10 significand bits.
// A ReLU unit is positively homogeneous - relu(s z) = s relu(z) for s > 0 - so unit j of the LAST hidden layer can carry any power of two s_j: its
// incoming weights and bias times s_j, the head's row j divided by s_j, the function unchanged EXACTLY.  s_j = 2^round((log2 max|out_j| - log2 max|in_j|) / 2)
// balances the two sides (in 0.1 against out 1e-5: both land at 1e-3, inside f16's normal range); a balanced unit keeps s_j = 1 and its bits.
inline std::vector<float> balance_last_hidden(const DenseView& last, std::initializer_list<DenseView> heads, int width) {
  std::vector<float> sc((size_t)width, 1.f);
  for (int j = 0; j < width; ++j) {
    double mi = last.bias ? std::fabs((double)last.bias[j]) : 0.0, mo = 0.0;
    for (int r = 0; r < last.in_dim; ++r) mi = std::max(mi, std::fabs((double)last.W(r, j)));
    for (const DenseView& h : heads)
      for (int c = 0; c < h.out_dim; ++c) mo = std::max(mo, std::fabs((double)h.W(j, c)));
    if (!(mi > 0.0) || !(mo > 0.0) || !std::isfinite(mi) || !std::isfinite(mo)) continue;
    int e = (int)std::lround(0.5 * (std::log2(mo) - std::log2(mi)));
    e = e < -14 ? -14 : (e > 14 ? 14 : e);
    sc[(size_t)j] = std::ldexp(1.f, e);
  }
  return sc;
}
inline Layer scale_outputs(Layer L, std::shared_ptr<std::vector<float>> sc) {      // W(:, c), B(c) times sc[c]
  auto W = L.W; auto B = L.B;
  L.W = [W, sc](int r, int c) { return W(r, c) * (*sc)[(size_t)c]; };
  L.B = [B, sc](int c) { return B(c) * (*sc)[(size_t)c]; };
  return L;
}
inline Layer divide_inputs(Layer L, std::shared_ptr<std::vector<float>> sc) {      // W(r, :) divided by sc[r] (r = the hidden unit's index: rows_tile(.., 0))
  auto W = L.W;
  L.W = [W, sc](int r, int c) { return W(r, c) / (*sc)[(size_t)r]; };
  return L;
}
inline void emit_mlp(StreamWriter& sw, const DenseView* hidden, int depth, int width, int in_dim, int in_chunks, int skip, int prec,
                     std::shared_ptr<std::vector<float>> last_scale = nullptr) {
  for (int l = 0; l < depth; ++l) {
    std::vector<Seg> segs;
    std::vector<int> rows;
    if (l == 0) {
      rows_linear(rows, in_chunks, [&](int s) { return s < in_dim ? s : -1; });
      segs.push_back(Seg{std::move(rows), prec});
    } else {
      rows_tile(rows, width, 0);
      segs.push_back(Seg{std::move(rows), prec});
      if (l == skip) {
        std::vector<int> raw;
        rows_linear(raw, in_chunks, [&](int s) { return s < in_dim ? width + s : -1; });
        segs.push_back(Seg{std::move(raw), prec});
      }
    }
    Layer L = plain_layer(hidden[l], std::move(segs), width);
    sw.emit((l == depth - 1 && last_scale) ? scale_outputs(std::move(L), last_scale) : std::move(L));
  }
}

struct SharedNets {       // views into nerfds_weights
  DenseView mask_hidden[16], mask_out;
  DenseView warp_hidden[16], warp_w, warp_v;
  DenseView hyper_hidden[16], hyper_out;
};
struct NerfNet {
  DenseView trunk[16], bottleneck, alpha, rgb_hidden[16], rgb;
  // Optional: rgb hidden_0 with the bottleneck already folded in, [(TRUNK_W + cond) x RGB_W] in the kernel's K order
  // [trunk_output | viewdir | normal] plus its bias [RGB_W]; pack_nerf then does not fold (the trainer packs parameter INDICES
  // through this packer and folds on the device every step, nerfds_train.cpp build_fused_forward).
  const float* prefolded = nullptr;
  const float* prefolded_bias = nullptr;
};

template <class G> void pack_shared(StreamWriter& sw, const SharedNets& n, Plan pl) {
  using D = Dims<G>;
  if constexpr (G::HAS_MASK) {
    auto sc = (sw.x3_f16 && sw.w) ? std::make_shared<std::vector<float>>(balance_last_hidden(n.mask_hidden[G::MASK_DEPTH - 1], {n.mask_out}, G::MASK_W)) : nullptr;
    emit_mlp(sw, n.mask_hidden, G::MASK_DEPTH, G::MASK_W, D::MASK_IN, D::MASK_KC, G::MASK_SKIP, pl.mask, sc);
    std::vector<int> rows;
    rows_tile(rows, G::MASK_W, 0);
    Layer H = head_layer(n.mask_out, std::move(rows), pl.mask);
    sw.emit(sc ? divide_inputs(std::move(H), sc) : std::move(H));
  }
  if constexpr (G::HAS_WARP) {
    auto sc = (sw.x3_f16 && sw.w) ? std::make_shared<std::vector<float>>(balance_last_hidden(n.warp_hidden[G::WARP_DEPTH - 1], {n.warp_w, n.warp_v}, G::WARP_W)) : nullptr;
    emit_mlp(sw, n.warp_hidden, G::WARP_DEPTH, G::WARP_W, D::WARP_IN, D::WARP_KC, G::WARP_SKIP, pl.warp, sc);
    std::vector<int> rows;
    rows_tile(rows, G::WARP_W, 0);
    Layer L;                                   // merged head: logical outputs 0-2 = w, 3-5 = v (warping.py:217-218)
    L.segs = {Seg{std::move(rows), pl.warp}};
    L.n_tiles = 1;
    L.is_head = true;
    L.n_out = 6;
    const DenseView w = n.warp_w, v = n.warp_v;
    L.W = [w, v](int r, int c) { return c < 3 ? w.W(r, c) : v.W(r, c - 3); };
    L.B = [w, v](int c) { return c < 3 ? (w.bias ? w.bias[c] : 0.f) : (v.bias ? v.bias[c - 3] : 0.f); };
    sw.emit(sc ? divide_inputs(std::move(L), sc) : std::move(L));
  }
  if constexpr (G::HAS_HYPER) {
    auto sc = (sw.x3_f16 && sw.w) ? std::make_shared<std::vector<float>>(balance_last_hidden(n.hyper_hidden[G::HYP_DEPTH - 1], {n.hyper_out}, G::HYP_W)) : nullptr;
    emit_mlp(sw, n.hyper_hidden, G::HYP_DEPTH, G::HYP_W, D::HYP_IN, D::HYP_KC, G::HYP_SKIP, pl.hyp, sc);
    std::vector<int> rows;
    rows_tile(rows, G::HYP_W, 0);
    Layer H = head_layer(n.hyper_out, std::move(rows), pl.hyp);
    sw.emit(sc ? divide_inputs(std::move(H), sc) : std::move(H));
  }
}

template <class G> void pack_nerf(StreamWriter& sw, const NerfNet& n, Plan pl) {
  using D = Dims<G>;
  constexpr int TW = G::TRUNK_W;
  emit_mlp(sw, n.trunk, G::TRUNK_DEPTH, TW, D::TRUNK_IN, D::TRUNK_KC, G::TRUNK_SKIP, pl.trunk);
  {  // alpha head on trunk_output (modules.py:273-274)
    std::vector<int> rows;
    rows_tile(rows, TW, 0);
    sw.emit(head_layer(n.alpha, std::move(rows), pl.trunk));
  }
  {  // rgb hidden_0 with the (activation-free, modules.py:255) bottleneck Dense folded in.
     // Reference (modules.py:296-310): rgb_pre = [bottleneck TW | viewdir 6*VD | trunk_output TW (if X_IN_RGB) | normal 6*NM] @ K + b
     //   with bottleneck = trunk_output @ Wb + bb.  Hence, with F = Wb @ K[bottleneck rows] (+ K[trunk_output rows]):
     //   rgb_pre = trunk_output @ F + [viewdir | normal] @ K[cond rows] + (b + bb @ K[bottleneck rows]).
     // Kernel K order: [trunk_output tiles | cond chunks = viewdir ++ normal]; virtual rows 0..TW-1 = F, TW.. = cond.
    constexpr int VD = D::VD_FEATS, NM = D::NM_FEATS;
    constexpr int row_vd = TW, row_x = TW + VD, row_nm = TW + VD + (G::X_IN_RGB ? TW : 0);
    const DenseView K = n.rgb_hidden[0], B = n.bottleneck;
    const int W = G::RGB_W;
    auto fused = std::make_shared<std::vector<float>>((size_t)(TW + VD + NM) * W);
    auto fbias = std::make_shared<std::vector<float>>(W);
    if (n.prefolded) {
      std::copy(n.prefolded, n.prefolded + fused->size(), fused->begin());
      std::copy(n.prefolded_bias, n.prefolded_bias + W, fbias->begin());
    }
    for (int c = 0; c < W && !n.prefolded; ++c) {
      for (int r = 0; r < TW; ++r) {
        double acc = G::X_IN_RGB ? (double)K.W(row_x + r, c) : 0.0;
        for (int k = 0; k < TW; ++k) acc += (double)B.W(r, k) * (double)K.W(k, c);
        (*fused)[(size_t)r * W + c] = (float)acc;
      }
      for (int q = 0; q < VD; ++q) (*fused)[(size_t)(TW + q) * W + c] = K.W(row_vd + q, c);
      for (int q = 0; q < NM; ++q) (*fused)[(size_t)(TW + VD + q) * W + c] = K.W(row_nm + q, c);
      double bacc = K.bias ? (double)K.bias[c] : 0.0;
      for (int k = 0; k < TW; ++k) bacc += (double)(B.bias ? B.bias[k] : 0.f) * (double)K.W(k, c);
      (*fbias)[c] = (float)bacc;
    }
    std::vector<int> rows, crows;
    rows_tile(rows, TW, 0);
    rows_linear(crows, D::COND_KC, [&](int s) { return s < VD + NM ? TW + s : -1; });
    Layer L;
    L.segs = {Seg{std::move(rows), pl.trunk}, Seg{std::move(crows), pl.rgb}};   // trunk_output is a trunk-precision tensor
    L.n_tiles = W / 32;
    L.n_out = W;
    L.W = [fused, W](int r, int c) { return (*fused)[(size_t)r * W + c]; };
    L.B = [fbias](int c) { return (*fbias)[c]; };
    sw.emit(L);
  }
  {  // rgb head
    std::vector<int> rows;
    rows_tile(rows, G::RGB_W, 0);
    sw.emit(head_layer(n.rgb, std::move(rows), pl.rgb));
  }
}

}  // namespace nerfds
