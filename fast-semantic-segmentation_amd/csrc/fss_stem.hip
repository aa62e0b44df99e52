// First convolution of a network, fed straight from the uint8 frame (gfx950).
//
// The reference pads the frame to a bounding box with zeros, subtracts the
// channel means, (for the low-resolution branch) halves it with an
// align_corners bilinear resize whose scale is exactly 2 - i.e. keeps every
// second pixel - and then runs a 3-channel convolution.  Materialising the
// preprocessed tensor costs more HBM traffic than the convolution itself, so
// this kernel applies pad + mean/scale + subsampling while it stages an input
// patch in LDS and convolves from there on the matrix cores:
//
//   D[cout][pixel] = W[cout][k] * P[k][pixel],   k = tap*4 + channel (channel 3 = 0)
//
// i.e. the weights are the MFMA "A" operand and 16 adjacent output pixels the
// "B" operand, so each lane ends up with consecutive output channels of ONE
// pixel and stores them as one row-contiguous piece.  Blocks are persistent:
// the packed weights are copied to LDS once per block, then the block walks
// over 8x32-pixel output tiles.
#include <algorithm>
#include <type_traits>

#include "fss_common.h"

namespace fss {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

struct StemArgs {
  const uint8_t* x;
  const void* w;
  const float* bias;
  void* y;
  int N, Hr, Wr, sub, H, W, Cout, KH, KW, stride, dil, pad_t, pad_l, Ho, Wo, act, y_ld;
  int K4pad, TW, PH, PW, tiles_x, tiles_y, total_tiles, unit_scale;
  float mean[3], scale[3];
};

// One block iteration makes an output tile of kStemTH rows x TW <= 32 columns from an
// input patch of PH <= 4*kStemRows rows x PW <= 64 columns: a wave fills patch rows
// wave, wave+4, ..., one lane per patch column.
constexpr int kStemTH = 8, kStemTWMax = 32;
constexpr int kStemRows = 6;

template <typename T> struct StemPix;      // one preprocessed pixel in LDS: 4 x T
template <> struct StemPix<bf16_t> {
  static __device__ __forceinline__ void store(char* p, float a, float b, float c, float d) {
    uint2 o;
    o.x = pack_bf16x2(a, b);
    o.y = pack_bf16x2(c, d);
    *reinterpret_cast<uint2*>(p) = o;
  }
};
template <> struct StemPix<f16_t> {
  static __device__ __forceinline__ void store(char* p, float a, float b, float c, float d) {
    uint2 o;
    o.x = pack_f16x2(a, b);
    o.y = pack_f16x2(c, d);
    *reinterpret_cast<uint2*>(p) = o;
  }
};
template <> struct StemPix<float> {
  static __device__ __forceinline__ void store(char* p, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
  }
};

// NJ = 16-channel groups of output channels (cout_pad = 16 * NJ); KS = MFMA k steps of the
// filter when known at compile time (bf16: 7 for 7x7, 2 for 3x3; the tap walk is then
// unrolled with its patch offsets in registers), 0 = any filter
// SPLIT (T = f16_t, TO = f32s_t): the float-quality form.  The patch holds the RAW frame bytes
// as half-precision numbers (exact) and, in channel 3, a 1 wherever the pixel lies inside the
// padded image; mean and scale live in the weights, which the caller packs as
//   W'[co][4 tap + c] = W[tap][c][co] * scale[c],  W'[co][4 tap + 3] = -sum_c W[tap][c][co] * mean[c] * scale[c]
// and hands over as two f16 matrices (high parts [cout_pad][k_pad], then remainders): conv(x) with
// x = (u - mean) * scale inside the image and 0 on the convolution's own padding is linear in
// (u, 1), so nothing of the activation operand is ever rounded, and a product costs two MFMAs
// (lo, hi) instead of the three of a two-sided split.  The output is a split-bfloat16 tensor.
template <typename T, int NJ, int KS, typename TO = T, bool SPLIT = false>
__global__ void __launch_bounds__(256) stem_conv_kernel(const StemArgs a) {
  constexpr int ES = (int)sizeof(T);
  static_assert(!SPLIT || (std::is_same<T, f16_t>::value &&
                           (std::is_same<TO, f32s_t>::value || std::is_same<TO, f16_t>::value)),
                "float-quality mode: f16 operands, split-bfloat16 or half-precision output");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* tapoff = reinterpret_cast<int*>(smem);            // 64 patch offsets (in pixels)
  const int WROW = a.K4pad * ES + 16;                    // +16: rows land on distinct banks
  char* wl = smem + 256;
  char* patch = wl + (SPLIT ? 2 : 1) * NJ * 16 * WROW;
  // float-quality mode: the remainders of the weights lie behind their high parts
  const int WLO = NJ * 16 * WROW;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int p = lane & 15, g = lane >> 4;

  if (tid < 64) {
    const int r = tid / a.KW, s = tid - r * a.KW;
    tapoff[tid] = tid < a.KH * a.KW ? (r * a.dil * a.PW + s * a.dil) : 0;
  }
  // Which output channel accumulator value q of tile nj is in lane group gg (= LDS weight row
  // 16*nj + 4*gg + q).  2-byte outputs: gg*4*NJ + 4*nj + q, so that the 4*NJ values of a lane are
  // consecutive channels (one or two 16-byte stores per lane).  Float outputs: the natural order
  // 16*nj + 4*gg + q, so that one store instruction writes, per pixel, the 64 consecutive bytes of
  // the four lane groups instead of four 16-byte pieces 4*NJ*4 bytes apart.
  constexpr bool NATURAL = std::is_same<TO, float>::value;
  constexpr bool SPLIT_OUT = std::is_same<TO, f32s_t>::value;
  auto chan_of = [](int nj, int gg, int q) { return NATURAL ? 16 * nj + 4 * gg + q : gg * (4 * NJ) + 4 * nj + q; };
  {
    const int cpr = a.K4pad * ES / 16;
    for (int idx = tid; idx < NJ * 16 * cpr; idx += 256) {
      const int rho = idx / cpr, c = idx - rho * cpr;
      const int co = chan_of(rho >> 4, (rho >> 2) & 3, rho & 3);
      *reinterpret_cast<uint4*>(wl + rho * WROW + c * 16) = *reinterpret_cast<const uint4*>(
          reinterpret_cast<const char*>(a.w) + ((size_t)co * a.K4pad * ES + c * 16));
      if constexpr (SPLIT)
        *reinterpret_cast<uint4*>(wl + WLO + rho * WROW + c * 16) = *reinterpret_cast<const uint4*>(
            reinterpret_cast<const char*>(a.w) + ((size_t)(NJ * 16 + co) * a.K4pad * ES + c * 16));
    }
  }
  float bias[NJ][4];
#pragma unroll
  for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int co = chan_of(nj, g, q);
      bias[nj][q] = (a.bias != nullptr && co < a.Cout) ? a.bias[co] : 0.f;
    }
  // (half-precision outputs saturate instead of overflowing to infinity)
  constexpr float kOutMax = std::is_same<TO, f16_t>::value ? 65504.f : INFINITY;
  const float act_lo = a.act == FSS_ACT_NONE ? -kOutMax : 0.f;
  const float act_hi = a.act == FSS_ACT_RELU6 ? 6.f : kOutMax;
  // whole-lane vector stores need every lane's piece 16-byte aligned
  // (split-bf16 output: a lane's 4*NJ consecutive channels must be whole groups of 8)
  const bool vec_ok = (a.y_ld * (int)sizeof(TO)) % (SPLIT_OUT ? 32 : 16) == 0 && (!SPLIT_OUT || NJ % 2 == 0) &&
                      (NATURAL ? 16 * NJ <= a.Cout : g * (4 * NJ) + 4 * NJ <= a.Cout);

  // this wave's four 16-pixel groups: output rows 2*wave, 2*wave+1, columns 0-15 / 16-31.
  // base: patch pixel of tap (0,0) (clamped for columns past the tile, whose results are
  // dropped); ooff: output element offset inside the tile
  int base[4], ooff[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int orow = 2 * wave + (mt >> 1), ocol = (mt & 1) * 16 + p;
    base[mt] = (orow * a.stride) * a.PW + min(ocol, a.TW - 1) * a.stride;
    ooff[mt] = (orow * a.Wo + ocol) * a.y_ld + (NATURAL ? 4 * g : g * (4 * NJ));
  }

  const int tiles_per_img = a.tiles_x * a.tiles_y;
  const int rowstride = a.sub * a.Wr * 3, colstride = a.sub * 3;   // frame bytes per logical step
  // logical rows / columns that are frame pixels (the rest of the image is its zero padding)
  const int hreal = min(a.H, (a.Hr + a.sub - 1) / a.sub), wreal = min(a.W, (a.Wr + a.sub - 1) / a.sub);

  // tile index -> (image, first output row, first output column); advanced by gridDim.x
  // tiles at a time with carries instead of divisions
  struct Tile { int n, ty, tx, oh0, ow0; };
  auto decode = [&](int tile) {
    Tile t;
    t.n = tile / tiles_per_img;
    const int rem = tile - t.n * tiles_per_img;
    t.ty = rem / a.tiles_x;
    t.tx = rem - t.ty * a.tiles_x;
    t.oh0 = t.ty * kStemTH;
    t.ow0 = t.tx * a.TW;
    return t;
  };
  const Tile step = decode(gridDim.x);
  auto advance = [&](Tile t) {
    t.tx += step.tx;
    if (t.tx >= a.tiles_x) { t.tx -= a.tiles_x; ++t.ty; }
    t.ty += step.ty;
    if (t.ty >= a.tiles_y) { t.ty -= a.tiles_y; ++t.n; }
    t.n += step.n;
    t.oh0 = t.ty * kStemTH;
    t.ow0 = t.tx * a.TW;
    return t;
  };

  // Reads the frame bytes of the patch of tile t: slot k of a lane is patch pixel
  // (wave + 4k, lane).  Branch-free: slots that are not frame pixels read byte 0 of the
  // frame and are replaced at commit, so every load is in flight before any is used.
  // a tile whose whole patch lies on frame pixels (and not on the last pixel of a frame row,
  // so that 4-byte loads stay inside the tensor) needs no masks: block-uniform test
  auto interior = [&](const Tile& t) {
    const int i0 = t.oh0 * a.stride - a.pad_t, j0 = t.ow0 * a.stride - a.pad_l;
    return i0 >= 0 && j0 >= 0 && i0 + a.PH <= hreal && j0 + a.PW < wreal;
  };
  auto fetch = [&](const Tile& t, unsigned (&u)[kStemRows]) {
    const int i0 = t.oh0 * a.stride - a.pad_t, j0 = t.ow0 * a.stride - a.pad_l;
    const uint8_t* frame = a.x + (size_t)t.n * a.Hr * a.Wr * 3;
    if (interior(t)) {
      const uint8_t* base = frame + (size_t)i0 * rowstride + (size_t)(j0 + min(lane, a.PW - 1)) * colstride;
#pragma unroll
      for (int k = 0; k < kStemRows; ++k) {
        const int pi = min(wave + 4 * k, a.PH - 1);                  // past the patch: unused
        u[k] = *reinterpret_cast<const unsigned*>(base + (size_t)pi * rowstride);
      }
      return;
    }
    const unsigned last_word = t.n == a.N - 1 ? (unsigned)(a.Hr * a.Wr * 3 - 4) : 0xffffffffu;
    const int j = j0 + lane;
    const bool col_real = lane < a.PW && (unsigned)j < (unsigned)wreal;
    const unsigned coloff = __umul24(j, colstride);       // < 2^31 with the row part: host check
#pragma unroll
    for (int k = 0; k < kStemRows; ++k) {
      const int pi = wave + 4 * k, i = i0 + pi;
      const bool row_real = pi < a.PH && (unsigned)i < (unsigned)hreal;     // wave-uniform
      const unsigned off = (row_real && col_real) ? (unsigned)i * (unsigned)rowstride + coloff : 0u;
      // one (unaligned) 4-byte load per pixel: its 3 bytes + 1 of the next; the very last
      // pixel of the tensor reads the 4 bytes that end with it instead
      const unsigned off4 = min(off, last_word);
      u[k] = *reinterpret_cast<const unsigned*>(frame + off4) >> ((off - off4) * 8u);
    }
  };
  // one pixel into the patch: its three preprocessed channels (float-quality mode: its raw
  // bytes) and `one` in channel 3 (0, or in the float-quality mode 1 inside the padded image)
  auto store_pix = [&](int pix, const float (&v)[3], float one) {
    StemPix<T>::store(patch + (size_t)pix * (4 * ES), v[0], v[1], v[2], one);
  };
  // zero padding of the frame, mean / scale, rounding; stores the patch into LDS
  auto commit = [&](const Tile& t, const unsigned (&u)[kStemRows]) {
    const int i0 = t.oh0 * a.stride - a.pad_t, j0 = t.ow0 * a.stride - a.pad_l;
    if (interior(t)) {
#pragma unroll
      for (int k = 0; k < kStemRows; ++k) {
        const int pi = wave + 4 * k;
        if (pi < a.PH) {                                                    // wave-uniform
          float v[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            v[c] = (float)((u[k] >> (8 * c)) & 255u);
            if constexpr (!SPLIT) {
              v[c] -= a.mean[c];
              if (!a.unit_scale) v[c] *= a.scale[c];
            }
          }
          if (lane < a.PW) store_pix(pi * a.PW + lane, v, SPLIT ? 1.f : 0.f);
        }
      }
      return;
    }
    const int j = j0 + lane;
    const bool col_in = (unsigned)j < (unsigned)a.W, col_real = (unsigned)j < (unsigned)wreal;
#pragma unroll
    for (int k = 0; k < kStemRows; ++k) {
      const int pi = wave + 4 * k, i = i0 + pi;
      if (pi < a.PH) {                                                      // wave-uniform
        const bool row_in = (unsigned)i < (unsigned)a.H, row_real = (unsigned)i < (unsigned)hreal;
        const float m_real = (row_real && col_real) ? 1.f : 0.f;
        const bool inside = row_in && col_in;          // else: zero padding of the convolution
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          if constexpr (SPLIT) {
            v[c] = (float)((u[k] >> (8 * c)) & 255u) * m_real;                     // u, or 0 on the frame's padding
          } else {
            v[c] = fmaf((float)((u[k] >> (8 * c)) & 255u), m_real, -a.mean[c]);    // exact: (u or 0) - mean
            if (!a.unit_scale) v[c] *= a.scale[c];
          }
          v[c] = inside ? v[c] : 0.f;
        }
        if (lane < a.PW) store_pix(pi * a.PW + lane, v, SPLIT && inside ? 1.f : 0.f);
      }
    }
  };

  // The next tile's bytes are fetched into registers before the current tile is computed,
  // so their latency hides behind its MFMAs and stores.
  unsigned ureg[kStemRows];
  Tile cur = decode(blockIdx.x);
  fetch(cur, ureg);
  for (int tile = blockIdx.x; tile < a.total_tiles; tile += gridDim.x) {
    __syncthreads();   // the previous tile's reads of the patch are done (first pass: LDS tables)
    commit(cur, ureg);
    __syncthreads();
    const Tile now = cur;
    // past the last tile this re-reads the current one: harmless, and keeps the loop branch-free
    if (tile + (int)gridDim.x < a.total_tiles) cur = advance(cur);
    fetch(cur, ureg);

    f32x4 acc[4][NJ];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[mt][nj][q] = 0.f;

    if constexpr (sizeof(T) == 2 && KS > 0) {
      // unrolled tap walk: byte offsets of this lane group's two taps per k step
      int bo[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) bo[mt] = base[mt] * 8;
#pragma unroll
      for (int st = 0; st < KS; ++st) {
        const int2 to = *reinterpret_cast<const int2*>(tapoff + 8 * st + 2 * g);
        uint4 wa[NJ];
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj)
          wa[nj] = *reinterpret_cast<const uint4*>(wl + (16 * nj + p) * WROW + (32 * st + 8 * g) * 2);
        uint4 wlo[SPLIT ? NJ : 1];
        if constexpr (SPLIT) {
#pragma unroll
          for (int nj = 0; nj < NJ; ++nj)
            wlo[nj] = *reinterpret_cast<const uint4*>(wl + WLO + (16 * nj + p) * WROW + (32 * st + 8 * g) * 2);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const uint2 b0 = *reinterpret_cast<const uint2*>(patch + (bo[mt] + to.x * 8));
          const uint2 b1 = *reinterpret_cast<const uint2*>(patch + (bo[mt] + to.y * 8));
          const uint4 b = make_uint4(b0.x, b0.y, b1.x, b1.y);
          if constexpr (SPLIT) {
#pragma unroll
            for (int nj = 0; nj < NJ; ++nj) acc[mt][nj] = mfma_16x16x32<T>(wlo[nj], b, acc[mt][nj]);
          }
#pragma unroll
          for (int nj = 0; nj < NJ; ++nj) acc[mt][nj] = mfma_16x16x32<T>(wa[nj], b, acc[mt][nj]);
        }
      }
    } else if constexpr (sizeof(T) == 2) {
      const int ksteps = a.K4pad / 32;
      for (int st = 0; st < ksteps; ++st) {
        const int2 to = *reinterpret_cast<const int2*>(tapoff + 8 * st + 2 * g);
        uint4 wa[NJ];
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj)
          wa[nj] = *reinterpret_cast<const uint4*>(wl + (16 * nj + p) * WROW + (32 * st + 8 * g) * 2);
        uint4 wlo[SPLIT ? NJ : 1];
        if constexpr (SPLIT) {
#pragma unroll
          for (int nj = 0; nj < NJ; ++nj)
            wlo[nj] = *reinterpret_cast<const uint4*>(wl + WLO + (16 * nj + p) * WROW + (32 * st + 8 * g) * 2);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const uint2 b0 = *reinterpret_cast<const uint2*>(patch + (size_t)(base[mt] + to.x) * 8);
          const uint2 b1 = *reinterpret_cast<const uint2*>(patch + (size_t)(base[mt] + to.y) * 8);
          const uint4 b = make_uint4(b0.x, b0.y, b1.x, b1.y);
          if constexpr (SPLIT) {
#pragma unroll
            for (int nj = 0; nj < NJ; ++nj) acc[mt][nj] = mfma_16x16x32<T>(wlo[nj], b, acc[mt][nj]);
          }
#pragma unroll
          for (int nj = 0; nj < NJ; ++nj) acc[mt][nj] = mfma_16x16x32<T>(wa[nj], b, acc[mt][nj]);
        }
      }
    } else {
      const int taps = a.K4pad / 4;
      for (int t = 0; t < taps; ++t) {
        const int to = tapoff[t];
        float wa[NJ];
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj)
          wa[nj] = *reinterpret_cast<const float*>(wl + (16 * nj + p) * WROW + (4 * t + g) * 4);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const float b = *reinterpret_cast<const float*>(patch + ((size_t)(base[mt] + to) * 4 + g) * 4);
#pragma unroll
          for (int nj = 0; nj < NJ; ++nj)
            acc[mt][nj] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[nj], b, acc[mt][nj], 0, 0, 0);
        }
      }
    }

    // lane (p, g) holds channels chan_of(nj, g, 0..3) of pixel p of each group
    TO* __restrict__ ytile = reinterpret_cast<TO*>(a.y) +
                            (((size_t)now.n * a.Ho + now.oh0) * a.Wo + now.ow0) * a.y_ld;
    const int rows_left = a.Ho - now.oh0, cols_left = min(a.TW, a.Wo - now.ow0);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      if (2 * wave + (mt >> 1) >= rows_left || (mt & 1) * 16 + p >= cols_left) continue;
      TO* dst = ytile + ooff[mt];
      float v[4 * NJ];
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          v[4 * nj + q] = __builtin_amdgcn_fmed3f(acc[mt][nj][q] + bias[nj][q], act_lo, act_hi);   // clamp
      if (vec_ok) {
        if constexpr (SPLIT_OUT) {
          if constexpr (NJ % 2 == 0) {
#pragma unroll
            for (int e = 0; e < NJ / 2; ++e) {
              const float v8[8] = {v[8 * e], v[8 * e + 1], v[8 * e + 2], v[8 * e + 3],
                                   v[8 * e + 4], v[8 * e + 5], v[8 * e + 6], v[8 * e + 7]};
              uint4 hi, lo;
              split8(v8, hi, lo);
              store16_nt(dst + 8 * e, hi.x, hi.y, hi.z, hi.w);
              store16_nt(dst + 8 * e + 4, lo.x, lo.y, lo.z, lo.w);
            }
          }
        } else if constexpr (sizeof(TO) == 4) {
#pragma unroll
          for (int nj = 0; nj < NJ; ++nj)
            store16_nt(dst + 16 * nj, __float_as_uint(v[4 * nj]), __float_as_uint(v[4 * nj + 1]),
                       __float_as_uint(v[4 * nj + 2]), __float_as_uint(v[4 * nj + 3]));
        } else {
          unsigned o[2 * NJ];
#pragma unroll
          for (int e = 0; e < 2 * NJ; ++e)
            o[e] = Half16<TO>::pack2(v[2 * e], v[2 * e + 1]);
          if constexpr (NJ % 2 == 0) {
#pragma unroll
            for (int e = 0; e < NJ / 2; ++e)
              store16_nt(dst + 8 * e, o[4 * e], o[4 * e + 1], o[4 * e + 2], o[4 * e + 3]);
          } else {
#pragma unroll
            for (int e = 0; e < NJ; ++e)
              *reinterpret_cast<uint2*>(dst + 4 * e) = make_uint2(o[2 * e], o[2 * e + 1]);
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4 * NJ; ++e) {
          // dst points at the lane's first channel; element e sits (channel - first channel) on
          const int off = chan_of(e >> 2, g, e & 3) - chan_of(0, g, 0);
          if (chan_of(e >> 2, g, e & 3) < a.Cout) ElemTraits<TO>::store(dst + off, v[e]);
        }
      }
    }
  }
}

template <typename T, int NJ, int KS = 0, typename TO = T, bool SPLIT = false>
static int launch_stem(const StemArgs& a, size_t smem, hipStream_t stream) {
  auto kern = stem_conv_kernel<T, NJ, KS, TO, SPLIT>;
  if (smem > 64 * 1024) {
    static size_t attr_bytes = 0;   // per instantiation
    if (smem > attr_bytes) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) {
        set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
        return 1;
      }
      attr_bytes = smem;
    }
  }
  // persistent blocks: as many as are resident on the chip at once, at most one per tile
  int per_cu = 0, dev = 0;
  hipDeviceProp_t prop;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 256,
                                                   smem) != hipSuccess || per_cu < 1)
    per_cu = 1;
  static int num_cu = 0;
  if (num_cu == 0) {
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      set_error("stem_conv: cannot query the device");
      return 1;
    }
    num_cu = prop.multiProcessorCount;
  }
  const int blocks = std::min(a.total_tiles, num_cu * per_cu);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), smem, stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error(std::string("stem_conv launch: ") + hipGetErrorString(e));
    return 3;
  }
  return 0;
}

}  // namespace fss

using namespace fss;

extern "C" int fss_stem_pack_dims(int kh, int kw, int cout, int dtype, int* k_pad, int* cout_pad) {
  FSS_REQUIRE(k_pad && cout_pad, "null argument");
  // (FSS_F16: the first layer of fss_stem2_conv_u8; same dimensions as bfloat16)
  FSS_REQUIRE(dtype == FSS_F32 || dtype == FSS_BF16 || dtype == FSS_F32_BF16X3 || dtype == FSS_F16 ||
                  dtype == FSS_F16_W2, "bad dtype");
  FSS_REQUIRE(kh > 0 && kw > 0 && kh * kw <= 64 && cout > 0 && cout <= 64, "bad shape");
  const int k4 = kh * kw * 4;
  *k_pad = dtype != FSS_F32 ? (k4 + 31) / 32 * 32 : k4;
  *cout_pad = (cout + 15) / 16 * 16;
  return 0;
}

extern "C" int fss_stem_conv_u8(const uint8_t* x, const void* w_packed, const float* bias, void* y,
                                int n, int hr, int wr, int sub, int h, int w, const float* mean3,
                                const float* scale3, int cout, int kh, int kw, int stride,
                                int dilation, int pad_top, int pad_left, int ho, int wo, int act,
                                int dtype, int y_ld, void* stream) {
  FSS_REQUIRE(x && w_packed && y && mean3 && scale3, "null argument");
  FSS_REQUIRE(dtype == FSS_F32 || dtype == FSS_BF16 || dtype == FSS_F32_BF16X3 || dtype == FSS_F16_W2, "bad dtype");
  FSS_REQUIRE(n > 0 && hr > 0 && wr > 0 && h > 0 && w > 0 && ho > 0 && wo > 0, "bad shape");
  FSS_REQUIRE(sub >= 1 && sub <= 8, "sub must be in [1, 8]");
  FSS_REQUIRE(kh > 0 && kw > 0 && kh * kw <= 64, "filter must have at most 64 taps");
  FSS_REQUIRE((long long)hr * wr * 3 >= 4, "frame too small");
  FSS_REQUIRE(cout > 0 && cout <= 64, "cout must be in [1, 64]");
  if (dtype == FSS_F32_BF16X3)
    FSS_REQUIRE(cout % 8 == 0 && y_ld % 8 == 0 && ((uintptr_t)y & 31) == 0,
                "split-bf16 output: cout and y_ld in multiples of 8, y 32-byte aligned");
  FSS_REQUIRE(stride > 0 && stride <= 4 && dilation > 0 && dilation <= 4, "bad stride / dilation");
  FSS_REQUIRE(pad_top >= 0 && pad_left >= 0, "negative padding");
  FSS_REQUIRE(act == FSS_ACT_NONE || act == FSS_ACT_RELU || act == FSS_ACT_RELU6, "bad act");
  FSS_REQUIRE(y_ld >= cout, "y_ld < cout");
  FSS_REQUIRE((long long)(ho - 1) * stride - pad_top < h && (long long)(wo - 1) * stride - pad_left < w,
              "output size exceeds the input");
  FSS_REQUIRE(((uintptr_t)w_packed & 15) == 0 && ((uintptr_t)y & 15) == 0, "unaligned pointer");
  FSS_REQUIRE((long long)hr * wr * 3 < 0x7fffffffLL && (long long)sub * wr * 3 < (1 << 24) && hr < (1 << 24),
              "frame too large");
  StemArgs a;
  a.x = x; a.w = w_packed; a.bias = bias; a.y = y;
  a.N = n; a.Hr = hr; a.Wr = wr; a.sub = sub; a.H = h; a.W = w; a.Cout = cout; a.KH = kh; a.KW = kw;
  a.stride = stride; a.dil = dilation; a.pad_t = pad_top; a.pad_l = pad_left; a.Ho = ho; a.Wo = wo;
  a.act = act; a.y_ld = y_ld;
  int k_pad = 0, cout_pad = 0;
  if (fss_stem_pack_dims(kh, kw, cout, dtype, &k_pad, &cout_pad)) return 2;
  a.K4pad = k_pad;
  // tile width: as many columns (at most 32) as keep the patch within 64 columns
  const int kext_w = (kw - 1) * dilation + 1, kext_h = (kh - 1) * dilation + 1;
  FSS_REQUIRE(kext_w <= 64 && (kStemTH - 1) * stride + kext_h <= 4 * kStemRows,
              "filter footprint too large: (kw-1)*dilation + 1 must be at most 64 and "
              "7*stride + (kh-1)*dilation + 1 at most 24");
  a.TW = std::min(kStemTWMax, (64 - kext_w) / stride + 1);
  a.PH = (kStemTH - 1) * stride + kext_h;
  a.PW = (a.TW - 1) * stride + kext_w;
  a.tiles_x = ceil_div(wo, a.TW);
  a.tiles_y = ceil_div(ho, kStemTH);
  const long long total = (long long)a.tiles_x * a.tiles_y * n;
  FSS_REQUIRE(total < 0x7fffffffLL, "too many tiles");
  a.total_tiles = (int)total;
  for (int i = 0; i < 3; ++i) { a.mean[i] = mean3[i]; a.scale[i] = scale3[i]; }
  a.unit_scale = scale3[0] == 1.f && scale3[1] == 1.f && scale3[2] == 1.f;
  const bool raw = dtype == FSS_F32_BF16X3 || dtype == FSS_F16_W2;     // the float-quality form
  const int es = dtype == FSS_F32 ? 4 : 2, wplanes = raw ? 2 : 1;
  const size_t smem = 256 + wplanes * (size_t)cout_pad * (k_pad * es + 16) + (size_t)a.PH * a.PW * 4 * es;
  FSS_REQUIRE(smem <= 160 * 1024, "filter too large for LDS");
  hipStream_t s = (hipStream_t)stream;
  const int nj = cout_pad / 16;
#define FSS_STEM(T)                                              \
  switch (nj) {                                                  \
    case 1: return launch_stem<T, 1>(a, smem, s);                \
    case 2: return launch_stem<T, 2>(a, smem, s);                \
    case 3: return launch_stem<T, 3>(a, smem, s);                \
    default: return launch_stem<T, 4>(a, smem, s);               \
  }
  if (dtype == FSS_F32) { FSS_STEM(float) }
  if (dtype == FSS_F32_BF16X3) {
    if (nj == 4 && k_pad == 224) return launch_stem<f16_t, 4, 7, f32s_t, true>(a, smem, s);
    if (nj == 2 && k_pad == 64) return launch_stem<f16_t, 2, 2, f32s_t, true>(a, smem, s);
    if (nj == 4 && k_pad == 64) return launch_stem<f16_t, 4, 2, f32s_t, true>(a, smem, s);
    switch (nj) {
      case 1: return launch_stem<f16_t, 1, 0, f32s_t, true>(a, smem, s);
      case 2: return launch_stem<f16_t, 2, 0, f32s_t, true>(a, smem, s);
      case 3: return launch_stem<f16_t, 3, 0, f32s_t, true>(a, smem, s);
      default: return launch_stem<f16_t, 4, 0, f32s_t, true>(a, smem, s);
    }
  }
  if (dtype == FSS_F16_W2) {
    if (nj == 4 && k_pad == 224) return launch_stem<f16_t, 4, 7, f16_t, true>(a, smem, s);
    switch (nj) {
      case 1: return launch_stem<f16_t, 1, 0, f16_t, true>(a, smem, s);
      case 2: return launch_stem<f16_t, 2, 0, f16_t, true>(a, smem, s);
      case 3: return launch_stem<f16_t, 3, 0, f16_t, true>(a, smem, s);
      default: return launch_stem<f16_t, 4, 0, f16_t, true>(a, smem, s);
    }
  }
  // the two filters of the reference's networks: unrolled tap walk
  if (nj == 4 && k_pad == 224) return launch_stem<bf16_t, 4, 7>(a, smem, s);
  if (nj == 2 && k_pad == 64) return launch_stem<bf16_t, 2, 2>(a, smem, s);
  if (nj == 4 && k_pad == 64) return launch_stem<bf16_t, 4, 2>(a, smem, s);
  FSS_STEM(bf16_t)
#undef FSS_STEM
}
