// Bandwidth-bound kernels of the segmentation forward pass (gfx950).
//
// All tensors are NHWC.  Kernels are written per 16-byte channel chunk: lane i
// of a wave touches chunk i of a pixel row, so a wave reads / writes 1 KiB of
// consecutive bytes whenever a pixel has >= 64 chunks, and whole pixels
// otherwise.  Arithmetic is float regardless of the storage dtype.
#include <algorithm>
#include <mutex>

#include "fss_common.h"

namespace fss {

static constexpr int kBlock = 256;
static inline unsigned grid_for(size_t work) {
  size_t g = (work + kBlock - 1) / kBlock;
  return (unsigned)(g < 1 ? 1 : g);
}

// ------------------------------------------------------------- legacy bilinear
// TF 1.x resize_bilinear source coordinate: in = out_index * scale (float32),
// lower = floor(in), upper = min(lower + 1, size - 1), lerp = in - lower.
__device__ __forceinline__ void interp_coord(int o, float scale, int in_size, int& lo, int& hi,
                                             float& lerp) {
  const float src = (float)o * scale;
  const float fl = floorf(src);
  lo = min((int)fl, in_size - 1);
  hi = min(lo + 1, in_size - 1);
  lerp = src - fl;
}
static inline float resize_scale(int in_size, int out_size, int align_corners) {
  if (align_corners && out_size > 1) return (float)(in_size - 1) / (float)(out_size - 1);
  return (float)in_size / (float)out_size;
}

// ------------------------------------------------------------------ max pool
template <typename T>
__global__ void maxpool_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W,
                               int C, int k, int stride, int pad_t, int pad_l, int Ho, int Wo) {
  constexpr int E = Chunk<T>::N;
  const int cc = C / E;
  const size_t total = (size_t)N * Ho * Wo * cc;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int ch = (int)(idx % cc);
  size_t p = idx / cc;
  const int ow = (int)(p % Wo); p /= Wo;
  const int oh = (int)(p % Ho);
  const int n = (int)(p / Ho);
  Chunk<T> best;
#pragma unroll
  for (int e = 0; e < E; ++e) best.v[e] = -INFINITY;
  for (int r = 0; r < k; ++r) {
    const int ih = oh * stride - pad_t + r;
    if ((unsigned)ih >= (unsigned)H) continue;
    for (int s = 0; s < k; ++s) {
      const int iw = ow * stride - pad_l + s;
      if ((unsigned)iw >= (unsigned)W) continue;
      Chunk<T> v;
      v.load(x + (((size_t)n * H + ih) * W + iw) * C + ch * E);
#pragma unroll
      for (int e = 0; e < E; ++e) best.v[e] = fmaxf(best.v[e], v.v[e]);
    }
  }
  best.store_nt(y + idx * E);
}

// ----------------------------------------------------------------- subsample
template <typename T>
__global__ void subsample_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W,
                                 int C, int stride, int Ho, int Wo) {
  constexpr int E = Chunk<T>::N;
  const int cc = C / E;
  const size_t total = (size_t)N * Ho * Wo * cc;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int ch = (int)(idx % cc);
  size_t p = idx / cc;
  const int ow = (int)(p % Wo); p /= Wo;
  const int oh = (int)(p % Ho);
  const int n = (int)(p / Ho);
  const uint4* src = reinterpret_cast<const uint4*>(
      x + (((size_t)n * H + (size_t)oh * stride) * W + (size_t)ow * stride) * C + ch * E);
  uint4* dst = reinterpret_cast<uint4*>(y + idx * E);
#pragma unroll
  for (int q = 0; q < ChunkBytes<T>::value / 16; ++q) dst[q] = src[q];
}

// ------------------------------------------------------------ avg pool (bins)
// One block per (image, bin); threadIdx.x walks channel chunks, threadIdx.y
// splits the window's pixels; partial sums meet in LDS.
template <typename T, int TX, int TY>
__global__ void avgpool_bins_kernel(const T* __restrict__ x, T* __restrict__ y, int H, int W, int C,
                                    int kh, int kw, int bh, int bw) {
  constexpr int E = Chunk<T>::N;
  __shared__ float part[TY][TX * E];
  const int bin = blockIdx.x;
  const int bx = bin % bw;
  const int by = (bin / bw) % bh;
  const int n = bin / (bw * bh);
  const int cc = C / E;
  const int npix = kh * kw;
  const float inv = 1.f / (float)npix;
  for (int ch0 = blockIdx.y * TX; ch0 < cc; ch0 += gridDim.y * TX) {
    const int ch = ch0 + threadIdx.x;
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
    if (ch < cc) {
      for (int p = threadIdx.y; p < npix; p += TY) {
        const int ih = by * kh + p / kw;
        const int iw = bx * kw + p % kw;
        Chunk<T> v;
        v.load(x + (((size_t)n * H + ih) * W + iw) * C + ch * E);
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] += v.v[e];
      }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) part[threadIdx.y][threadIdx.x * E + e] = acc[e];
    __syncthreads();
    if (threadIdx.y == 0 && ch < cc) {
      Chunk<T> o;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < TY; ++t) s += part[t][threadIdx.x * E + e];
        o.v[e] = s * inv;
      }
      o.store(y + (((size_t)n * bh + by) * bw + bx) * C + ch * E);
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------ bilinear resize
template <typename T, bool VEC>
__global__ void resize_bilinear_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H,
                                       int W, int C, int Ho, int Wo, int y_ld, float sh,
                                       float sw) {
  constexpr int E = VEC ? Chunk<T>::N : 1;
  const int cc = C / E;
  const size_t total = (size_t)N * Ho * Wo * cc;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int ch = (int)(idx % cc);
  size_t p = idx / cc;
  const int ow = (int)(p % Wo); p /= Wo;
  const int oh = (int)(p % Ho);
  const int n = (int)(p / Ho);
  int y0, y1, x0, x1;
  float ly, lx;
  interp_coord(oh, sh, H, y0, y1, ly);
  interp_coord(ow, sw, W, x0, x1, lx);
  const T* b = x + (size_t)n * H * W * C + ch * E;
  T* out = y + (((size_t)n * Ho + oh) * Wo + ow) * y_ld + ch * E;
  if constexpr (VEC) {
    Chunk<T> tl, tr, bl, br, o;
    tl.load(b + ((size_t)y0 * W + x0) * C);
    tr.load(b + ((size_t)y0 * W + x1) * C);
    bl.load(b + ((size_t)y1 * W + x0) * C);
    br.load(b + ((size_t)y1 * W + x1) * C);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float top = tl.v[e] + (tr.v[e] - tl.v[e]) * lx;
      const float bot = bl.v[e] + (br.v[e] - bl.v[e]) * lx;
      o.v[e] = top + (bot - top) * ly;
    }
    o.store_nt(out);
  } else {
    const float tl = ElemTraits<T>::load(b + ((size_t)y0 * W + x0) * C);
    const float tr = ElemTraits<T>::load(b + ((size_t)y0 * W + x1) * C);
    const float bl = ElemTraits<T>::load(b + ((size_t)y1 * W + x0) * C);
    const float br = ElemTraits<T>::load(b + ((size_t)y1 * W + x1) * C);
    const float top = tl + (tr - tl) * lx;
    const float bot = bl + (br - bl) * lx;
    ElemTraits<T>::store(out, top + (bot - top) * ly);
  }
}

// ---------------------------------------------------- pyramid upsample + add
struct PyramidArgs {
  const void* p[8];
  int bh[8], bw[8];
  float sh[8], sw[8];
  int levels;
};

// One block per (image, output row, group of kPyrPix pixels); threads run over the
// 16-byte channel chunks, so all coordinates are block-uniform.  Per level the two bin
// rows are blended vertically once per column pair (v0, v1 - v0), which adjacent pixels
// share, and each pixel adds v0 + (v1 - v0) * lx.  (The reference blends horizontally
// first; the two orders agree to float rounding.)
constexpr int kPyrPix = 8;
template <typename T>
__global__ void pyramid_add_kernel(const T* __restrict__ x, T* __restrict__ y, PyramidArgs a, int N,
                                   int H, int W, int C) {
  constexpr int E = Chunk<T>::N;
  const int cc = C / E;
  const int groups = (W + kPyrPix - 1) / kPyrPix;
  int b = blockIdx.x;
  const int g = b % groups; b /= groups;
  const int oh = b % H;
  const int n = b / H;
  const int ow0 = g * kPyrPix;
  const size_t row_base = (((size_t)n * H + oh) * W + ow0) * C;
  for (int ch = threadIdx.x; ch < cc; ch += blockDim.x) {
    Chunk<T> acc[kPyrPix];
#pragma unroll
    for (int k = 0; k < kPyrPix; ++k)
      if (ow0 + k < W) {
        if (x != nullptr) {
          acc[k].load(x + row_base + (size_t)k * C + (size_t)ch * E);
        } else {
#pragma unroll
          for (int e = 0; e < E; ++e) acc[k].v[e] = 0.f;
        }
      }
    for (int l = 0; l < a.levels; ++l) {
      const int bh = a.bh[l], bw = a.bw[l];
      int y0, y1;
      float ly;
      interp_coord(oh, a.sh[l], bh, y0, y1, ly);
      const T* pb = reinterpret_cast<const T*>(a.p[l]) + (size_t)n * bh * bw * C + (size_t)ch * E;
      const T* r0 = pb + (size_t)y0 * bw * C;
      const T* r1 = pb + (size_t)y1 * bw * C;
      int px0 = -1, px1 = -1;
      float v0[E], dv[E];
#pragma unroll
      for (int k = 0; k < kPyrPix; ++k) {
        if (ow0 + k < W) {
          int x0, x1;
          float lx;
          interp_coord(ow0 + k, a.sw[l], bw, x0, x1, lx);
          if (x0 != px0 || x1 != px1) {             // block-uniform
            Chunk<T> tl, tr, bl, br;
            tl.load(r0 + (size_t)x0 * C);
            bl.load(r1 + (size_t)x0 * C);
            tr.load(r0 + (size_t)x1 * C);
            br.load(r1 + (size_t)x1 * C);
#pragma unroll
            for (int e = 0; e < E; ++e) {
              v0[e] = tl.v[e] + (bl.v[e] - tl.v[e]) * ly;
              dv[e] = tr.v[e] + (br.v[e] - tr.v[e]) * ly - v0[e];
            }
            px0 = x0; px1 = x1;
          }
#pragma unroll
          for (int e = 0; e < E; ++e) acc[k].v[e] += v0[e] + dv[e] * lx;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < kPyrPix; ++k)
      if (ow0 + k < W) acc[k].store_nt(y + row_base + (size_t)k * C + (size_t)ch * E);
  }
}

// ------------------------------------------------------------------- add+act
template <typename T>
__global__ void add_act_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y,
                               size_t chunks, int act) {
  constexpr int E = Chunk<T>::N;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= chunks) return;
  Chunk<T> va, vb;
  va.load(a + idx * E);
  vb.load(b + idx * E);
#pragma unroll
  for (int e = 0; e < E; ++e) va.v[e] = apply_act(va.v[e] + vb.v[e], act);
  va.store(y + idx * E);
}

// ------------------------------------------------------------- to half precision
template <typename T>
__global__ void to_half_kernel(const T* __restrict__ x, f16_t* __restrict__ y, size_t chunks) {
  constexpr int E = Chunk<T>::N;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= chunks) return;
  Chunk<T> v;
  v.load(x + idx * E);
  if constexpr (E == 8) {
    Chunk<f16_t> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.v[e] = v.v[e];
    o.store_nt(y + idx * 8);
  } else {
    *reinterpret_cast<uint2*>(y + idx * 4) =
        make_uint2(Chunk<f16_t>::sat2(v.v[0], v.v[1]), Chunk<f16_t>::sat2(v.v[2], v.v[3]));
  }
}

// ------------------------------------- x2 zoom behind a convolution (border pieces)
// conv3x3 with dilation 2 of an align_corners x2 zoom equals the x2 zoom of the plain conv3x3
// of the source, except on output rows / columns 1 and 2H-3 / 2W-3: there one of the two
// (four) source results that are averaged must not see the border row (column) of the source,
// because the tap that reaches it lands on the zero padding of the zoomed map.  These results
// come from convolutions of small crops of the source in which that row / column is zero:
//   tb: [2n][3][W][C]  top (rows 0,1,2; row 0 zero) and bottom (rows H-3..H-1; last zero)
//   lr: [2n][H][3][C]  left (columns 0,1,2; column 0 zero) and right
//   cn: [4n][3][3][C]  corners TL, TR, BL, BR with both the border row and column zero
template <typename T>
__global__ void zoom2_border_gather_kernel(const T* __restrict__ x, T* __restrict__ tb, T* __restrict__ lr,
                                           T* __restrict__ cn, int N, int H, int W, int C) {
  constexpr int E = Chunk<T>::N;
  const int cc = C / E;
  const size_t n_tb = (size_t)2 * N * 3 * W * cc, n_lr = (size_t)2 * N * H * 3 * cc, n_cn = (size_t)4 * N * 9 * cc;
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_tb + n_lr + n_cn) return;
  T* dst;
  int f, sy, sx;          // frame, source row, source column
  bool zero;
  int ch;
  if (idx < n_tb) {
    dst = tb + idx * E;
    ch = (int)(idx % cc); idx /= cc;
    const int xw = (int)(idx % W); idx /= W;
    const int r = (int)(idx % 3); idx /= 3;
    const int side = (int)(idx / N);
    f = (int)(idx % N);
    sy = side == 0 ? r : H - 3 + r;
    sx = xw;
    zero = side == 0 ? r == 0 : r == 2;
  } else if (idx < n_tb + n_lr) {
    idx -= n_tb;
    dst = lr + idx * E;
    ch = (int)(idx % cc); idx /= cc;
    const int c3 = (int)(idx % 3); idx /= 3;
    const int yh = (int)(idx % H); idx /= H;
    const int side = (int)(idx / N);
    f = (int)(idx % N);
    sy = yh;
    sx = side == 0 ? c3 : W - 3 + c3;
    zero = side == 0 ? c3 == 0 : c3 == 2;
  } else {
    idx -= n_tb + n_lr;
    dst = cn + idx * E;
    ch = (int)(idx % cc); idx /= cc;
    const int c3 = (int)(idx % 3); idx /= 3;
    const int r = (int)(idx % 3); idx /= 3;
    const int k = (int)(idx / N);      // 0 TL, 1 TR, 2 BL, 3 BR
    f = (int)(idx % N);
    const bool bottom = k >= 2, right = (k & 1) != 0;
    sy = bottom ? H - 3 + r : r;
    sx = right ? W - 3 + c3 : c3;
    zero = (bottom ? r == 2 : r == 0) || (right ? c3 == 2 : c3 == 0);
  }
  Chunk<T> v;
  if (zero) {
#pragma unroll
    for (int e = 0; e < E; ++e) v.v[e] = 0.f;
  } else {
    v.load(x + (((size_t)f * H + sy) * W + sx) * C + (size_t)ch * E);
  }
  v.store(dst);
}

// r[2H-1][2W-1] = the x2 zoom of y[H][W] with the border results substituted as described
// above (ytb [2n][W], ylr [2n][H], ycn [4n] = the middle row / column / pixel of the
// convolutions of tb / lr / cn: the caller convolves them without padding across the 3).
template <typename T>
__global__ void zoom2_conv_commute_kernel(const T* __restrict__ y, const T* __restrict__ ytb,
                                          const T* __restrict__ ylr, const T* __restrict__ ycn,
                                          T* __restrict__ r, int N, int H, int W, int C) {
  constexpr int E = Chunk<T>::N;
  const int cc = C / E;
  const int Ho = 2 * H - 1, Wo = 2 * W - 1;
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)N * Ho * Wo * cc) return;
  T* dst = r + idx * E;
  const int ch = (int)(idx % cc); idx /= cc;
  const int q = (int)(idx % Wo); idx /= Wo;
  const int p = (int)(idx % Ho);
  const int f = (int)(idx / Ho);
  const int i0 = p >> 1, j0 = q >> 1, ni = 1 + (p & 1), nj = 1 + (q & 1);
  float acc[E];
#pragma unroll
  for (int e = 0; e < E; ++e) acc[e] = 0.f;
  for (int a = 0; a < ni; ++a)
    for (int b = 0; b < nj; ++b) {
      const int i = i0 + a, j = j0 + b;
      const int rf = (p == 1 && i == 1) ? 1 : (p == Ho - 2 && i == H - 2) ? 2 : 0;   // 1 top, 2 bottom
      const int cf = (q == 1 && j == 1) ? 1 : (q == Wo - 2 && j == W - 2) ? 2 : 0;   // 1 left, 2 right
      const T* src;
      if (rf == 0 && cf == 0) src = y + (((size_t)f * H + i) * W + j) * C;
      else if (cf == 0) src = ytb + (((size_t)(rf - 1) * N + f) * W + j) * C;
      else if (rf == 0) src = ylr + (((size_t)(cf - 1) * N + f) * H + i) * C;
      else src = ycn + ((size_t)((rf - 1) * 2 + (cf - 1)) * N + f) * C;
      Chunk<T> v;
      v.load(src + (size_t)ch * E);
#pragma unroll
      for (int e = 0; e < E; ++e) acc[e] += v.v[e];
    }
  const float sc = 1.0f / (float)(ni * nj);
  Chunk<T> o;
#pragma unroll
  for (int e = 0; e < E; ++e) o.v[e] = acc[e] * sc;
  o.store_nt(dst);
}

// ------------------------------------------------------------- channel copy
template <typename T>
__global__ void copy_channels_kernel(const T* __restrict__ x, int x_ld, T* __restrict__ y, int y_ld,
                                     size_t pixels, int C) {
  constexpr int E = Chunk<T>::N;
  const int cc = C / E;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= pixels * cc) return;
  const int ch = (int)(idx % cc);
  const size_t p = idx / cc;
  const uint4* src = reinterpret_cast<const uint4*>(x + p * x_ld + ch * E);
  uint4* dst = reinterpret_cast<uint4*>(y + p * y_ld + ch * E);
#pragma unroll
  for (int q = 0; q < ChunkBytes<T>::value / 16; ++q) dst[q] = src[q];
}

// ------------------------------------------------------- pyramid average pooling
// All bin-average maps of a pyramid pooling module in one pass over the input.
// Stage 1: one thread per (image row, 16-byte channel chunk) walks the row once and
// emits, for every level, the sums of its column bins (float).  Stage 2 adds the kh rows
// of each bin, scales and rounds.  The row sums live in a caller-provided workspace.
constexpr int kPyrLevels = 4;
constexpr int kPyrPieces = 32;   // column pieces of a row (<= sum of the levels' bin columns)
constexpr int kPyrBinCols = 16;  // bin columns of one level
struct PyrArgs {
  int L, total_cols, total_bins;
  int kh[kPyrLevels], kw[kPyrLevels], bh[kPyrLevels], bw[kPyrLevels];
  int col_off[kPyrLevels], bin_off[kPyrLevels];
  void* y[kPyrLevels];
  // column pieces: the intervals between consecutive bin edges of all levels; bin column bx
  // of level l is the pieces plo[l][bx] .. plo[l][bx + 1] - 1
  int P, edge[kPyrPieces + 1];
  unsigned char plo[kPyrLevels][kPyrBinCols + 1];
};

template <typename T>
__global__ void pyramid_rowsum_kernel(const T* __restrict__ x, float* __restrict__ part, int N, int H,
                                      int W, int C, PyrArgs a) {
  constexpr int E = Chunk<T>::N;
  const int cc = C / E;
  const size_t total = (size_t)N * H * cc;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int ch = (int)(idx % cc);
  const size_t nr = idx / cc;                       // n * H + r
  const T* row = x + nr * W * C + (size_t)ch * E;
  float* out = part + nr * a.total_cols * C + (size_t)ch * E;
  float acc[kPyrLevels][E];
  int cnt[kPyrLevels], bin[kPyrLevels];
#pragma unroll
  for (int l = 0; l < kPyrLevels; ++l) {
    cnt[l] = bin[l] = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) acc[l][e] = 0.f;
  }
#pragma unroll 4
  for (int col = 0; col < W; ++col) {
    Chunk<T> v;
    v.load(row + (size_t)col * C);
#pragma unroll
    for (int l = 0; l < kPyrLevels; ++l) {
      if (l < a.L && bin[l] < a.bw[l]) {            // uniform over the block
#pragma unroll
        for (int e = 0; e < E; ++e) acc[l][e] += v.v[e];
        if (++cnt[l] == a.kw[l]) {
          float* o = out + (size_t)(a.col_off[l] + bin[l]) * C;
#pragma unroll
          for (int e = 0; e < E; e += 4)
            *reinterpret_cast<float4*>(o + e) =
                make_float4(acc[l][e], acc[l][e + 1], acc[l][e + 2], acc[l][e + 3]);
#pragma unroll
          for (int e = 0; e < E; ++e) acc[l][e] = 0.f;
          cnt[l] = 0;
          ++bin[l];
        }
      }
    }
  }
}

// Same first stage with more parallelism (one thread per row, column PIECE and channel
// chunk): the row is cut at every bin edge of every level, so each piece belongs to exactly
// one bin column per level and stage 2 adds whole pieces.  Used when rows x chunks alone do
// not fill the chip (small batches).
template <typename T>
__global__ void pyramid_piecesum_kernel(const T* __restrict__ x, float* __restrict__ part, int N, int H,
                                        int W, int C, PyrArgs a) {
  constexpr int E = Chunk<T>::N;
  const int cc = C / E;
  const size_t total = (size_t)N * H * a.P * cc;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int ch = (int)(idx % cc);
  size_t q = idx / cc;
  const int piece = (int)(q % a.P);
  const size_t nr = q / a.P;                        // n * H + r
  const T* row = x + nr * W * C + (size_t)ch * E;
  float acc[E];
#pragma unroll
  for (int e = 0; e < E; ++e) acc[e] = 0.f;
  for (int col = a.edge[piece]; col < a.edge[piece + 1]; ++col) {
    Chunk<T> v;
    v.load(row + (size_t)col * C);
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] += v.v[e];
  }
  float* o = part + (nr * a.P + piece) * C + (size_t)ch * E;
#pragma unroll
  for (int e = 0; e < E; e += 4)
    *reinterpret_cast<float4*>(o + e) = make_float4(acc[e], acc[e + 1], acc[e + 2], acc[e + 3]);
}

// Stage 2 of the small-problem form: kPyrRS adjacent lanes share one (bin, channel chunk),
// each adds every kPyrRS-th row of the bin, and a butterfly over the lanes finishes the sum
// (the one-bin level would otherwise walk all rows x pieces in a single thread).
constexpr int kPyrRS = 8;
template <typename T>
__global__ void pyramid_piecebin_kernel(const float* __restrict__ part, int N, int H, int C, PyrArgs a) {
  constexpr int E = Chunk<T>::N;
  const int cc = C / E;
  const size_t total = (size_t)N * a.total_bins * cc * kPyrRS;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = idx < total;                    // whole groups of kPyrRS lanes are live or not
  const size_t id = live ? idx : 0;
  const int rs = (int)(id % kPyrRS);
  size_t p = id / kPyrRS;
  const int ch = (int)(p % cc);
  p /= cc;
  const int b = (int)(p % a.total_bins);
  const int n = (int)(p / a.total_bins);
  int l = 0;
#pragma unroll
  for (int k = 1; k < kPyrLevels; ++k)
    if (k < a.L && b >= a.bin_off[k]) l = k;
  const int lb = b - a.bin_off[l];
  const int by = lb / a.bw[l], bx = lb - by * a.bw[l];
  const int p0 = a.plo[l][bx], p1 = a.plo[l][bx + 1];
  float s[E];
#pragma unroll
  for (int e = 0; e < E; ++e) s[e] = 0.f;
  for (int r = by * a.kh[l] + rs; r < (by + 1) * a.kh[l]; r += kPyrRS)
    for (int pc = p0; pc < p1; ++pc) {
      const float* q = part + (((size_t)n * H + r) * a.P + pc) * C + (size_t)ch * E;
#pragma unroll
      for (int e = 0; e < E; e += 4) {
        const float4 v = *reinterpret_cast<const float4*>(q + e);
        s[e] += v.x; s[e + 1] += v.y; s[e + 2] += v.z; s[e + 3] += v.w;
      }
    }
#pragma unroll
  for (int off = 1; off < kPyrRS; off <<= 1)
#pragma unroll
    for (int e = 0; e < E; ++e) s[e] += __shfl_xor(s[e], off);
  if (!live || rs != 0) return;
  const float inv = 1.f / (float)(a.kh[l] * a.kw[l]);
  Chunk<T> o;
#pragma unroll
  for (int e = 0; e < E; ++e) o.v[e] = s[e] * inv;
  o.store(reinterpret_cast<T*>(a.y[l]) + (((size_t)n * a.bh[l] + by) * a.bw[l] + bx) * C + (size_t)ch * E);
}

template <typename T>
__global__ void pyramid_binavg_kernel(const float* __restrict__ part, int N, int H, int C, PyrArgs a) {
  constexpr int E = Chunk<T>::N;
  const int cc = C / E;
  const size_t total = (size_t)N * a.total_bins * cc;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int ch = (int)(idx % cc);
  size_t p = idx / cc;
  const int b = (int)(p % a.total_bins);
  const int n = (int)(p / a.total_bins);
  int l = 0;
#pragma unroll
  for (int k = 1; k < kPyrLevels; ++k)
    if (k < a.L && b >= a.bin_off[k]) l = k;
  const int lb = b - a.bin_off[l];
  const int by = lb / a.bw[l], bx = lb - by * a.bw[l];
  float s[E];
#pragma unroll
  for (int e = 0; e < E; ++e) s[e] = 0.f;
  for (int r = by * a.kh[l]; r < (by + 1) * a.kh[l]; ++r) {
    const float* q = part + (((size_t)n * H + r) * a.total_cols + a.col_off[l] + bx) * C + (size_t)ch * E;
#pragma unroll
    for (int e = 0; e < E; e += 4) {
      const float4 v = *reinterpret_cast<const float4*>(q + e);
      s[e] += v.x; s[e + 1] += v.y; s[e + 2] += v.z; s[e + 3] += v.w;
    }
  }
  const float inv = 1.f / (float)(a.kh[l] * a.kw[l]);
  Chunk<T> o;
#pragma unroll
  for (int e = 0; e < E; ++e) o.v[e] = s[e] * inv;
  o.store(reinterpret_cast<T*>(a.y[l]) + (((size_t)n * a.bh[l] + by) * a.bw[l] + bx) * C + (size_t)ch * E);
}

// ---------------------------------------------------------------- preprocess
struct PreArgs { float mean[3], scale[3]; };

template <typename T>
__global__ void preprocess_kernel(const uint8_t* __restrict__ x, T* __restrict__ y, int N, int H,
                                  int W, int Ho, int Wo, int Cout, PreArgs a) {
  const size_t total = (size_t)N * Ho * Wo;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  size_t p = idx;
  const int ow = (int)(p % Wo); p /= Wo;
  const int oh = (int)(p % Ho);
  const int n = (int)(p / Ho);
  float v[3] = {0.f, 0.f, 0.f};
  if (oh < H && ow < W) {
    const uint8_t* s = x + (((size_t)n * H + oh) * W + ow) * 3;
    v[0] = (float)s[0]; v[1] = (float)s[1]; v[2] = (float)s[2];
  }
  T* out = y + idx * Cout;
  if (Cout == Chunk<T>::N) {  // one 16-byte store per pixel
    Chunk<T> o;
#pragma unroll
    for (int c = 0; c < Chunk<T>::N; ++c) o.v[c] = c < 3 ? (v[c] - a.mean[c]) * a.scale[c] : 0.f;
    o.store(out);
    return;
  }
  for (int c = 0; c < Cout; ++c) {
    const float o = c < 3 ? (v[c] - a.mean[c]) * a.scale[c] : 0.f;
    ElemTraits<T>::store(out + c, o);
  }
}

// ------------------------------------------------------- upsample + argmax
// Generic form: one thread per output pixel.
__global__ void upsample_argmax_kernel(const float* __restrict__ x, uint8_t* __restrict__ labels,
                                       float* __restrict__ logits_out, int N, int H, int W, int C,
                                       int x_ld, int Ho, int Wo, float sh, float sw) {
  const size_t total = (size_t)N * Ho * Wo;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  size_t p = idx;
  const int ow = (int)(p % Wo); p /= Wo;
  const int oh = (int)(p % Ho);
  const int n = (int)(p / Ho);
  int y0, y1, x0, x1;
  float ly, lx;
  interp_coord(oh, sh, H, y0, y1, ly);
  interp_coord(ow, sw, W, x0, x1, lx);
  const float* b = x + (size_t)n * H * W * x_ld;
  const float* ptl = b + ((size_t)y0 * W + x0) * x_ld;
  const float* ptr = b + ((size_t)y0 * W + x1) * x_ld;
  const float* pbl = b + ((size_t)y1 * W + x0) * x_ld;
  const float* pbr = b + ((size_t)y1 * W + x1) * x_ld;
  float best = -INFINITY;
  int arg = 0;
  for (int c = 0; c < C; ++c) {
    const float tl = ptl[c], tr = ptr[c], bl = pbl[c], br = pbr[c];
    const float top = tl + (tr - tl) * lx;
    const float bot = bl + (br - bl) * lx;
    const float v = top + (bot - top) * ly;
    if (logits_out) logits_out[idx * C + c] = v;
    if (v > best || c == 0) { best = v; arg = c; }
  }
  labels[idx] = (uint8_t)arg;
}

// Power-of-two align_corners zoom Z (out = (in-1)*Z + 1).  A block owns a tile of
// kZoomTX x kZoomTY INPUT cells; the thread of a cell produces the Z x Z output pixels whose
// top-left neighbour the cell is.  Source coordinates are exact here (scale = 1/Z), so the
// lerp weights are (i/Z, j/Z).
//
// Most cells never interpolate: if the four corner pixels of a cell have the same argmax k,
// every pixel inside has it too (for c < k all four corners have l_c < l_k, for c > k
// l_c <= l_k, and a bilinear blend is a convex combination, so k stays the first maximum).
// Pass 1 computes the argmax of every input pixel of the tile (+ one halo row and column)
// into LDS; cells with four equal corners store their label block at once, the others are
// queued in LDS and pass 2 hands them out densely, so that waves do not idle beside a few
// slow lanes.
constexpr int kZoomTX = 32, kZoomTY = 8;

template <int Z> struct ZoomOut {
  // stores the Z labels of one output row of a cell (4 per unaligned dword store)
  static __device__ __forceinline__ void row(uint8_t* dst, const int (&l)[Z], int valid) {
    if (valid >= Z) {
#pragma unroll
      for (int j = 0; j < Z; j += 4) {
        if constexpr (Z >= 4) {
          *reinterpret_cast<unsigned*>(dst + j) =
              (unsigned)l[j] | ((unsigned)l[j + 1] << 8) | ((unsigned)l[j + 2] << 16) | ((unsigned)l[j + 3] << 24);
        } else {
          *reinterpret_cast<unsigned short*>(dst + j) = (unsigned short)((unsigned)l[j] | ((unsigned)l[j + 1] << 8));
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < Z; ++j)
        if (j < valid) dst[j] = (uint8_t)l[j];
    }
  }
};

__device__ __forceinline__ int pixel_argmax(const float* __restrict__ px, int C) {
  const float4* p4 = reinterpret_cast<const float4*>(px);
  float best = -INFINITY;
  int arg = 0;
  const int groups = (C + 3) >> 2;
  for (int g = 0; g < groups; ++g) {
    const float4 v4 = p4[g];
    const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = g * 4 + e;
      if (c < C && (v[e] > best || c == 0)) { best = v[e]; arg = c; }
    }
  }
  return arg;
}

template <int Z>
__global__ void __launch_bounds__(kZoomTX * kZoomTY)
upsample_argmax_zoom_kernel(const float* __restrict__ x, uint8_t* __restrict__ labels, int N, int H,
                            int W, int C, int x_ld, int Ho, int Wo, int tiles_x, int tiles_y) {
  constexpr int NTH = kZoomTX * kZoomTY;
  __shared__ uint8_t amax[kZoomTY + 1][kZoomTX + 4];
  __shared__ unsigned short queue[NTH];
  __shared__ int qcount;
  const int tid = threadIdx.x;
  int tile = blockIdx.x;
  const int tx = tile % tiles_x; tile /= tiles_x;
  const int ty = tile % tiles_y;
  const int n = tile / tiles_y;
  const float* b = x + (size_t)n * H * W * x_ld;
  const int cx = tx * kZoomTX, cy = ty * kZoomTY;

  // ---- pass 1: argmax of the tile's pixels and of the halo column / row (clamped to the image:
  // the same clamping the interpolation applies to x0 + 1, y0 + 1)
  {
    const int lx = tid % kZoomTX, ly = tid / kZoomTX;
    amax[ly][lx] = (uint8_t)pixel_argmax(b + ((size_t)min(cy + ly, H - 1) * W + min(cx + lx, W - 1)) * x_ld, C);
    if (tid < kZoomTY + 1 + kZoomTX) {
      const int hy = tid <= kZoomTY ? tid : kZoomTY, hx = tid <= kZoomTY ? kZoomTX : tid - (kZoomTY + 1);
      amax[hy][hx] = (uint8_t)pixel_argmax(b + ((size_t)min(cy + hy, H - 1) * W + min(cx + hx, W - 1)) * x_ld, C);
    }
    if (tid == 0) qcount = 0;
  }
  __syncthreads();

  // Z x Z labels of cell (y0, x0) -> output
  auto store_cell = [&](int y0, int x0, const int (&arg)[Z][Z]) {
    const int cols = min(Z, Wo - x0 * Z);
#pragma unroll
    for (int i = 0; i < Z; ++i) {
      const int oh = y0 * Z + i;
      if (oh < Ho) ZoomOut<Z>::row(labels + ((size_t)n * Ho + oh) * Wo + (size_t)x0 * Z, arg[i], cols);
    }
  };
  {
    const int lx = tid % kZoomTX, ly = tid / kZoomTX;
    const int x0 = cx + lx, y0 = cy + ly;
    if (x0 < W && y0 < H) {
      const int a00 = amax[ly][lx], a01 = amax[ly][lx + 1], a10 = amax[ly + 1][lx], a11 = amax[ly + 1][lx + 1];
      if (a00 == a01 && a00 == a10 && a00 == a11) {
        int arg[Z][Z];
#pragma unroll
        for (int i = 0; i < Z; ++i)
#pragma unroll
          for (int j = 0; j < Z; ++j) arg[i][j] = a00;
        store_cell(y0, x0, arg);
      } else {
        queue[atomicAdd(&qcount, 1)] = (unsigned short)tid;
      }
    }
  }
  __syncthreads();

  // ---- pass 2: the cells that need the interpolation
  const int nq = qcount;
  for (int q = tid; q < nq; q += NTH) {
    const int cell = queue[q];
    const int x0 = cx + cell % kZoomTX, y0 = cy + cell / kZoomTX;
    const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
    const float4* ptl = reinterpret_cast<const float4*>(b + ((size_t)y0 * W + x0) * x_ld);
    const float4* ptr = reinterpret_cast<const float4*>(b + ((size_t)y0 * W + x1) * x_ld);
    const float4* pbl = reinterpret_cast<const float4*>(b + ((size_t)y1 * W + x0) * x_ld);
    const float4* pbr = reinterpret_cast<const float4*>(b + ((size_t)y1 * W + x1) * x_ld);
    float best[Z][Z];
    int arg[Z][Z];
#pragma unroll
    for (int i = 0; i < Z; ++i)
#pragma unroll
      for (int j = 0; j < Z; ++j) { best[i][j] = -INFINITY; arg[i][j] = 0; }
    const int groups = (C + 3) >> 2;
    for (int g = 0; g < groups; ++g) {
      const float4 tl4 = ptl[g], tr4 = ptr[g], bl4 = pbl[g], br4 = pbr[g];
      const float tl[4] = {tl4.x, tl4.y, tl4.z, tl4.w}, tr[4] = {tr4.x, tr4.y, tr4.z, tr4.w};
      const float bl[4] = {bl4.x, bl4.y, bl4.z, bl4.w}, br[4] = {br4.x, br4.y, br4.z, br4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = g * 4 + e;
        if (c < C) {
#pragma unroll
          for (int j = 0; j < Z; ++j) {
            const float lx = (float)j * (1.0f / Z);
            const float top = tl[e] + (tr[e] - tl[e]) * lx;
            const float bot = bl[e] + (br[e] - bl[e]) * lx;
#pragma unroll
            for (int i = 0; i < Z; ++i) {
              const float v = top + (bot - top) * ((float)i * (1.0f / Z));
              if (v > best[i][j] || c == 0) { best[i][j] = v; arg[i][j] = c; }
            }
          }
        }
      }
    }
    store_cell(y0, x0, arg);
  }
}

// ----------------------------------------------------------------- label LUT
__global__ void label_lut_kernel(const uint8_t* __restrict__ labels, float* __restrict__ y,
                                 const float* __restrict__ table, int N, int H, int W, int Ho,
                                 int Wo, int CH, int num_classes) {
  const size_t total = (size_t)N * Ho * Wo;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  size_t p = idx;
  const int ow = (int)(p % Wo); p /= Wo;
  const int oh = (int)(p % Ho);
  const int n = (int)(p / Ho);
  int l = labels[((size_t)n * H + oh) * W + ow];
  l = min(l, num_classes - 1);
  for (int c = 0; c < CH; ++c) y[idx * CH + c] = table[l * CH + c];
}

// ---------------------------------------------------------- confusion matrix
// Per-wave private histograms in LDS (int32), flushed once per block with
// 64-bit global atomics.  A thread folds runs of equal (label, pred) pairs
// before touching LDS, which removes most same-address LDS atomics on label
// maps made of large uniform regions.
__global__ void confusion_kernel(const uint8_t* __restrict__ pred, const uint8_t* __restrict__ lab,
                                 size_t count, int nc, int ignore,
                                 unsigned long long* __restrict__ cm) {
  extern __shared__ unsigned int hist[];  // [waves][nc*nc]
  const int cells = nc * nc;
  const int waves = blockDim.x >> 6;
  for (int i = threadIdx.x; i < cells * waves; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  unsigned int* mine = hist + (threadIdx.x >> 6) * cells;
  constexpr int PER = 16;  // consecutive pixels per thread per step
  const bool vec16 = ((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(lab)) & 15) == 0;
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  for (size_t base = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * PER; base < count;
       base += nthreads * PER) {
    int run_key = -1;
    unsigned run_len = 0;
    auto take = [&](int l, int pr) {
      int key = -1;
      if (l != ignore && l < nc && pr < nc) key = l * nc + pr;
      if (key == run_key) {
        ++run_len;
      } else {
        if (run_key >= 0) atomicAdd(&mine[run_key], run_len);
        run_key = key;
        run_len = 1;
      }
    };
    if (vec16 && base + PER <= count) {
      // the thread's 16 labels and 16 predictions in one 16-byte load each
      const uint4 l4 = *reinterpret_cast<const uint4*>(lab + base);
      const uint4 p4 = *reinterpret_cast<const uint4*>(pred + base);
      const unsigned lw[4] = {l4.x, l4.y, l4.z, l4.w}, pw[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) take((int)((lw[q] >> (8 * e)) & 255u), (int)((pw[q] >> (8 * e)) & 255u));
    } else {
      const size_t end = base + PER < count ? base + PER : count;
      for (size_t i = base; i < end; ++i) take(lab[i], pred[i]);
    }
    if (run_key >= 0) atomicAdd(&mine[run_key], run_len);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < cells; i += blockDim.x) {
    unsigned long long s = 0;
    for (int w = 0; w < waves; ++w) s += hist[w * cells + i];
    if (s) atomicAdd(&cm[i], s);
  }
}

// ------------------------------------------------------------ depthwise conv
template <typename T>
__global__ void dwconv_kernel(const T* __restrict__ x, const float* __restrict__ w,
                              const float* __restrict__ bias, T* __restrict__ y, int N, int H, int W,
                              int C, int kh, int kw, int stride, int dil, int pad_t, int pad_l,
                              int Ho, int Wo, int act) {
  constexpr int E = Chunk<T>::N;
  const int cc = C / E;
  const size_t total = (size_t)N * Ho * Wo * cc;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int ch = (int)(idx % cc);
  size_t p = idx / cc;
  const int ow = (int)(p % Wo); p /= Wo;
  const int oh = (int)(p % Ho);
  const int n = (int)(p / Ho);
  float acc[E];
#pragma unroll
  for (int e = 0; e < E; ++e) acc[e] = 0.f;
  for (int r = 0; r < kh; ++r) {
    const int ih = oh * stride - pad_t + r * dil;
    if ((unsigned)ih >= (unsigned)H) continue;
    for (int s = 0; s < kw; ++s) {
      const int iw = ow * stride - pad_l + s * dil;
      if ((unsigned)iw >= (unsigned)W) continue;
      Chunk<T> v;
      v.load(x + (((size_t)n * H + ih) * W + iw) * C + ch * E);
      const float* wp = w + (size_t)(r * kw + s) * C + ch * E;
#pragma unroll
      for (int e = 0; e < E; ++e) acc[e] = fmaf(v.v[e], wp[e], acc[e]);
    }
  }
  Chunk<T> o;
#pragma unroll
  for (int e = 0; e < E; ++e)
    o.v[e] = apply_act(acc[e] + (bias ? bias[ch * E + e] : 0.f), act);
  o.store(y + idx * E);
}

// 3x3 depthwise convolution, one thread per 16-byte channel chunk and run of R output
// pixels of one row: the 9 x E filter taps of the chunk stay in registers, and every input
// chunk the run touches is loaded and unpacked once and used by all the outputs it feeds
// (stride 1: R + 2*DIL columns per filter row instead of 3*R).
template <typename T, int STRIDE, int DIL, int R>
__global__ void dwconv3x3_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                 const float* __restrict__ bias, T* __restrict__ y, int N, int H, int W,
                                 int C, int pad_t, int pad_l, int Ho, int Wo, int runs, int act) {
  constexpr int E = Chunk<T>::N;
  constexpr int COLS = (R - 1) * STRIDE + 2 * DIL + 1;   // input columns under one filter row
  const int cc = C / E;
  const size_t total = (size_t)N * Ho * runs * cc;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int ch = (int)(idx % cc);
  size_t p = idx / cc;
  const int run = (int)(p % runs); p /= runs;
  const int oh = (int)(p % Ho);
  const int n = (int)(p / Ho);
  const int ow0 = run * R;
  float wt[9][E];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < E; e += 4) {
      const float4 w4 = *reinterpret_cast<const float4*>(w + (size_t)t * C + ch * E + e);
      wt[t][e] = w4.x; wt[t][e + 1] = w4.y; wt[t][e + 2] = w4.z; wt[t][e + 3] = w4.w;
    }
  float acc[R][E];
#pragma unroll
  for (int j = 0; j < R; ++j)
#pragma unroll
    for (int e = 0; e < E; ++e) acc[j][e] = 0.f;
  const int iw0 = ow0 * STRIDE - pad_l;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int ih = oh * STRIDE - pad_t + r * DIL;
    if ((unsigned)ih >= (unsigned)H) continue;
    const T* row = x + (((size_t)n * H + ih) * W) * C + ch * E;
#pragma unroll
    for (int t = 0; t < COLS; ++t) {
      // does any output of the run read column t of this filter row?
      bool used = false;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int num = t - s * DIL;
        if (num >= 0 && num % STRIDE == 0 && num / STRIDE < R) used = true;
      }
      if (!used) continue;
      const int iw = iw0 + t;
      if ((unsigned)iw >= (unsigned)W) continue;
      Chunk<T> v;
      v.load(row + (size_t)iw * C);
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int num = t - s * DIL;
        if (num >= 0 && num % STRIDE == 0 && num / STRIDE < R) {
          const int j = num / STRIDE;
#pragma unroll
          for (int e = 0; e < E; ++e) acc[j][e] = fmaf(v.v[e], wt[r * 3 + s][e], acc[j][e]);
        }
      }
    }
  }
  float b[E];
#pragma unroll
  for (int e = 0; e < E; ++e) b[e] = bias ? bias[ch * E + e] : 0.f;
  T* out = y + ((((size_t)n * Ho + oh) * Wo + ow0) * C) + ch * E;
#pragma unroll
  for (int j = 0; j < R; ++j) {
    if (ow0 + j >= Wo) break;
    Chunk<T> o;
#pragma unroll
    for (int e = 0; e < E; ++e) o.v[e] = apply_act(acc[j][e] + b[e], act);
    o.store_nt(out + (size_t)j * C);
  }
}

}  // namespace fss

using namespace fss;

// tensor types of the bandwidth kernels: float, bfloat16, split-bfloat16 (FSS_F32_BF16X3: 4-byte
// slots, 8 channels = 32 bytes of 8 high parts + 8 remainders)
#define FSS_DTYPE_OK(dt) FSS_REQUIRE((dt) == FSS_F32 || (dt) == FSS_BF16 || (dt) == FSS_F32_BF16X3, "bad dtype")
static inline int chunk_of(int dt) { return dt == FSS_F32 ? 4 : 8; }   // channels per chunk
#define FSS_CHUNK_OK(c, dt)                                                      \
  FSS_REQUIRE((c) > 0 && (c) % chunk_of(dt) == 0,                   \
              "channel count must be a multiple of 16 bytes of elements")
#define FSS_ALIGNED(p) FSS_REQUIRE(((uintptr_t)(p) & (dtype == FSS_F32_BF16X3 ? 31 : 15)) == 0, "unaligned pointer")

extern "C" {

int fss_maxpool_nhwc(const void* x, void* y, int n, int h, int wd, int c, int k, int stride,
                     int pad_top, int pad_left, int ho, int wo, int dtype, void* stream) {
  FSS_REQUIRE(x && y, "null argument");
  // (half precision too: the largest of a set of half-precision numbers is one of them)
  FSS_REQUIRE(dtype == FSS_F32 || dtype == FSS_BF16 || dtype == FSS_F32_BF16X3 || dtype == FSS_F16, "bad dtype");
  FSS_CHUNK_OK(c, dtype);
  FSS_ALIGNED(x); FSS_ALIGNED(y);
  FSS_REQUIRE(n > 0 && h > 0 && wd > 0 && ho > 0 && wo > 0 && k > 0 && stride > 0, "bad shape");
  // every window must contain at least one real pixel
  FSS_REQUIRE(pad_top >= 0 && pad_left >= 0 && pad_top < k && pad_left < k, "bad padding");
  FSS_REQUIRE((long long)(ho - 1) * stride - pad_top < h && (long long)(wo - 1) * stride - pad_left < wd,
              "output size exceeds the input");
  const size_t total = (size_t)n * ho * wo * (c / chunk_of(dtype));
  hipStream_t s = (hipStream_t)stream;
  if (dtype == FSS_F32)
    hipLaunchKernelGGL(maxpool_kernel<float>, dim3(grid_for(total)), dim3(kBlock), 0, s,
                       (const float*)x, (float*)y, n, h, wd, c, k, stride, pad_top, pad_left, ho, wo);
  else if (dtype == FSS_F32_BF16X3)
    hipLaunchKernelGGL(maxpool_kernel<f32s_t>, dim3(grid_for(total)), dim3(kBlock), 0, s,
                       (const f32s_t*)x, (f32s_t*)y, n, h, wd, c, k, stride, pad_top, pad_left, ho, wo);
  else if (dtype == FSS_F16)
    hipLaunchKernelGGL(maxpool_kernel<f16_t>, dim3(grid_for(total)), dim3(kBlock), 0, s,
                       (const f16_t*)x, (f16_t*)y, n, h, wd, c, k, stride, pad_top, pad_left, ho, wo);
  else
    hipLaunchKernelGGL(maxpool_kernel<bf16_t>, dim3(grid_for(total)), dim3(kBlock), 0, s,
                       (const bf16_t*)x, (bf16_t*)y, n, h, wd, c, k, stride, pad_top, pad_left, ho, wo);
  FSS_LAUNCH_CHECK();
  return 0;
}

int fss_subsample_nhwc(const void* x, void* y, int n, int h, int wd, int c, int stride, int ho,
                       int wo, int dtype, void* stream) {
  FSS_REQUIRE(x && y, "null argument");
  FSS_DTYPE_OK(dtype);
  FSS_CHUNK_OK(c, dtype);
  FSS_ALIGNED(x); FSS_ALIGNED(y);
  FSS_REQUIRE(n > 0 && h > 0 && wd > 0 && ho > 0 && wo > 0 && stride > 0, "bad shape");
  FSS_REQUIRE((long long)(ho - 1) * stride < h && (long long)(wo - 1) * stride < wd,
              "output size exceeds the input");
  const size_t total = (size_t)n * ho * wo * (c / chunk_of(dtype));
  hipStream_t s = (hipStream_t)stream;
  if (dtype == FSS_F32)
    hipLaunchKernelGGL(subsample_kernel<float>, dim3(grid_for(total)), dim3(kBlock), 0, s,
                       (const float*)x, (float*)y, n, h, wd, c, stride, ho, wo);
  else if (dtype == FSS_F32_BF16X3)
    hipLaunchKernelGGL(subsample_kernel<f32s_t>, dim3(grid_for(total)), dim3(kBlock), 0, s,
                       (const f32s_t*)x, (f32s_t*)y, n, h, wd, c, stride, ho, wo);
  else
    hipLaunchKernelGGL(subsample_kernel<bf16_t>, dim3(grid_for(total)), dim3(kBlock), 0, s,
                       (const bf16_t*)x, (bf16_t*)y, n, h, wd, c, stride, ho, wo);
  FSS_LAUNCH_CHECK();
  return 0;
}

int fss_avgpool_bins_nhwc(const void* x, void* y, int n, int h, int wd, int c, int kh, int kw,
                          int bh, int bw, int dtype, void* stream) {
  FSS_REQUIRE(x && y, "null argument");
  FSS_DTYPE_OK(dtype);
  FSS_CHUNK_OK(c, dtype);
  FSS_ALIGNED(x); FSS_ALIGNED(y);
  FSS_REQUIRE(n > 0 && h > 0 && wd > 0 && kh > 0 && kw > 0 && bh > 0 && bw > 0, "bad shape");
  FSS_REQUIRE((long long)bh * kh <= h && (long long)bw * kw <= wd, "bins exceed the input");
  constexpr int TX = 16, TY = 16;
  const int cc = c / chunk_of(dtype);
  dim3 grid((unsigned)(n * bh * bw), (unsigned)ceil_div(cc, TX));
  hipStream_t s = (hipStream_t)stream;
  if (dtype == FSS_F32)
    hipLaunchKernelGGL((avgpool_bins_kernel<float, TX, TY>), grid, dim3(TX, TY), 0, s,
                       (const float*)x, (float*)y, h, wd, c, kh, kw, bh, bw);
  else if (dtype == FSS_F32_BF16X3)
    hipLaunchKernelGGL((avgpool_bins_kernel<f32s_t, TX, TY>), grid, dim3(TX, TY), 0, s,
                       (const f32s_t*)x, (f32s_t*)y, h, wd, c, kh, kw, bh, bw);
  else
    hipLaunchKernelGGL((avgpool_bins_kernel<bf16_t, TX, TY>), grid, dim3(TX, TY), 0, s,
                       (const bf16_t*)x, (bf16_t*)y, h, wd, c, kh, kw, bh, bw);
  FSS_LAUNCH_CHECK();
  return 0;
}

// Both launches of the one-pass pyramid pooling; `workspace` (device memory, checked against
// workspace_bytes) carries the row sums from the first launch to the second.
static int pyramid_pool_launch(const void* x, void* const* y, const int* kh, const int* kw,
                               const int* bh, const int* bw, int levels, float* workspace,
                               size_t workspace_bytes, const char* too_small, int n, int h, int wd,
                               int c, int dtype, void* stream) {
  FSS_REQUIRE(x && y && kh && kw && bh && bw && workspace, "null argument");
  FSS_DTYPE_OK(dtype);
  FSS_CHUNK_OK(c, dtype);
  FSS_ALIGNED(x); FSS_ALIGNED(workspace);
  FSS_REQUIRE(levels >= 1 && levels <= kPyrLevels, "between 1 and 4 levels");
  FSS_REQUIRE(n > 0 && h > 0 && wd > 0, "bad shape");
  PyrArgs a;
  a.L = levels;
  a.P = 0;
  for (int i = 0; i <= kPyrPieces; ++i) a.edge[i] = 0;
  for (int l = 0; l < kPyrLevels; ++l)
    for (int j = 0; j <= kPyrBinCols; ++j) a.plo[l][j] = 0;
  a.total_cols = a.total_bins = 0;
  for (int l = 0; l < kPyrLevels; ++l) {
    a.kh[l] = a.kw[l] = a.bh[l] = a.bw[l] = 1;
    a.col_off[l] = a.bin_off[l] = 0;
    a.y[l] = nullptr;
  }
  for (int l = 0; l < levels; ++l) {
    FSS_REQUIRE(y[l] != nullptr, "null output");
    FSS_ALIGNED(y[l]);
    FSS_REQUIRE(kh[l] > 0 && kw[l] > 0 && bh[l] > 0 && bw[l] > 0, "bad level");
    FSS_REQUIRE((long long)bh[l] * kh[l] <= h && (long long)bw[l] * kw[l] <= wd, "bins exceed the input");
    a.kh[l] = kh[l]; a.kw[l] = kw[l]; a.bh[l] = bh[l]; a.bw[l] = bw[l];
    a.col_off[l] = a.total_cols;
    a.bin_off[l] = a.total_bins;
    a.total_cols += bw[l];
    a.total_bins += bh[l] * bw[l];
    a.y[l] = y[l];
  }
  FSS_REQUIRE(workspace_bytes >= (size_t)n * h * a.total_cols * c * sizeof(float), too_small);
  const int cc = c / chunk_of(dtype);
  hipStream_t s = (hipStream_t)stream;
  const size_t t1 = (size_t)n * h * cc, t2 = (size_t)n * a.total_bins * cc;
  // small problems: cut the rows into pieces at every bin edge for more threads
  bool pieces_ok = t1 < 98304;
  for (int l = 0; l < levels; ++l) pieces_ok = pieces_ok && bw[l] <= kPyrBinCols;
  if (pieces_ok) {
    int edges[kPyrLevels * kPyrBinCols + 1], ne = 0;
    edges[ne++] = 0;
    for (int l = 0; l < levels; ++l)
      for (int j = 1; j <= bw[l]; ++j) edges[ne++] = j * kw[l];
    std::sort(edges, edges + ne);
    ne = (int)(std::unique(edges, edges + ne) - edges);
    pieces_ok = ne - 1 <= kPyrPieces && ne - 1 <= a.total_cols;
    if (pieces_ok) {
      a.P = ne - 1;
      for (int i = 0; i < ne; ++i) a.edge[i] = edges[i];
      for (int l = 0; l < levels; ++l)
        for (int j = 0; j <= bw[l]; ++j)
          a.plo[l][j] = (unsigned char)(std::lower_bound(edges, edges + ne, j * kw[l]) - edges);
    }
  }
  if (pieces_ok) {
    const size_t tp = t1 * a.P;
    if (dtype == FSS_F32) {
      hipLaunchKernelGGL(pyramid_piecesum_kernel<float>, dim3(grid_for(tp)), dim3(kBlock), 0, s,
                         (const float*)x, workspace, n, h, wd, c, a);
      hipLaunchKernelGGL(pyramid_piecebin_kernel<float>, dim3(grid_for(t2 * kPyrRS)), dim3(kBlock), 0, s,
                         workspace, n, h, c, a);
    } else if (dtype == FSS_F32_BF16X3) {
      hipLaunchKernelGGL(pyramid_piecesum_kernel<f32s_t>, dim3(grid_for(tp)), dim3(kBlock), 0, s,
                         (const f32s_t*)x, workspace, n, h, wd, c, a);
      hipLaunchKernelGGL(pyramid_piecebin_kernel<f32s_t>, dim3(grid_for(t2 * kPyrRS)), dim3(kBlock), 0, s,
                         workspace, n, h, c, a);
    } else {
      hipLaunchKernelGGL(pyramid_piecesum_kernel<bf16_t>, dim3(grid_for(tp)), dim3(kBlock), 0, s,
                         (const bf16_t*)x, workspace, n, h, wd, c, a);
      hipLaunchKernelGGL(pyramid_piecebin_kernel<bf16_t>, dim3(grid_for(t2 * kPyrRS)), dim3(kBlock), 0, s,
                         workspace, n, h, c, a);
    }
    FSS_LAUNCH_CHECK();
    return 0;
  }
  if (dtype == FSS_F32) {
    hipLaunchKernelGGL(pyramid_rowsum_kernel<float>, dim3(grid_for(t1)), dim3(kBlock), 0, s,
                       (const float*)x, workspace, n, h, wd, c, a);
    hipLaunchKernelGGL(pyramid_binavg_kernel<float>, dim3(grid_for(t2)), dim3(kBlock), 0, s,
                       workspace, n, h, c, a);
  } else if (dtype == FSS_F32_BF16X3) {
    hipLaunchKernelGGL(pyramid_rowsum_kernel<f32s_t>, dim3(grid_for(t1)), dim3(kBlock), 0, s,
                       (const f32s_t*)x, workspace, n, h, wd, c, a);
    hipLaunchKernelGGL(pyramid_binavg_kernel<f32s_t>, dim3(grid_for(t2)), dim3(kBlock), 0, s,
                       workspace, n, h, c, a);
  } else {
    hipLaunchKernelGGL(pyramid_rowsum_kernel<bf16_t>, dim3(grid_for(t1)), dim3(kBlock), 0, s,
                       (const bf16_t*)x, workspace, n, h, wd, c, a);
    hipLaunchKernelGGL(pyramid_binavg_kernel<bf16_t>, dim3(grid_for(t2)), dim3(kBlock), 0, s,
                       workspace, n, h, c, a);
  }
  FSS_LAUNCH_CHECK();
  return 0;
}

int fss_avgpool_pyramid_nhwc(const void* x, void* const* y, const int* kh, const int* kw,
                             const int* bh, const int* bw, int levels, float* workspace,
                             size_t workspace_bytes, int n, int h, int wd, int c, int dtype,
                             void* stream) {
  return pyramid_pool_launch(x, y, kh, kw, bh, bw, levels, workspace, workspace_bytes,
                             "workspace smaller than n*h*sum(bw)*c floats", n, h, wd, c, dtype, stream);
}

// Row-sum scratch of fss_avgpool_pyramid_nhwc_static: one array per GPU the library runs on,
// held until the process ends.  Host code never uses the array's name as an address (it is a
// host-side stand-in); the device address is resolved once per device and kept below.
constexpr size_t kPyrScratchBytes = (size_t)FSS_PYRAMID_STATIC_MAX_BYTES;
static __device__ __align__(256) float scratch[kPyrScratchBytes / sizeof(float)];
constexpr int kMaxDevices = 64;
static float* g_scratch_ptr[kMaxDevices];
static std::mutex g_scratch_mutex;

// Device address of `scratch` on the current device (resolved on the first call per device).
static int pyramid_scratch(float** out) {
  int dev = -1;
  FSS_CHECK_HIP(hipGetDevice(&dev));
  FSS_REQUIRE(dev >= 0 && dev < kMaxDevices, "device index beyond the scratch table");
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (g_scratch_ptr[dev] == nullptr) {
    void* p = nullptr;
    FSS_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    FSS_REQUIRE(p != nullptr, "no device address for the scratch array");
    g_scratch_ptr[dev] = static_cast<float*>(p);
  }
  *out = g_scratch_ptr[dev];
  return 0;
}

// Set-up hook of fss_set_device: resolves the address for the device just made current, so
// that the entry point's own first call on it has nothing left to look up.
extern "C++" int fss::pyramid_scratch_init() {
  float* p = nullptr;
  return pyramid_scratch(&p);
}

int fss_avgpool_pyramid_nhwc_static(const void* x, void* const* y, const int* kh, const int* kw,
                                    const int* bh, const int* bw, int levels, int n, int h, int wd,
                                    int c, int dtype, void* stream) {
  float* ws = nullptr;
  const int rc = pyramid_scratch(&ws);
  if (rc) return rc;
  // every row sum the second launch reads is written by the first, so nothing is cleared
  return pyramid_pool_launch(x, y, kh, kw, bh, bw, levels, ws, kPyrScratchBytes,
                             "n*h*sum(bw)*c floats exceed FSS_PYRAMID_STATIC_MAX_BYTES", n, h, wd, c,
                             dtype, stream);
}

int fss_resize_bilinear_nhwc(const void* x, void* y, int n, int h, int wd, int c, int ho, int wo,
                             int y_ld, int align_corners, int dtype, void* stream) {
  FSS_REQUIRE(x && y, "null argument");
  FSS_DTYPE_OK(dtype);
  FSS_REQUIRE(n > 0 && h > 0 && wd > 0 && c > 0 && ho > 0 && wo > 0, "bad shape");
  FSS_REQUIRE(y_ld >= c, "y_ld < c");
  const float sh = resize_scale(h, ho, align_corners), sw = resize_scale(wd, wo, align_corners);
  const int e = chunk_of(dtype);
  const unsigned amask = dtype == FSS_F32_BF16X3 ? 31u : 15u;
  const bool vec = (c % e == 0) && (y_ld % e == 0) && (((uintptr_t)x & amask) == 0) &&
                   (((uintptr_t)y & amask) == 0);
  FSS_REQUIRE(vec || dtype != FSS_F32_BF16X3,
              "split-bf16 tensors: 32-byte aligned, channels and pixel stride in multiples of 8");
  hipStream_t s = (hipStream_t)stream;
  const size_t total = (size_t)n * ho * wo * (vec ? c / e : c);
#define FSS_RESIZE(T, V)                                                                        \
  hipLaunchKernelGGL((resize_bilinear_kernel<T, V>), dim3(grid_for(total)), dim3(kBlock), 0, s, \
                     (const T*)x, (T*)y, n, h, wd, c, ho, wo, y_ld, sh, sw)
  if (dtype == FSS_F32) { if (vec) FSS_RESIZE(float, true); else FSS_RESIZE(float, false); }
  else if (dtype == FSS_F32_BF16X3) { if (vec) FSS_RESIZE(f32s_t, true); else FSS_RESIZE(f32s_t, false); }
  else { if (vec) FSS_RESIZE(bf16_t, true); else FSS_RESIZE(bf16_t, false); }
#undef FSS_RESIZE
  FSS_LAUNCH_CHECK();
  return 0;
}

int fss_pyramid_add_nhwc(const void* x, const void* const* p, const int* bh, const int* bw,
                         int levels, void* y, int n, int h, int wd, int c, int dtype,
                         void* stream) {
  FSS_REQUIRE(y && p && bh && bw, "null argument");
  FSS_DTYPE_OK(dtype);
  FSS_CHUNK_OK(c, dtype);
  FSS_ALIGNED(x); FSS_ALIGNED(y);
  FSS_REQUIRE(levels >= 0 && levels <= 8, "at most 8 pyramid levels");
  FSS_REQUIRE(n > 0 && h > 0 && wd > 0, "bad shape");
  PyramidArgs a;
  a.levels = levels;
  for (int l = 0; l < 8; ++l) { a.p[l] = nullptr; a.bh[l] = a.bw[l] = 1; a.sh[l] = a.sw[l] = 0.f; }
  for (int l = 0; l < levels; ++l) {
    FSS_REQUIRE(p[l] != nullptr && bh[l] > 0 && bw[l] > 0, "bad pyramid level");
    FSS_ALIGNED(p[l]);
    a.p[l] = p[l]; a.bh[l] = bh[l]; a.bw[l] = bw[l];
    a.sh[l] = resize_scale(bh[l], h, 1);
    a.sw[l] = resize_scale(bw[l], wd, 1);
  }
  const int cc = c / chunk_of(dtype);
  const long long blocks = (long long)n * h * ((wd + kPyrPix - 1) / kPyrPix);
  FSS_REQUIRE(blocks < 0x7fffffffLL, "too many pixels");
  const int threads = cc >= 256 ? 256 : (cc >= 128 ? 128 : 64);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == FSS_F32)
    hipLaunchKernelGGL(pyramid_add_kernel<float>, dim3((unsigned)blocks), dim3(threads), 0, s,
                       (const float*)x, (float*)y, a, n, h, wd, c);
  else if (dtype == FSS_F32_BF16X3)
    hipLaunchKernelGGL(pyramid_add_kernel<f32s_t>, dim3((unsigned)blocks), dim3(threads), 0, s,
                       (const f32s_t*)x, (f32s_t*)y, a, n, h, wd, c);
  else
    hipLaunchKernelGGL(pyramid_add_kernel<bf16_t>, dim3((unsigned)blocks), dim3(threads), 0, s,
                       (const bf16_t*)x, (bf16_t*)y, a, n, h, wd, c);
  FSS_LAUNCH_CHECK();
  return 0;
}

int fss_add_act(const void* a, const void* b, void* y, size_t count, int act, int dtype,
                void* stream) {
  FSS_REQUIRE(a && b && y, "null argument");
  FSS_DTYPE_OK(dtype);
  const int e = chunk_of(dtype);
  FSS_REQUIRE(count > 0 && count % e == 0, "count must be a multiple of 16 bytes of elements");
  FSS_ALIGNED(a); FSS_ALIGNED(b); FSS_ALIGNED(y);
  const size_t chunks = count / e;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == FSS_F32)
    hipLaunchKernelGGL(add_act_kernel<float>, dim3(grid_for(chunks)), dim3(kBlock), 0, s,
                       (const float*)a, (const float*)b, (float*)y, chunks, act);
  else if (dtype == FSS_F32_BF16X3)
    hipLaunchKernelGGL(add_act_kernel<f32s_t>, dim3(grid_for(chunks)), dim3(kBlock), 0, s,
                       (const f32s_t*)a, (const f32s_t*)b, (f32s_t*)y, chunks, act);
  else
    hipLaunchKernelGGL(add_act_kernel<bf16_t>, dim3(grid_for(chunks)), dim3(kBlock), 0, s,
                       (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)y, chunks, act);
  FSS_LAUNCH_CHECK();
  return 0;
}

int fss_to_half(const void* x, void* y, size_t count, int dtype, void* stream) {
  FSS_REQUIRE(x && y, "null argument");
  FSS_REQUIRE(dtype == FSS_F32 || dtype == FSS_F32_BF16X3, "dtype must be FSS_F32 or FSS_F32_BF16X3");
  const int e = chunk_of(dtype);
  FSS_REQUIRE(count > 0 && count % e == 0, "count must be a multiple of the channel group");
  FSS_ALIGNED(x);
  FSS_REQUIRE(((uintptr_t)y & (e == 8 ? 15 : 7)) == 0, "unaligned pointer");
  const size_t chunks = count / e;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == FSS_F32)
    hipLaunchKernelGGL(to_half_kernel<float>, dim3(grid_for(chunks)), dim3(kBlock), 0, s,
                       (const float*)x, (f16_t*)y, chunks);
  else
    hipLaunchKernelGGL(to_half_kernel<f32s_t>, dim3(grid_for(chunks)), dim3(kBlock), 0, s,
                       (const f32s_t*)x, (f16_t*)y, chunks);
  FSS_LAUNCH_CHECK();
  return 0;
}

int fss_copy_channels(const void* x, int x_ld, void* y, int y_ld, size_t pixels, int c, int dtype,
                      void* stream) {
  FSS_REQUIRE(x && y, "null argument");
  FSS_DTYPE_OK(dtype);
  FSS_CHUNK_OK(c, dtype);
  const int e = chunk_of(dtype);
  FSS_REQUIRE(x_ld >= c && y_ld >= c && x_ld % e == 0 && y_ld % e == 0, "bad pixel stride");
  FSS_ALIGNED(x); FSS_ALIGNED(y);
  FSS_REQUIRE(pixels > 0, "bad shape");
  const size_t total = pixels * (c / e);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == FSS_F32)
    hipLaunchKernelGGL(copy_channels_kernel<float>, dim3(grid_for(total)), dim3(kBlock), 0, s,
                       (const float*)x, x_ld, (float*)y, y_ld, pixels, c);
  else if (dtype == FSS_F32_BF16X3)
    hipLaunchKernelGGL(copy_channels_kernel<f32s_t>, dim3(grid_for(total)), dim3(kBlock), 0, s,
                       (const f32s_t*)x, x_ld, (f32s_t*)y, y_ld, pixels, c);
  else
    hipLaunchKernelGGL(copy_channels_kernel<bf16_t>, dim3(grid_for(total)), dim3(kBlock), 0, s,
                       (const bf16_t*)x, x_ld, (bf16_t*)y, y_ld, pixels, c);
  FSS_LAUNCH_CHECK();
  return 0;
}

int fss_preprocess_u8(const uint8_t* x, void* y, int n, int h, int wd, int ho, int wo, int c_out,
                      const float* mean3, const float* scale3, int dtype, void* stream) {
  FSS_REQUIRE(x && y && mean3 && scale3, "null argument");
  FSS_DTYPE_OK(dtype);
  FSS_REQUIRE(n > 0 && h > 0 && wd > 0 && ho >= h && wo >= wd, "bad shape");
  FSS_REQUIRE(c_out >= 3 && c_out <= 64, "c_out must be in [3, 64]");
  PreArgs a;
  for (int i = 0; i < 3; ++i) { a.mean[i] = mean3[i]; a.scale[i] = scale3[i]; }
  const size_t total = (size_t)n * ho * wo;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == FSS_F32)
    hipLaunchKernelGGL(preprocess_kernel<float>, dim3(grid_for(total)), dim3(kBlock), 0, s, x,
                       (float*)y, n, h, wd, ho, wo, c_out, a);
  else if (dtype == FSS_F32_BF16X3)
    hipLaunchKernelGGL(preprocess_kernel<f32s_t>, dim3(grid_for(total)), dim3(kBlock), 0, s, x,
                       (f32s_t*)y, n, h, wd, ho, wo, c_out, a);
  else
    hipLaunchKernelGGL(preprocess_kernel<bf16_t>, dim3(grid_for(total)), dim3(kBlock), 0, s, x,
                       (bf16_t*)y, n, h, wd, ho, wo, c_out, a);
  FSS_LAUNCH_CHECK();
  return 0;
}

int fss_zoom2_border_gather(const void* x, void* tb, void* lr, void* cn, int n, int h, int wd, int c,
                            int dtype, void* stream) {
  FSS_REQUIRE(x && tb && lr && cn, "null argument");
  FSS_DTYPE_OK(dtype);
  FSS_CHUNK_OK(c, dtype);
  FSS_ALIGNED(x); FSS_ALIGNED(tb); FSS_ALIGNED(lr); FSS_ALIGNED(cn);
  FSS_REQUIRE(n > 0 && h >= 4 && wd >= 4, "bad shape (h, w >= 4)");
  const int cc = c / chunk_of(dtype);
  const size_t total = ((size_t)2 * n * 3 * wd + (size_t)2 * n * h * 3 + (size_t)4 * n * 9) * cc;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == FSS_F32)
    hipLaunchKernelGGL(zoom2_border_gather_kernel<float>, dim3(grid_for(total)), dim3(kBlock), 0, s,
                       (const float*)x, (float*)tb, (float*)lr, (float*)cn, n, h, wd, c);
  else if (dtype == FSS_F32_BF16X3)
    hipLaunchKernelGGL(zoom2_border_gather_kernel<f32s_t>, dim3(grid_for(total)), dim3(kBlock), 0, s,
                       (const f32s_t*)x, (f32s_t*)tb, (f32s_t*)lr, (f32s_t*)cn, n, h, wd, c);
  else
    hipLaunchKernelGGL(zoom2_border_gather_kernel<bf16_t>, dim3(grid_for(total)), dim3(kBlock), 0, s,
                       (const bf16_t*)x, (bf16_t*)tb, (bf16_t*)lr, (bf16_t*)cn, n, h, wd, c);
  FSS_LAUNCH_CHECK();
  return 0;
}

int fss_zoom2_conv_commute(const void* y, const void* ytb, const void* ylr, const void* ycn, void* r,
                           int n, int h, int wd, int c, int dtype, void* stream) {
  FSS_REQUIRE(y && ytb && ylr && ycn && r, "null argument");
  FSS_DTYPE_OK(dtype);
  FSS_CHUNK_OK(c, dtype);
  FSS_ALIGNED(y); FSS_ALIGNED(ytb); FSS_ALIGNED(ylr); FSS_ALIGNED(ycn); FSS_ALIGNED(r);
  FSS_REQUIRE(n > 0 && h >= 4 && wd >= 4, "bad shape (h, w >= 4)");
  const int cc = c / chunk_of(dtype);
  const size_t total = (size_t)n * (2 * h - 1) * (2 * wd - 1) * cc;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == FSS_F32)
    hipLaunchKernelGGL(zoom2_conv_commute_kernel<float>, dim3(grid_for(total)), dim3(kBlock), 0, s,
                       (const float*)y, (const float*)ytb, (const float*)ylr, (const float*)ycn,
                       (float*)r, n, h, wd, c);
  else if (dtype == FSS_F32_BF16X3)
    hipLaunchKernelGGL(zoom2_conv_commute_kernel<f32s_t>, dim3(grid_for(total)), dim3(kBlock), 0, s,
                       (const f32s_t*)y, (const f32s_t*)ytb, (const f32s_t*)ylr, (const f32s_t*)ycn,
                       (f32s_t*)r, n, h, wd, c);
  else
    hipLaunchKernelGGL(zoom2_conv_commute_kernel<bf16_t>, dim3(grid_for(total)), dim3(kBlock), 0, s,
                       (const bf16_t*)y, (const bf16_t*)ytb, (const bf16_t*)ylr, (const bf16_t*)ycn,
                       (bf16_t*)r, n, h, wd, c);
  FSS_LAUNCH_CHECK();
  return 0;
}

int fss_upsample_argmax(const float* logits, uint8_t* labels, float* logits_out, int n, int h,
                        int wd, int c, int x_ld, int ho, int wo, void* stream) {
  FSS_REQUIRE(logits && labels, "null argument");
  FSS_REQUIRE(n > 0 && h > 0 && wd > 0 && ho > 0 && wo > 0, "bad shape");
  FSS_REQUIRE(c > 0 && c <= 256, "class count must be in [1, 256]");
  FSS_REQUIRE(x_ld >= c, "x_ld < c");
  hipStream_t s = (hipStream_t)stream;
  // exact power-of-two zoom with 16-byte readable pixels: one thread per input cell
  const int zh = h > 1 ? (ho - 1) / (h - 1) : 0, zw = wd > 1 ? (wo - 1) / (wd - 1) : 0;
  const bool pow2 = zh == zw && (zh == 2 || zh == 4 || zh == 8) && (ho - 1) == zh * (h - 1) &&
                    (wo - 1) == zw * (wd - 1);
  if (pow2 && logits_out == nullptr && (x_ld & 3) == 0 && ((c + 3) & ~3) <= x_ld &&
      ((uintptr_t)logits & 15) == 0) {
    const int tiles_x = ceil_div(wd, kZoomTX), tiles_y = ceil_div(h, kZoomTY);
    const long long tiles = (long long)tiles_x * tiles_y * n;
    FSS_REQUIRE(tiles < 0x7fffffffLL, "too many tiles");
#define FSS_ZOOM(Z)                                                                                \
  hipLaunchKernelGGL(upsample_argmax_zoom_kernel<Z>, dim3((unsigned)tiles), dim3(kZoomTX * kZoomTY), 0, s, \
                     logits, labels, n, h, wd, c, x_ld, ho, wo, tiles_x, tiles_y)
    if (zh == 2) FSS_ZOOM(2); else if (zh == 4) FSS_ZOOM(4); else FSS_ZOOM(8);
#undef FSS_ZOOM
    FSS_LAUNCH_CHECK();
    return 0;
  }
  const float sh = resize_scale(h, ho, 1), sw = resize_scale(wd, wo, 1);
  const size_t total = (size_t)n * ho * wo;
  hipLaunchKernelGGL(upsample_argmax_kernel, dim3(grid_for(total)), dim3(kBlock), 0, s, logits,
                     labels, logits_out, n, h, wd, c, x_ld, ho, wo, sh, sw);
  FSS_LAUNCH_CHECK();
  return 0;
}

int fss_label_lut(const uint8_t* labels, float* y, const float* table, int n, int h, int wd,
                  int ho, int wo, int channels, int num_classes, void* stream) {
  FSS_REQUIRE(labels && y && table, "null argument");
  FSS_REQUIRE(n > 0 && h > 0 && wd > 0 && ho > 0 && wo > 0 && ho <= h && wo <= wd, "bad shape");
  FSS_REQUIRE(channels > 0 && num_classes > 0 && num_classes <= 256, "bad table");
  const size_t total = (size_t)n * ho * wo;
  hipLaunchKernelGGL(label_lut_kernel, dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream,
                     labels, y, table, n, h, wd, ho, wo, channels, num_classes);
  FSS_LAUNCH_CHECK();
  return 0;
}

int fss_confusion_accumulate(const uint8_t* pred, const uint8_t* labels, size_t count,
                             int num_classes, int ignore_label, int64_t* cm, void* stream) {
  FSS_REQUIRE(pred && labels && cm, "null argument");
  FSS_REQUIRE(num_classes > 0 && num_classes <= 64, "num_classes must be in [1, 64]");
  if (count == 0) return 0;
  const int threads = 256;
  const size_t smem = (size_t)(threads / 64) * num_classes * num_classes * sizeof(unsigned int);
  size_t blocks = (count + (size_t)threads * 16 - 1) / ((size_t)threads * 16);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(confusion_kernel, dim3((unsigned)blocks), dim3(threads), smem,
                     (hipStream_t)stream, pred, labels, count, num_classes, ignore_label,
                     (unsigned long long*)cm);
  FSS_LAUNCH_CHECK();
  return 0;
}

int fss_dwconv2d_nhwc(const void* x, const float* w, const float* bias, void* y, int n, int h,
                      int wd, int c, int kh, int kw, int stride, int dilation, int pad_top,
                      int pad_left, int ho, int wo, int act, int dtype, void* stream) {
  FSS_REQUIRE(x && w && y, "null argument");
  FSS_REQUIRE(dtype == FSS_F32 || dtype == FSS_BF16 || dtype == FSS_F32_BF16X3 || dtype == FSS_F16, "bad dtype");
  FSS_CHUNK_OK(c, dtype);
  FSS_ALIGNED(x); FSS_ALIGNED(y);
  FSS_REQUIRE(n > 0 && h > 0 && wd > 0 && ho > 0 && wo > 0, "bad shape");
  FSS_REQUIRE(kh > 0 && kw > 0 && stride > 0 && dilation > 0 && pad_top >= 0 && pad_left >= 0,
              "bad filter");
  hipStream_t s = (hipStream_t)stream;
  // 3x3 with the strides / dilations the MobileNet backbones use: the register-blocked kernel
  // (small problems keep one thread per output chunk: more threads fill the chip better)
  constexpr int R = 4;
  const int runs = ceil_div(wo, R);
  const size_t threads = (size_t)n * ho * runs * (c / chunk_of(dtype));
  if (kh == 3 && kw == 3 && ((uintptr_t)w & 15) == 0 && threads >= 300000) {
#define FSS_DW(T, ST, DL)                                                                            \
  hipLaunchKernelGGL((dwconv3x3_kernel<T, ST, DL, R>), dim3(grid_for(threads)), dim3(kBlock), 0, s,  \
                     (const T*)x, w, bias, (T*)y, n, h, wd, c, pad_top, pad_left, ho, wo, runs, act)
#define FSS_DW_ANY(T)                                                  \
  if (stride == 1 && dilation == 1) { FSS_DW(T, 1, 1); }               \
  else if (stride == 1 && dilation == 2) { FSS_DW(T, 1, 2); }          \
  else if (stride == 1 && dilation == 4) { FSS_DW(T, 1, 4); }          \
  else if (stride == 2 && dilation == 1) { FSS_DW(T, 2, 1); }          \
  else handled = false;
    bool handled = true;
    if (dtype == FSS_F32) { FSS_DW_ANY(float) } else if (dtype == FSS_F32_BF16X3) { FSS_DW_ANY(f32s_t) }
    else if (dtype == FSS_F16) { FSS_DW_ANY(f16_t) } else { FSS_DW_ANY(bf16_t) }
#undef FSS_DW_ANY
#undef FSS_DW
    if (handled) {
      FSS_LAUNCH_CHECK();
      return 0;
    }
  }
  const size_t total = (size_t)n * ho * wo * (c / chunk_of(dtype));
  if (dtype == FSS_F32)
    hipLaunchKernelGGL(dwconv_kernel<float>, dim3(grid_for(total)), dim3(kBlock), 0, s,
                       (const float*)x, w, bias, (float*)y, n, h, wd, c, kh, kw, stride, dilation,
                       pad_top, pad_left, ho, wo, act);
  else if (dtype == FSS_F32_BF16X3)
    hipLaunchKernelGGL(dwconv_kernel<f32s_t>, dim3(grid_for(total)), dim3(kBlock), 0, s,
                       (const f32s_t*)x, w, bias, (f32s_t*)y, n, h, wd, c, kh, kw, stride, dilation,
                       pad_top, pad_left, ho, wo, act);
  else if (dtype == FSS_F16)
    hipLaunchKernelGGL(dwconv_kernel<f16_t>, dim3(grid_for(total)), dim3(kBlock), 0, s,
                       (const f16_t*)x, w, bias, (f16_t*)y, n, h, wd, c, kh, kw, stride, dilation,
                       pad_top, pad_left, ho, wo, act);
  else
    hipLaunchKernelGGL(dwconv_kernel<bf16_t>, dim3(grid_for(total)), dim3(kBlock), 0, s,
                       (const bf16_t*)x, w, bias, (bf16_t*)y, n, h, wd, c, kh, kw, stride, dilation,
                       pad_top, pad_left, ho, wo, act);
  FSS_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
