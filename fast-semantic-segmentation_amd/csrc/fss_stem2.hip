// The first TWO convolutions of a full-resolution branch, fed straight from the
// uint8 frame, in one kernel (gfx950; bfloat16 or half precision, named by the kernel's template
// parameter H: the two differ in the MFMA instruction and in the roundings only).
//
// ICNet's high-resolution branch (architectures/icnet_architecture.py:213-226 in the
// reference) starts with 3x3 stride-2 convolutions on the 1025x2049 frame.  The
// output of the first one is the largest activation of the network (64 channels at
// 513x1025); writing it and reading it back costs several times more HBM traffic than
// everything else the two layers touch.  Here it only ever exists as a tile in LDS:
//
//   frame bytes --pad/mean/scale--> patch (LDS, 4 x bf16 per pixel)
//       conv1 3x3/2 + bias + act  --> y1 tile (LDS, bf16, (2*TH2+1) x 33 pixels x C1)
//       conv2 3x3/2 + bias + act  --> y (HBM), tile of TH2 x 16 output pixels
//
// Both convolutions run on the matrix cores as D[cout][pixel] = W[cout][k] * X[k][pixel]
// (weights are the MFMA "A" operand, 16 adjacent pixels the "B" operand), so a lane
// ends up with consecutive output channels of one pixel.  The second filter never
// touches LDS: wave (ch, pp) owns 32 output channels and a share of the tile's rows, and
// keeps its 9 * C1/32 * 2 weight fragments in registers for the whole (persistent) block.
//
// A block is a two-stage pipeline over its tiles with double-buffered patch and y1 tile:
// waves 0..NW-1 ("producers") run conv1 of tile k-1 while waves NW..2NW-1 ("consumers")
// run conv2 of tile k-2, stage the patch of tile k and request the bytes of tile k+1; one
// barrier per step separates the steps.  Every SIMD hosts one wave of each kind, so the
// first layer's index and conversion work overlaps the second layer's MFMAs.
//
// y1 is rounded to bf16 exactly where the unfused path stores it, so the result equals
// running the two layers one after the other up to the order of the float additions.
//
// conv1's K axis is 64 slots of 4 channels: step 0 holds taps 0-7 (lane group g: taps g and
// g + 4, so that the two lane groups served together by a ds_read_b64 read pixels of
// opposite column parity and never share a bank), step 1 holds tap 8 and, against a
// constant 1.0 in the B operand, the bias as a bf16 pair (high part + remainder), which
// makes the bias part of the MFMA sum.  Index arithmetic per pixel group (patch address of
// the lane's pixel, y1 address, position for the border mask) comes from a table built
// once per block.
//
// y1 tile layout: pixel (i, j) lives at pixel slot i*33 + (j&1)*17 + (j>>1) (columns
// split by parity: a stride-2 tap reads 16 consecutive slots) and its 16-byte chunk c
// is stored at chunk c ^ f(j>>1, j&1); f spreads the 16 lanes of every ds_read_b128
// service group, and the 8 lanes of every ds_write_b128 group, over all banks.
#include <algorithm>
#include <type_traits>
#include <utility>

#include "fss_common.h"

namespace fss {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// f(integral_constant<B>), ..., f(integral_constant<E-1>)
template <int B, int E, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// 16-byte LDS read whose latency the compiler does not wait for: the consumer calls
// lds_wait<N>(dst) before the first use, N = reads issued after this one (LDS returns in
// order).  The compiler's own lgkmcnt waits stay correct: they can only wait longer.
template <int OFFSET> __device__ __forceinline__ void lds_read16_async(u32x4_t& dst, unsigned lds_addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "n"(OFFSET) : "memory");
}
template <int N> __device__ __forceinline__ void lds_wait(u32x4_t& r) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(r) : "n"(N) : "memory");
}

struct Stem2Args {
  const uint8_t* x;
  const void* w1;
  const float* b1;
  const void* w2;
  const float* b2;
  void* y;
  int N, Hr, Wr, sub, H, W;
  int pad_t1, pad_l1, H1, W1, act1;
  int pad_t2, pad_l2, H2, W2, act2;
  int w2_kpad, y_ld;
  int tiles_x, tiles_y, total_tiles, unit_scale;
  float mean[3], scale[3];
};

constexpr int kS2TW = 16;                  // output columns of a tile (one MFMA pixel group per row)
constexpr int kS2Y1W = 2 * kS2TW + 1;      // y1 columns of a tile
constexpr int kS2Even = kS2TW + 1;         // even y1 columns of a tile (stored first in a row)
constexpr int kS2PW = 2 * kS2Y1W + 1;      // patch columns
constexpr int kS2K4 = 64;                  // conv1 K: 9 taps x 4 channels, padded to two MFMA steps
constexpr int kS2WRow = kS2K4 * 2 + 16;    // LDS bytes per conv1 weight row (+16: distinct banks)

// LDS map of a block
template <int KS1, int TH2> struct Stem2Lds {
  static constexpr int R1 = 2 * TH2 + 1;                       // y1 rows of a tile
  static constexpr int PP = (2 * R1 + 1) * kS2PW;              // patch pixels
  static constexpr int G1 = (R1 * kS2Y1W + 15) / 16;           // conv1 pixel groups of a tile
  static constexpr int OFF_B2 = 0, OFF_W1 = 256;
  static constexpr int OFF_TAB = OFF_W1 + 32 * KS1 * kS2WRow;
  static constexpr int OFF_PATCH = OFF_TAB + G1 * 64 * 8;
  static constexpr int PATCH_BYTES = (PP * 8 + 15) / 16 * 16;               // two of them
  static constexpr int OFF_Y1 = (OFF_PATCH + 2 * PATCH_BYTES + 255) / 256 * 256;
  static constexpr int Y1_BYTES = (R1 * kS2Y1W * 64 * KS1 + 255) / 256 * 256;   // two of them
  static constexpr int BYTES = OFF_Y1 + 2 * Y1_BYTES;
};

// KS1 = C1 / 32, CW = C2 / 32, TH2 = output rows per tile, NW = waves of each kind per block
template <typename H, int KS1, int CW, int TH2, int NW>
__global__ void __launch_bounds__(2 * NW * 64) __attribute__((amdgpu_waves_per_eu(2)))
stem2_conv_kernel(const Stem2Args a) {
  typedef Half16<H> HH;
  constexpr int NT = NW * 64;                // threads of each kind
  constexpr int R1 = 2 * TH2 + 1;            // y1 rows of a tile
  constexpr int PH = 2 * R1 + 1;             // patch rows
  constexpr int PP = PH * kS2PW;             // patch pixels
  constexpr int FS = (PP + NT - 1) / NT;     // patch pixels per thread
  constexpr int PXB = 64 * KS1;              // bytes of one y1 pixel
  constexpr int CPP = PXB / 16;              // chunks per y1 pixel (4 or 8)
  constexpr int Y1PIX = R1 * kS2Y1W;
  constexpr int G1 = (Y1PIX + 15) / 16;      // conv1 pixel groups of a tile
  constexpr int GS = NW / KS1;                // waves that share conv1's pixel groups
  constexpr int G1W = (G1 + GS - 1) / GS;    // ... groups per wave
  constexpr int PGW = TH2 * CW / NW;         // output rows per wave
  static_assert(TH2 * CW % NW == 0 && PGW >= 1, "rows must split evenly over the waves");
  static_assert(KS1 == 1 || KS1 == 2, "C1 must be 32 or 64");

  typedef Stem2Lds<KS1, TH2> L;
  static_assert(L::PP == PP && L::G1 == G1, "LDS map out of step with the kernel");
  constexpr int OFF_B2 = L::OFF_B2, OFF_W1 = L::OFF_W1, OFF_TAB = L::OFF_TAB, OFF_PATCH = L::OFF_PATCH,
                OFF_Y1 = L::OFF_Y1;

  extern __shared__ __attribute__((aligned(256))) char smem[];
  float* lb2 = reinterpret_cast<float*>(smem + OFF_B2);
  char* wl = smem + OFF_W1;
  uint2* tab = reinterpret_cast<uint2*>(smem + OFF_TAB);
  char* const patch0 = smem + OFF_PATCH;
  char* const y10 = smem + OFF_Y1;

  const int btid = threadIdx.x;              // thread of the block: only for the shared tables
  const int bwave = __builtin_amdgcn_readfirstlane(btid >> 6), lane = btid & 63;
  const bool consumer = bwave >= NW;
  const int wave = consumer ? bwave - NW : bwave;      // wave and thread inside its kind
  const int tid = wave * 64 + lane;
  const int p = lane & 15, g = lane >> 4;
  // XOR term of the chunk index of y1 pixel slot hc in the plane of column parity par
  auto swz = [](int hc, int par) {
    return CPP == 8 ? (((hc >> 1) ^ ((hc & 1) << 1) ^ (par << 2)) & 7)
                    : (((hc >> 2) ^ ((hc >> 1) & 1) ^ (par << 1)) & 3);
  };

  // ---- per-block tables
  if (btid < 64) lb2[btid] = btid < 32 * CW ? a.b2[btid] : 0.f;
  {  // conv1 weights, one padded row per output channel, 16 slots of 4 k values:
     // slot (st, g, half) = tap g + 4*half for st = 0; tap 8 for (1, 0, 0); the bias as
     // bf16 high part + remainder for (1, 0, 1); zero otherwise
    for (int idx = btid; idx < 32 * KS1 * 16; idx += 2 * NT) {
      const int co = idx >> 4, slot = idx & 15;
      const int st = slot >> 3, sg = (slot >> 1) & 3, half = slot & 1;
      const int tap = st == 0 ? sg + 4 * half : (sg == 0 && half == 0 ? 8 : -1);
      uint2 v = make_uint2(0u, 0u);
      if (tap >= 0) {
        v = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(a.w1) + ((size_t)co * kS2K4 * 2 + tap * 8));
      } else if (st == 1 && sg == 0) {
        const float b = a.b1[co];
        const unsigned hi = HH::bits(b);
        const unsigned lo = HH::bits(b - HH::value(hi));
        v.x = hi | (lo << 16);
      }
      *reinterpret_cast<uint2*>(wl + co * kS2WRow + (32 * st + 8 * sg + 4 * half) * 2) = v;
    }
  }
  // pixel group table: lane l of group grp is pixel 16*grp + (l & 15) = (i, j) of the y1 tile.
  //   .x = i << 20 | j << 14 | byte offset of patch pixel (2i, 2j)
  //   .y = byte offset of the lane's chunk (l >> 4) in the y1 tile (channel half 0), -1 past the tile
  for (int idx = btid; idx < G1 * 64; idx += 2 * NT) {
    const int l = idx & 63, pix = 16 * (idx >> 6) + (l & 15);
    const int pc = min(pix, Y1PIX - 1);
    const int i = pc / kS2Y1W, j = pc - i * kS2Y1W;
    const int hc = j >> 1, par = j & 1;
    uint2 e;
    e.x = ((unsigned)i << 20) | ((unsigned)j << 14) | (unsigned)(((2 * i) * kS2PW + 2 * j) * 8);
    e.y = pix < Y1PIX ? (unsigned)((i * kS2Y1W + par * kS2Even + hc) * PXB + (((l >> 4) ^ swz(hc, par)) & (CPP - 1)) * 16)
                      : 0xffffffffu;
    tab[idx] = e;
  }

  const int ch = wave % CW, pp = wave / CW;      // consumer: channel half, share of the rows

  // ---- conv1: wave (c1h, gs) makes 32 channels (c1h) of the pixel groups gs, gs + GS, ...;
  // lane (p, g) of group grp is pixel 16*grp + p of the y1 tile, channels 32*c1h + 8*g .. +7
  const int c1h = wave % KS1, gs = wave / KS1;
  // byte offsets, from the lane's patch pixel, of this lane group's conv1 taps: taps g and
  // g + 4 in k step 0; in step 1 tap 8 for lane group 0, else a pixel of the group's own
  // column parity (its weights are zero: any finite value will do)
  auto tap_off = [](int t) { return ((t / 3) * kS2PW + t % 3) * 8; };
  const int toA = tap_off(g), toB = tap_off(g + 4), toC = g == 0 ? tap_off(8) : (g & 1) * 8;
  const unsigned one_pair = g == 0 ? HH::kOnePair : 0u;   // B operand of the bias slots: 1.0, 1.0

  // ---- conv2 read offsets of this lane: pixel slot p + e (e = tap column >> 1), chunk 4*kh + g
  // (one per tap column s: plane s & 1, slot p + (s >> 1))
  int roff[3][KS1];
#pragma unroll
  for (int sc = 0; sc < 3; ++sc)
#pragma unroll
    for (int kh = 0; kh < KS1; ++kh)
      roff[sc][kh] = ((sc & 1) * kS2Even + p + (sc >> 1)) * PXB +
                     (((4 * kh + g) ^ swz(p + (sc >> 1), sc & 1)) & (CPP - 1)) * 16;

  const float lo1 = a.act1 == FSS_ACT_NONE ? -INFINITY : 0.f, hi1 = a.act1 == FSS_ACT_RELU6 ? 6.f : INFINITY;
  const float lo2 = a.act2 == FSS_ACT_NONE ? -INFINITY : 0.f, hi2 = a.act2 == FSS_ACT_RELU6 ? 6.f : INFINITY;

  // patch pixel tid + NT*k of this thread -> (row, column), -1 past the end
  auto patch_rc = [&](int k, int& pi, int& pj) {
    const int idx = tid + NT * k;
    pi = idx / kS2PW;
    pj = idx - pi * kS2PW;
    return idx < PP;
  };
  const int rowstride = a.sub * a.Wr * 3, colstride = a.sub * 3;
  const int hreal = min(a.H, (a.Hr + a.sub - 1) / a.sub), wreal = min(a.W, (a.Wr + a.sub - 1) / a.sub);

  const int tiles_per_img = a.tiles_x * a.tiles_y;
  struct Tile { int n, ty, tx; };
  auto decode = [&](int tile) {
    Tile t;
    t.n = tile / tiles_per_img;
    const int rem = tile - t.n * tiles_per_img;
    t.ty = rem / a.tiles_x;
    t.tx = rem - t.ty * a.tiles_x;
    return t;
  };
  const Tile step = decode(gridDim.x);
  auto advance = [&](Tile t) {
    t.tx += step.tx;
    if (t.tx >= a.tiles_x) { t.tx -= a.tiles_x; ++t.ty; }
    t.ty += step.ty;
    if (t.ty >= a.tiles_y) { t.ty -= a.tiles_y; ++t.n; }
    t.n += step.n;
    return t;
  };
  // first y1 row / column of a tile, first input row / column of its patch
  auto y1_i0 = [&](const Tile& t) { return t.ty * TH2 * 2 - a.pad_t2; };
  auto y1_j0 = [&](const Tile& t) { return t.tx * kS2TW * 2 - a.pad_l2; };

  // A tile whose whole patch lies on frame pixels (and not on the last pixel of a row, so
  // that 4-byte loads stay inside the tensor) needs no masks; so does a tile whose y1 rows
  // and columns all exist.  Both tests are block-uniform.
  auto patch_interior = [&](const Tile& t) {
    const int i0 = y1_i0(t) * 2 - a.pad_t1, j0 = y1_j0(t) * 2 - a.pad_l1;
    return i0 >= 0 && j0 >= 0 && i0 + PH <= hreal && j0 + kS2PW < wreal;
  };
  // Frame bytes of the patch.  Branch-free: slots that are not frame pixels read byte 0 of
  // the frame and are replaced at commit, so every load is in flight before any is used.
  auto fetch = [&](const Tile& t, unsigned (&u)[FS]) {
    const int i0 = y1_i0(t) * 2 - a.pad_t1, j0 = y1_j0(t) * 2 - a.pad_l1;
    const uint8_t* frame = a.x + (size_t)t.n * a.Hr * a.Wr * 3;
    if (patch_interior(t)) {
      const uint8_t* base = frame + (size_t)i0 * rowstride + (size_t)j0 * colstride;
#pragma unroll
      for (int k = 0; k < FS; ++k) {
        int pi, pj;
        const bool slot = patch_rc(k, pi, pj);
        const unsigned off = slot ? (unsigned)pi * (unsigned)rowstride + (unsigned)pj * (unsigned)colstride : 0u;
        u[k] = *reinterpret_cast<const unsigned*>(base + off);
      }
      return;
    }
    const unsigned last_word = t.n == a.N - 1 ? (unsigned)(a.Hr * a.Wr * 3 - 4) : 0xffffffffu;
#pragma unroll
    for (int k = 0; k < FS; ++k) {
      int pi, pj;
      const bool slot = patch_rc(k, pi, pj);
      const int i = i0 + pi, j = j0 + pj;
      const bool real = slot && (unsigned)i < (unsigned)hreal && (unsigned)j < (unsigned)wreal;
      const unsigned off = real ? (unsigned)i * (unsigned)rowstride + (unsigned)j * (unsigned)colstride : 0u;
      // one (unaligned) 4-byte load per pixel: its 3 bytes + 1 of the next; the very last
      // pixel of the tensor reads the 4 bytes that end with it instead
      const unsigned off4 = min(off, last_word);
      u[k] = *reinterpret_cast<const unsigned*>(frame + off4) >> ((off - off4) * 8u);
    }
  };
  // zero fill of the bounding box, mean / scale, zero padding of conv1, rounding
  auto commit = [&](const Tile& t, const unsigned (&u)[FS], char* patch) {
    const int i0 = y1_i0(t) * 2 - a.pad_t1, j0 = y1_j0(t) * 2 - a.pad_l1;
    auto store = [&](int k, bool slot, const float (&v)[3]) {
      if (slot) {
        uint2 o;
        o.x = HH::pack2(v[0], v[1]);
        o.y = HH::pack2(v[2], 0.f);
        *reinterpret_cast<uint2*>(patch + (size_t)(tid + NT * k) * 8) = o;
      }
    };
    if (patch_interior(t)) {
#pragma unroll
      for (int k = 0; k < FS; ++k) {
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          v[c] = (float)((u[k] >> (8 * c)) & 255u) - a.mean[c];
          if (!a.unit_scale) v[c] *= a.scale[c];
        }
        store(k, tid + NT * k < PP, v);
      }
      return;
    }
#pragma unroll
    for (int k = 0; k < FS; ++k) {
      int pi, pj;
      const bool slot = patch_rc(k, pi, pj);
      const int i = i0 + pi, j = j0 + pj;
      const bool real = (unsigned)i < (unsigned)hreal && (unsigned)j < (unsigned)wreal;
      const bool inside = (unsigned)i < (unsigned)a.H && (unsigned)j < (unsigned)a.W;
      const float m_real = real ? 1.f : 0.f;
      float v[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        v[c] = fmaf((float)((u[k] >> (8 * c)) & 255u), m_real, -a.mean[c]);    // exact: (u or 0) - mean
        if (!a.unit_scale) v[c] *= a.scale[c];
        v[c] = inside ? v[c] : 0.f;
      }
      store(k, slot, v);
    }
  };

  // ---- the two stages
  auto conv1 = [&](const Tile& now, const char* patch, char* y1) {
    uint4 w1f[2][2];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int nj = 0; nj < 2; ++nj)
        w1f[st][nj] = *reinterpret_cast<const uint4*>(
            wl + (32 * c1h + 8 * (p >> 2) + 4 * nj + (p & 3)) * kS2WRow + (32 * st + 8 * g) * 2);
    const int i1_0 = y1_i0(now), j1_0 = y1_j0(now);
    // pixels of the tile that lie outside conv1's output are conv2's zero padding
    const bool y1_interior = i1_0 >= 0 && j1_0 >= 0 && i1_0 + R1 <= a.H1 && j1_0 + kS2Y1W <= a.W1;
    // Two pixel groups per iteration (four independent MFMA chains, the epilogue of one
    // group under the MFMAs of the other); table entries are read two iterations ahead and
    // the B operands (patch pixels of this lane's taps) one iteration ahead, so that no LDS
    // latency sits in front of the MFMAs.  Groups past the tile are computed on clamped
    // indices and not stored.
    auto load_b = [&](const uint2& e, uint2 (&b)[3]) {
      const char* src = patch + (e.x & 0x3fffu);
      b[0] = *reinterpret_cast<const uint2*>(src + toA);
      b[1] = *reinterpret_cast<const uint2*>(src + toB);
      b[2] = *reinterpret_cast<const uint2*>(src + toC);
    };
    auto entry = [&](int grp) { return tab[min(grp, G1 - 1) * 64 + lane]; };
    constexpr int NPAIR = (G1W + 1) / 2;
    uint2 en[2] = {entry(gs), entry(gs + GS)};
    uint2 en2[2] = {entry(gs + 2 * GS), entry(gs + 3 * GS)};
    uint2 bq[2][3];
    load_b(en[0], bq[0]);
    load_b(en[1], bq[1]);
#pragma unroll 1
    for (int t = 0; t < NPAIR; ++t) {
      const int grp0 = gs + GS * (2 * t);
      const uint2 ec[2] = {en[0], en[1]};
      uint2 bc[2][3];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int q = 0; q < 3; ++q) bc[u][q] = bq[u][q];
      en[0] = en2[0];
      en[1] = en2[1];
      en2[0] = entry(grp0 + 4 * GS);
      en2[1] = entry(grp0 + 5 * GS);
      load_b(en[0], bq[0]);
      load_b(en[1], bq[1]);
      f32x4 acc[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[u][0] = acc[u][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const uint4 b = st == 0 ? make_uint4(bc[u][0].x, bc[u][0].y, bc[u][1].x, bc[u][1].y)
                                  : make_uint4(bc[u][2].x, bc[u][2].y, one_pair, 0u);
#pragma unroll
          for (int nj = 0; nj < 2; ++nj)
            acc[u][nj] = mfma_16x16x32<H>(w1f[st][nj], b, acc[u][nj]);
        }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float v[8] = {acc[u][0][0], acc[u][0][1], acc[u][0][2], acc[u][0][3],
                      acc[u][1][0], acc[u][1][1], acc[u][1][2], acc[u][1][3]};
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = __builtin_amdgcn_fmed3f(v[q], lo1, hi1);
        uint4 o = make_uint4(HH::pack2(v[0], v[1]), HH::pack2(v[2], v[3]), HH::pack2(v[4], v[5]),
                             HH::pack2(v[6], v[7]));
        if (!y1_interior) {
          const int ci = (int)(ec[u].x >> 20), cj = (int)((ec[u].x >> 14) & 63u);
          const bool inside = (unsigned)(i1_0 + ci) < (unsigned)a.H1 && (unsigned)(j1_0 + cj) < (unsigned)a.W1;
          const unsigned m = inside ? 0xffffffffu : 0u;
          o.x &= m; o.y &= m; o.z &= m; o.w &= m;
        }
        if (grp0 + GS * u < G1 && ec[u].y != 0xffffffffu)
          *reinterpret_cast<uint4*>(y1 + (ec[u].y ^ (unsigned)(c1h << 6))) = o;
      }
    }
  };
  // ---- pipeline over this block's tiles blockIdx.x + k*gridDim.x, k < nt.  Step k:
  //   consumers: conv2 of tile k-2 (y1 buffer k&1), then the patch of tile k (bytes requested
  //              one step ago) into patch buffer k&1, then the request for tile k+1;
  //   producers: conv1 of tile k-1 (patch buffer (k-1)&1 -> y1 buffer (k-1)&1).
  // The barrier that ends a step orders every buffer hand-over (and, in step 0, the tables).
  // The two kinds run separate loops with the same number of barriers, so that the second
  // filter's registers are live only in the consumers' code.
  const int nt = (a.total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  Tile tk = decode(blockIdx.x), tk1 = advance(tk), tm1 = tk, tm2 = tk;
  if (consumer) {
    // conv2 weights: this wave's 32 output channels, in registers for good.
    // A-row r of tile nj is channel 32*ch + 8*(r>>2) + 4*nj + (r&3): lane group g then holds
    // the 8 consecutive channels 32*ch + 8*g .. +7 of its pixel.
    uint4 w2f[9][KS1][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int kh = 0; kh < KS1; ++kh)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
          const int co = 32 * ch + 8 * (p >> 2) + 4 * nj + (p & 3);
          w2f[t][kh][nj] = *reinterpret_cast<const uint4*>(
              reinterpret_cast<const bf16_t*>(a.w2) + (size_t)co * a.w2_kpad + t * (32 * KS1) + kh * 32 + 8 * g);
        }
    auto conv2 = [&](const Tile& now, const char* y1) {
      f32x4 acc2[PGW][2];
  #pragma unroll
      for (int t = 0; t < PGW; ++t)
  #pragma unroll
        for (int nj = 0; nj < 2; ++nj) acc2[t][nj] = f32x4{0.f, 0.f, 0.f, 0.f};
      // B operand number i = ((r*3 + s)*KS1 + kh)*PGW + t.  RING of them are kept in flight
      // (reads the compiler does not wait for), so that the LDS latency hides behind the MFMAs
      // of earlier operands.
      constexpr int NB2 = 9 * KS1 * PGW, RING = NB2 < 8 ? NB2 : 8;
      const unsigned y1_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)y1 +
                              (unsigned)((2 * pp * PGW) * (kS2Y1W * PXB));
      unsigned baddr[3][KS1];
  #pragma unroll
      for (int sc = 0; sc < 3; ++sc)
  #pragma unroll
        for (int kh = 0; kh < KS1; ++kh) baddr[sc][kh] = y1_lds + (unsigned)roff[sc][kh];
      u32x4_t ring[RING];
      auto issue = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int t = i % PGW, kh = (i / PGW) % KS1, rs = i / (PGW * KS1);
        lds_read16_async<((2 * t + rs / 3) * kS2Y1W) * PXB>(ring[i % RING], baddr[rs % 3][kh]);
      };
      auto step = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int t = i % PGW, kh = (i / PGW) % KS1, rs = i / (PGW * KS1);
        constexpr int younger = (i + RING - 1 < NB2 - 1 ? i + RING - 1 : NB2 - 1) - i;
        lds_wait<younger>(ring[i % RING]);
        const uint4 b = __builtin_bit_cast(uint4, ring[i % RING]);
  #pragma unroll
        for (int nj = 0; nj < 2; ++nj) acc2[t][nj] = mfma_16x16x32<H>(w2f[rs][kh][nj], b, acc2[t][nj]);
        if constexpr (i + RING < NB2) issue(std::integral_constant<int, i + RING>{});
      };
      static_for<0, RING>(issue);
      static_for<0, NB2>(step);
      // lane (p, g): channels 32*ch + 8*g .. +7 of pixel (row, p)
      const int ow = now.tx * kS2TW + p;
      const float4 c0 = *reinterpret_cast<const float4*>(lb2 + 32 * ch + 8 * g);
      const float4 c1 = *reinterpret_cast<const float4*>(lb2 + 32 * ch + 8 * g + 4);
  #pragma unroll
      for (int t = 0; t < PGW; ++t) {
        const int oh = now.ty * TH2 + pp * PGW + t;
        if (oh < a.H2 && ow < a.W2) {
          const float v[8] = {acc2[t][0][0] + c0.x, acc2[t][0][1] + c0.y, acc2[t][0][2] + c0.z, acc2[t][0][3] + c0.w,
                              acc2[t][1][0] + c1.x, acc2[t][1][1] + c1.y, acc2[t][1][2] + c1.z, acc2[t][1][3] + c1.w};
          unsigned o[4];
  #pragma unroll
          for (int e = 0; e < 4; ++e)
            o[e] = HH::pack2(__builtin_amdgcn_fmed3f(v[2 * e], lo2, hi2),
                             __builtin_amdgcn_fmed3f(v[2 * e + 1], lo2, hi2));
          bf16_t* dst = reinterpret_cast<bf16_t*>(a.y) +
                        (((size_t)now.n * a.H2 + oh) * a.W2 + ow) * a.y_ld + 32 * ch + 8 * g;
          store16_nt(dst, o[0], o[1], o[2], o[3]);
        }
      }
    };

    unsigned ureg[FS];
    fetch(tk, ureg);
    for (int k = 0; k < nt + 2; ++k) {
      if (k >= 2) conv2(tm2, y10 + (k & 1) * L::Y1_BYTES);
      if (k < nt) commit(tk, ureg, patch0 + (k & 1) * L::PATCH_BYTES);
      if (k + 1 < nt) fetch(tk1, ureg);
      __builtin_amdgcn_s_barrier();
      tm2 = tm1; tm1 = tk; tk = tk1; tk1 = advance(tk1);
    }
  } else {
    for (int k = 0; k < nt + 2; ++k) {
      if (k >= 1 && k <= nt)
        conv1(tm1, patch0 + ((k - 1) & 1) * L::PATCH_BYTES, y10 + ((k - 1) & 1) * L::Y1_BYTES);
      __builtin_amdgcn_s_barrier();
      tm1 = tk; tk = tk1; tk1 = advance(tk1);
    }
  }
}

template <typename H, int KS1, int CW, int TH2, int NW>
static int launch_stem2(const Stem2Args& a0, hipStream_t stream) {
  Stem2Args a = a0;
  constexpr size_t smem = Stem2Lds<KS1, TH2>::BYTES;
  static_assert(smem <= 160 * 1024, "tile too large for LDS");
  a.tiles_x = ceil_div(a.W2, kS2TW);
  a.tiles_y = ceil_div(a.H2, TH2);
  const long long total = (long long)a.tiles_x * a.tiles_y * a.N;
  if (total <= 0 || total >= 0x7fffffffLL) {
    set_error("fss_stem2_conv_u8: too many tiles");
    return 2;
  }
  a.total_tiles = (int)total;
  auto kern = stem2_conv_kernel<H, KS1, CW, TH2, NW>;
  static bool attr_set = false;   // per instantiation
  if (smem > 64 * 1024 && !attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) {
      set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
      return 1;
    }
    attr_set = true;
  }
  // persistent blocks: as many as are resident on the chip at once, at most one per tile
  static int per_cu = 0, num_cu = 0;
  if (num_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      set_error("fss_stem2_conv_u8: cannot query the device");
      return 1;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 2 * NW * 64,
                                                     smem) != hipSuccess || per_cu < 1)
      per_cu = 1;
    num_cu = prop.multiProcessorCount;
  }
  const int blocks = std::min(a.total_tiles, num_cu * per_cu);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(2 * NW * 64), smem, stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error(std::string("fss_stem2_conv_u8 launch: ") + hipGetErrorString(e));
    return 3;
  }
  return 0;
}

}  // namespace fss

using namespace fss;

extern "C" int fss_stem2_conv_u8(const uint8_t* x, const void* w1_packed, const float* bias1,
                                 const void* w2_packed, const float* bias2, void* y, int n, int hr,
                                 int wr, int sub, int h, int w, const float* mean3, const float* scale3,
                                 int c1, int pad_top1, int pad_left1, int h1, int w1, int act1, int c2,
                                 int pad_top2, int pad_left2, int h2, int w2, int act2, int dtype,
                                 int y_ld, void* stream) {
  FSS_REQUIRE(x && w1_packed && bias1 && w2_packed && bias2 && y && mean3 && scale3, "null argument");
  FSS_REQUIRE(dtype == FSS_BF16 || dtype == FSS_F16, "only bfloat16 and half precision are supported");
  FSS_REQUIRE(n > 0 && hr > 0 && wr > 0 && h > 0 && w > 0 && h1 > 0 && w1 > 0 && h2 > 0 && w2 > 0, "bad shape");
  FSS_REQUIRE(sub >= 1 && sub <= 8, "sub must be in [1, 8]");
  FSS_REQUIRE((long long)hr * wr * 3 >= 4, "frame too small");
  FSS_REQUIRE(c1 == 32 || c1 == 64, "c1 must be 32 or 64");
  FSS_REQUIRE(c2 == 32 || c2 == 64, "c2 must be 32 or 64");
  FSS_REQUIRE(pad_top1 >= 0 && pad_left1 >= 0 && pad_top2 >= 0 && pad_left2 >= 0 && pad_top1 <= 2 &&
                  pad_left1 <= 2 && pad_top2 <= 2 && pad_left2 <= 2, "padding must be in [0, 2]");
  for (int act : {act1, act2})
    FSS_REQUIRE(act == FSS_ACT_NONE || act == FSS_ACT_RELU || act == FSS_ACT_RELU6, "bad act");
  FSS_REQUIRE(y_ld >= c2 && y_ld % 8 == 0, "y_ld must be a multiple of 8 and at least c2");
  // both layers: 3x3, stride 2; every output pixel's first tap starts inside the padded input
  FSS_REQUIRE((long long)(h1 - 1) * 2 - pad_top1 < h && (long long)(w1 - 1) * 2 - pad_left1 < w,
              "first output size exceeds the input");
  FSS_REQUIRE((long long)(h2 - 1) * 2 - pad_top2 < h1 && (long long)(w2 - 1) * 2 - pad_left2 < w1,
              "second output size exceeds the first");
  FSS_REQUIRE(((uintptr_t)w1_packed & 15) == 0 && ((uintptr_t)w2_packed & 15) == 0 && ((uintptr_t)y & 15) == 0,
              "unaligned pointer");
  FSS_REQUIRE((long long)hr * wr * 3 < 0x7fffffffLL && (long long)sub * wr * 3 < (1 << 24) && hr < (1 << 24),
              "frame too large");
  Stem2Args a;
  a.x = x; a.w1 = w1_packed; a.b1 = bias1; a.w2 = w2_packed; a.b2 = bias2; a.y = y;
  a.N = n; a.Hr = hr; a.Wr = wr; a.sub = sub; a.H = h; a.W = w;
  a.pad_t1 = pad_top1; a.pad_l1 = pad_left1; a.H1 = h1; a.W1 = w1; a.act1 = act1;
  a.pad_t2 = pad_top2; a.pad_l2 = pad_left2; a.H2 = h2; a.W2 = w2; a.act2 = act2;
  int k_pad = 0, cout_pad = 0;
  if (fss_stem_pack_dims(3, 3, c1, dtype, &k_pad, &cout_pad)) return 2;
  FSS_REQUIRE(k_pad == kS2K4 && cout_pad == c1, "unexpected first-layer packing");
  if (fss_conv_pack_dims(c1, 3, 3, c2, dtype, &k_pad, &cout_pad)) return 2;
  a.w2_kpad = k_pad;
  a.y_ld = y_ld;
  a.tiles_x = a.tiles_y = a.total_tiles = 0;
  for (int i = 0; i < 3; ++i) { a.mean[i] = mean3[i]; a.scale[i] = scale3[i]; }
  a.unit_scale = scale3[0] == 1.f && scale3[1] == 1.f && scale3[2] == 1.f;
  hipStream_t s = (hipStream_t)stream;
  // tiles of 4 x 16 output pixels, 4 + 4 waves: one block per CU
  if (dtype == FSS_F16) {
    switch ((c1 / 32) * 10 + c2 / 32) {
      case 11: return launch_stem2<f16_t, 1, 1, 4, 4>(a, s);
      case 12: return launch_stem2<f16_t, 1, 2, 4, 4>(a, s);
      case 21: return launch_stem2<f16_t, 2, 1, 4, 4>(a, s);
      default: return launch_stem2<f16_t, 2, 2, 4, 4>(a, s);
    }
  }
  switch ((c1 / 32) * 10 + c2 / 32) {
    case 11: return launch_stem2<bf16_t, 1, 1, 4, 4>(a, s);
    case 12: return launch_stem2<bf16_t, 1, 2, 4, 4>(a, s);
    case 21: return launch_stem2<bf16_t, 2, 1, 4, 4>(a, s);
    default: return launch_stem2<bf16_t, 2, 2, 4, 4>(a, s);
  }
}
