// 3x3 convolution of a 64-channel tensor with the whole filter resident in LDS (gfx950; bfloat16
// or half precision: the template parameter H names the 16-bit type, everything but the MFMA
// instruction and the rounding of the outputs is the same):
//
//   y = act(conv3x3(x, W) + bias),  cin = 64, cout = 64 or 128, any stride / dilation / padding
//
// These are the layers of the networks where the implicit-GEMM kernel is bound by staging,
// not by arithmetic: with 64 input channels the K axis is 576 long, every activation byte
// goes through the global -> LDS path nine times and the filter (72 or 144 KiB) is
// re-streamed for every 128 or 256 output pixels (the second convolution of the first
// ResNet block's units, third_party/dilated_resnet_v1.py:95-97, and `Conv_2` of ICNet's
// full-resolution branch, architectures/icnet_architecture.py:220-222 in the reference).
// Here the filter is read once per block and no activation touches LDS: the products run
// on the matrix cores as D[channel][pixel] = W[channel][k] * X[k][pixel], a wave owns
// consecutive output pixels, and lane (r, g) loads the 16 bytes of channels
// 32 h + 8 g .. + 7 of the input pixel under tap t of its output pixel r straight into the
// B operand of k step 2 t + h (taps on the padding read a zero chunk instead).  The rows
// of a 16-row weight tile are ordered so that the 8 accumulator values a lane owns in the
// tile pair (2 u, 2 u + 1) are the 8 consecutive channels 32 u + 8 g .. + 7 of its pixel:
// one 16-byte store per pair.  A wave works on two such groups of 16 pixels at once, so that
// every weight fragment read from LDS feeds two MFMAs.  Blocks are persistent; the operand
// loads run two taps ahead of the MFMAs through a ring of registers, across pass boundaries,
// and the loop body has no branch around a load or a store (see fss_chain.hip for why).
#include <algorithm>
#include <type_traits>

#include "fss_common.h"

namespace fss {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

// (tensor pointers as raw 16-bit words: bfloat16 or half, the kernels' template parameter says which)
struct Conv3Args {
  const bf16_t* x;
  const bf16_t* w;
  const float* bias;
  bf16_t* y;
  long long M;                  // n * ho * wo
  int N, H, W, Ho, Wo, stride, dil, pad_t, pad_l, x_ld, y_ld, act;
};

// 16 zero bytes every padding tap is fetched from
__device__ __attribute__((aligned(16))) unsigned int g_c3_zero[4] = {0u, 0u, 0u, 0u};

constexpr int kC3Waves = 8;                  // waves per block
constexpr int kC3Pixels = 32 * kC3Waves;     // pixels per block and pass (two groups of 16 per wave)
constexpr int kC3Steps = 18;                 // k steps of 32 channels: 9 taps x 2 halves

// byte offset, inside the filter image in LDS, of chunk c (8 k values) of k step s of row rho
// of tile j: one MFMA step of one tile is 1 KiB of consecutive bytes (16 rows x 64 bytes),
// chunks XOR-swizzled with the row so that every ds_read_b128 service group spreads over all
// banks
__device__ __forceinline__ int c3_lds_offset(int j, int s, int rho, int c) {
  return ((j * kC3Steps + s) * 16 + rho) * 64 + ((c ^ ((0 - (rho >> 2)) & 3)) << 4);
}
// channel held by row rho of 16-row tile j (8 consecutive channels per lane and tile pair)
__device__ __forceinline__ int c3_channel(int j, int rho) {
  return 32 * (j >> 1) + 8 * (rho >> 2) + 4 * (j & 1) + (rho & 3);
}

// C1P = cout / 32
template <typename H, int C1P>
__global__ void __launch_bounds__(kC3Waves * 64) conv3x3_c64_kernel(const Conv3Args a) {
  constexpr int C1 = 32 * C1P, NT = kC3Waves * 64;
  constexpr int OFF_B = 2 * C1P * kC3Steps * 1024;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wl = smem;
  float* bl = reinterpret_cast<float*>(smem + OFF_B);

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int r = lane & 15, g = lane >> 4;

  // filter: [cout][576] with k = tap * 64 + channel; chunk c of a row = k 8 c .. 8 c + 7
  for (int idx = tid; idx < C1 * 72; idx += NT) {
    const int row = idx / 72, c = idx - row * 72;
    *reinterpret_cast<uint4*>(wl + c3_lds_offset(row >> 4, c >> 2, row & 15, c & 3)) =
        *reinterpret_cast<const uint4*>(a.w + (size_t)c3_channel(row >> 4, row & 15) * 576 + c * 8);
  }
  for (int idx = tid; idx < C1; idx += NT) bl[idx] = a.bias != nullptr ? a.bias[idx] : 0.f;
  __syncthreads();

  const float lo = a.act == FSS_ACT_NONE ? -INFINITY : 0.f, hi = a.act == FSS_ACT_RELU6 ? 6.f : INFINITY;
  const char* wfrag = wl + c3_lds_offset(0, 0, r, g);
  const long long passes = (a.M + kC3Pixels - 1) / kC3Pixels;
  const int HoWo = a.Ho * a.Wo;
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_c3_zero);
  constexpr int NTILE = 2 * C1P;

  // where this lane's output pixel of group `grp` (16 pixels each, two groups per wave and pass)
  // starts reading: first byte of its frame (+ the lane's channel group), row and column of tap
  // (0, 0); the output pixel itself.  Past the last pass / pixel: the last one again.
  struct Coord { const bf16_t* frame; int off00; unsigned ok; long long m; };
  auto coord = [&](long long pass, int grp) {
    pass = pass < passes ? pass : passes - 1;
    long long m = pass * kC3Pixels + wave * 32 + grp * 16 + r;
    m = m < a.M ? m : a.M - 1;
    const int n = (int)(m / HoWo), rem = (int)(m - (long long)n * HoWo);
    const int oh = rem / a.Wo, ow = rem - oh * a.Wo;
    const int ih0 = oh * a.stride - a.pad_t, iw0 = ow * a.stride - a.pad_l;
    Coord c;
    c.frame = a.x + ((size_t)n * a.H * a.W) * a.x_ld + 8 * g;
    // element offset of tap (0, 0) inside the frame (may be negative; only used for real taps)
    c.off00 = (ih0 * a.W + iw0) * a.x_ld;
    // bit t: tap t = (kr, ks) lies on the frame
    unsigned rows = 0, cols = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      rows |= ((unsigned)(ih0 + k * a.dil) < (unsigned)a.H ? 7u : 0u) << (3 * k);
      cols |= ((unsigned)(iw0 + k * a.dil) < (unsigned)a.W ? 0x49u : 0u) << k;
    }
    c.ok = rows & cols;
    c.m = m;
    return c;
  };
  // element offset of tap t from tap (0, 0): the same for every pixel (scalar registers)
  int dtap[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) dtap[t] = ((t / 3) * a.dil * a.W + (t % 3) * a.dil) * a.x_ld;
  // the two operand chunks (channel halves) of tap t of a pixel; taps on the padding read zeros
  auto load_tap = [&](const Coord& c, int t, uint4 (&dst)[2]) {
    const bool ok = (c.ok >> t) & 1u;
    const bf16_t* p = ok ? c.frame + (c.off00 + dtap[t]) : zero;
    dst[0] = *reinterpret_cast<const uint4*>(p);
    dst[1] = *reinterpret_cast<const uint4*>(p + (ok ? 32 : 0));
  };

  // A ring of three taps per pixel group: tap t is consumed from slot t % 3, which is then
  // refilled with tap t + 3 (of the next pass from tap 6 on), so the loads of two taps are in
  // flight under the 4 * NTILE MFMAs of a third.  Nine taps are three turns of the ring: every
  // index below is a compile-time constant.
  const long long step = gridDim.x;
  long long pass = blockIdx.x;
  if (pass >= passes) return;
  uint4 ring[3][2][2];
  Coord c0 = coord(pass, 0), c1 = coord(pass, 1);
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    load_tap(c0, t, ring[t][0]);
    load_tap(c1, t, ring[t][1]);
  }
  while (true) {
    const Coord n0 = coord(pass + step, 0), n1 = coord(pass + step, 1);
    f32x4 acc[NTILE][2];
#pragma unroll
    for (int j = 0; j < NTILE; ++j) {
      const float4 b = *reinterpret_cast<const float4*>(bl + 32 * (j >> 1) + 8 * g + 4 * (j & 1));
      acc[j][0] = acc[j][1] = f32x4{b.x, b.y, b.z, b.w};
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int j = 0; j < NTILE; ++j) {
          const uint4 wa = *reinterpret_cast<const uint4*>(wfrag + (j * kC3Steps + 2 * t + h) * 1024);
#pragma unroll
          for (int grp = 0; grp < 2; ++grp)
            acc[j][grp] = mfma_16x16x32<H>(wa, ring[t % 3][grp][h], acc[j][grp]);
        }
        // keep the weight reads of later k steps from being hoisted above this point (they
        // would all be live at once)
        __builtin_amdgcn_sched_barrier(0);
      }
      if (t + 3 < 9) {
        load_tap(c0, t + 3, ring[t % 3][0]);
        load_tap(c1, t + 3, ring[t % 3][1]);
      } else {
        load_tap(n0, t + 3 - 9, ring[t % 3][0]);
        load_tap(n1, t + 3 - 9, ring[t % 3][1]);
      }
    }
#pragma unroll
    for (int grp = 0; grp < 2; ++grp) {
      bf16_t* py = a.y + (size_t)(grp == 0 ? c0.m : c1.m) * a.y_ld + 8 * g;
#pragma unroll
      for (int u = 0; u < C1P; ++u) {
        float v[8] = {acc[2 * u][grp][0], acc[2 * u][grp][1], acc[2 * u][grp][2], acc[2 * u][grp][3],
                      acc[2 * u + 1][grp][0], acc[2 * u + 1][grp][1], acc[2 * u + 1][grp][2], acc[2 * u + 1][grp][3]};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], lo, hi);
        store16_nt(py + 32 * u, Half16<H>::pack2(v[0], v[1]), Half16<H>::pack2(v[2], v[3]),
                   Half16<H>::pack2(v[4], v[5]), Half16<H>::pack2(v[6], v[7]));
      }
    }
    c0 = n0;
    c1 = n1;
    pass += step;
    if (pass >= passes) break;
  }
}

// Stride 1, dilation 1: the three horizontal taps of 16 consecutive pixels of one image row are
// 18 consecutive input pixels, so tap (kr, 0) of the group's lanes plus the two pixels after
// them are all a filter row needs; taps (kr, 1) and (kr, 2) are the same registers shifted by
// one and two lanes (DPP row shifts: the 16 lanes of a DPP row are exactly one pixel group of
// one channel group).  Two loads per filter row, channel half and group (the second with only
// lanes 14 and 15 active) instead of three: the load path, which bounds this kernel, carries
// 0.4x the bytes.  A group whose pixels wrap around the end of an image row takes the plain
// three loads.  The ring is two filter rows deep; two passes (six rows) per trip keep its
// indices compile-time constants.
//
// Stride 2: tap (kr, 2) of a pixel is tap (kr, 0) of its right neighbour, so tap 0 and tap 1 are
// loaded by every lane, tap 2 by lane 15 only, and the other lanes take tap 2 from the next
// lane's tap 0.
template <typename H, int C1P, int STRIDE>
__global__ void __launch_bounds__(kC3Waves * 64) conv3x3_c64_shared_kernel(const Conv3Args a) {
  static_assert(STRIDE == 1 || STRIDE == 2, "lane sharing is worked out for strides 1 and 2");
  constexpr int C1 = 32 * C1P, NT = kC3Waves * 64;
  constexpr int OFF_B = 2 * C1P * kC3Steps * 1024;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wl = smem;
  float* bl = reinterpret_cast<float*>(smem + OFF_B);
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int r = lane & 15, g = lane >> 4;
  for (int idx = tid; idx < C1 * 72; idx += NT) {
    const int row = idx / 72, c = idx - row * 72;
    *reinterpret_cast<uint4*>(wl + c3_lds_offset(row >> 4, c >> 2, row & 15, c & 3)) =
        *reinterpret_cast<const uint4*>(a.w + (size_t)c3_channel(row >> 4, row & 15) * 576 + c * 8);
  }
  for (int idx = tid; idx < C1; idx += NT) bl[idx] = a.bias != nullptr ? a.bias[idx] : 0.f;
  __syncthreads();

  const float lo = a.act == FSS_ACT_NONE ? -INFINITY : 0.f, hi = a.act == FSS_ACT_RELU6 ? 6.f : INFINITY;
  const char* wfrag = wl + c3_lds_offset(0, 0, r, g);
  const long long passes = (a.M + kC3Pixels - 1) / kC3Pixels;
  const int HoWo = a.Ho * a.Wo;
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_c3_zero);
  constexpr int NTILE = 2 * C1P;

  struct Coord { const bf16_t* frame; int off00; unsigned ok; long long m; bool wrap; };
  auto coord = [&](long long pass, int grp) {
    pass = pass < passes ? pass : passes - 1;
    const long long m_real = pass * kC3Pixels + wave * 32 + grp * 16 + r;
    const long long m = m_real < a.M ? m_real : a.M - 1;
    const int n = (int)(m / HoWo), rem = (int)(m - (long long)n * HoWo);
    const int oh = rem / a.Wo, ow = rem - oh * a.Wo;
    const int ih0 = oh * STRIDE - a.pad_t, iw0 = ow * STRIDE - a.pad_l;
    Coord c;
    c.frame = a.x + ((size_t)n * a.H * a.W) * a.x_ld + 8 * g;
    c.off00 = (ih0 * a.W + iw0) * a.x_ld;
    unsigned rows = 0, cols = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      rows |= ((unsigned)(ih0 + k) < (unsigned)a.H ? 7u : 0u) << (3 * k);
      cols |= ((unsigned)(iw0 + k) < (unsigned)a.W ? 0x49u : 0u) << k;
    }
    c.ok = rows & cols;
    c.m = m;
    // the 16 pixels of the group are 16 consecutive columns of one image row (and none of them
    // stands in for a pixel past the end): lane 0's column decides, for every lane alike
    const int ow_first = __builtin_amdgcn_readfirstlane(ow);     // lane 0 of the wave = pixel 0 of the group
    const long long m_first = (long long)pass * kC3Pixels + wave * 32 + grp * 16;
    c.wrap = ow_first + 15 >= a.Wo || m_first + 15 >= a.M;
    return c;
  };
  int dtap[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) dtap[t] = ((t / 3) * a.W + (t % 3)) * a.x_ld;
  // raw registers of one filter row of one group: shared form {tap 0 of every lane, tap 2 of
  // lanes 14 and 15}, plain form {tap 0, tap 1, tap 2}; two channel halves each
  struct Row { uint4 v[3][2]; };
  auto load_one = [&](const Coord& c, int t, uint4 (&dst)[2]) {
    const bool ok = (c.ok >> t) & 1u;
    const bf16_t* p = ok ? c.frame + (c.off00 + dtap[t]) : zero;
    dst[0] = *reinterpret_cast<const uint4*>(p);
    dst[1] = *reinterpret_cast<const uint4*>(p + (ok ? 32 : 0));
  };
  auto load_row = [&](const Coord& c, int kr, Row& row) {
    load_one(c, 3 * kr, row.v[0]);
    if (c.wrap) {                       // wave-uniform
      load_one(c, 3 * kr + 1, row.v[1]);
      load_one(c, 3 * kr + 2, row.v[2]);
    } else {
      if (STRIDE == 2) load_one(c, 3 * kr + 1, row.v[1]);
      if (r >= (STRIDE == 1 ? 14 : 15)) load_one(c, 3 * kr + 2, row.v[2]);
    }
  };
  auto shl = [](unsigned old, unsigned src, auto ctrl) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)src, decltype(ctrl)::value, 0xf, 0xf, false);
  };
  // the three taps of a filter row from its raw registers
  auto taps_of = [&](const Row& row, bool wrap, uint4 (&tap)[3][2]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      tap[0][h] = row.v[0][h];
      if (wrap) {
        tap[1][h] = row.v[1][h];
        tap[2][h] = row.v[2][h];
      } else {
        const unsigned l0[4] = {row.v[0][h].x, row.v[0][h].y, row.v[0][h].z, row.v[0][h].w};
        const unsigned xx[4] = {row.v[2][h].x, row.v[2][h].y, row.v[2][h].z, row.v[2][h].w};
        unsigned t1[4], t2[4];
        if constexpr (STRIDE == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // lane 15 of tap 1: lane 14's extra pixel; lanes 14, 15 of tap 2: their own extra pixels
            const unsigned xs = shl(0u, xx[e], std::integral_constant<int, 0x111>{});     // row_shr:1
            t1[e] = shl(xs, l0[e], std::integral_constant<int, 0x101>{});                // row_shl:1
            t2[e] = shl(xx[e], l0[e], std::integral_constant<int, 0x102>{});             // row_shl:2
          }
          tap[1][h] = make_uint4(t1[0], t1[1], t1[2], t1[3]);
        } else {
          // lane 15 of tap 2: its own extra pixel
#pragma unroll
          for (int e = 0; e < 4; ++e) t2[e] = shl(xx[e], l0[e], std::integral_constant<int, 0x101>{});
          tap[1][h] = row.v[1][h];
        }
        tap[2][h] = make_uint4(t2[0], t2[1], t2[2], t2[3]);
      }
    }
  };

  const long long step = gridDim.x;
  long long pass = blockIdx.x;
  if (pass >= passes) return;
  Row ring[2][2];                       // [slot][group]
  Coord c0 = coord(pass, 0), c1 = coord(pass, 1);
  load_row(c0, 0, ring[0][0]);
  load_row(c1, 0, ring[0][1]);
  f32x4 acc[NTILE][2];
  // one filter row of the current pass from ring slot `slot`, after which the slot is refilled
  // with row `kr + 2` (of the next pass for kr >= 1)
  auto do_row = [&](auto slot_c, int kr, const Coord& cc0, const Coord& cc1, const Coord& nn0, const Coord& nn1) {
    constexpr int slot = decltype(slot_c)::value;
    // the row after this one goes into the other slot first, so that it is in flight under
    // this row's MFMAs
    if (kr + 1 < 3) {
      load_row(cc0, kr + 1, ring[slot ^ 1][0]);
      load_row(cc1, kr + 1, ring[slot ^ 1][1]);
    } else {
      load_row(nn0, 0, ring[slot ^ 1][0]);
      load_row(nn1, 0, ring[slot ^ 1][1]);
    }
    uint4 tap[2][3][2];
    taps_of(ring[slot][0], cc0.wrap, tap[0]);
    taps_of(ring[slot][1], cc1.wrap, tap[1]);
    // The six k steps of the row (3 taps x 2 channel halves).  The weight fragments of step
    // i + 1 are read while the MFMAs of step i run (two register sets), except with 128 filters
    // at stride 1, where no registers are left for that.
    constexpr bool WPIPE = NTILE <= 4 || STRIDE == 2;
    uint4 wq[2][NTILE];
    auto read_w = [&](int i, uint4 (&dst)[NTILE]) {
#pragma unroll
      for (int j = 0; j < NTILE; ++j)
        dst[j] = *reinterpret_cast<const uint4*>(wfrag + (j * kC3Steps + 6 * kr + i) * 1024);
    };
    if constexpr (WPIPE) {
      read_w(0, wq[0]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if constexpr (WPIPE) {
        if (i + 1 < 6) read_w(i + 1, wq[(i + 1) & 1]);
      } else {
        read_w(i, wq[0]);
      }
#pragma unroll
      for (int j = 0; j < NTILE; ++j)
#pragma unroll
        for (int grp = 0; grp < 2; ++grp)
          acc[j][grp] = mfma_16x16x32<H>(wq[WPIPE ? (i & 1) : 0][j], tap[grp][i >> 1][i & 1], acc[j][grp]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto begin_pass = [&]() {
#pragma unroll
    for (int j = 0; j < NTILE; ++j) {
      const float4 b = *reinterpret_cast<const float4*>(bl + 32 * (j >> 1) + 8 * g + 4 * (j & 1));
      acc[j][0] = acc[j][1] = f32x4{b.x, b.y, b.z, b.w};
    }
  };
  auto end_pass = [&](const Coord& cc0, const Coord& cc1) {
#pragma unroll
    for (int grp = 0; grp < 2; ++grp) {
      bf16_t* py = a.y + (size_t)(grp == 0 ? cc0.m : cc1.m) * a.y_ld + 8 * g;
#pragma unroll
      for (int u = 0; u < C1P; ++u) {
        float v[8] = {acc[2 * u][grp][0], acc[2 * u][grp][1], acc[2 * u][grp][2], acc[2 * u][grp][3],
                      acc[2 * u + 1][grp][0], acc[2 * u + 1][grp][1], acc[2 * u + 1][grp][2], acc[2 * u + 1][grp][3]};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], lo, hi);
        store16_nt(py + 32 * u, Half16<H>::pack2(v[0], v[1]), Half16<H>::pack2(v[2], v[3]),
                   Half16<H>::pack2(v[4], v[5]), Half16<H>::pack2(v[6], v[7]));
      }
    }
  };
  typedef std::integral_constant<int, 0> S0;
  typedef std::integral_constant<int, 1> S1;
  while (true) {
    // pass A: rows 0, 1, 2 from slots 0, 1, 0
    Coord n0 = coord(pass + step, 0), n1 = coord(pass + step, 1);
    begin_pass();
    do_row(S0{}, 0, c0, c1, n0, n1);
    do_row(S1{}, 1, c0, c1, n0, n1);
    do_row(S0{}, 2, c0, c1, n0, n1);
    end_pass(c0, c1);
    pass += step;
    if (pass >= passes) break;
    // pass B: rows 0, 1, 2 from slots 1, 0, 1
    c0 = n0; c1 = n1;
    n0 = coord(pass + step, 0); n1 = coord(pass + step, 1);
    begin_pass();
    do_row(S1{}, 0, c0, c1, n0, n1);
    do_row(S0{}, 1, c0, c1, n0, n1);
    do_row(S1{}, 2, c0, c1, n0, n1);
    end_pass(c0, c1);
    pass += step;
    if (pass >= passes) break;
    c0 = n0; c1 = n1;
  }
}

// SHARED: 0 = every tap loaded, 1 / 2 = taps shared between lanes at that stride
template <typename H, int C1P, int SHARED>
static int launch_conv3(const Conv3Args& a, hipStream_t stream) {
  constexpr size_t smem = (size_t)2 * C1P * kC3Steps * 1024 + 32 * C1P * 4;
  static_assert(smem <= 160 * 1024, "filter too large for LDS");
  auto kern = SHARED == 1 ? conv3x3_c64_shared_kernel<H, C1P, 1>
              : SHARED == 2 ? conv3x3_c64_shared_kernel<H, C1P, 2> : conv3x3_c64_kernel<H, C1P>;
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
  // persistent blocks: as many as are resident on the chip at once, at most one per pass
  static int per_cu = 0, num_cu = 0;
  if (num_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      set_error("fss_conv3x3_c64_nhwc: cannot query the device");
      return 1;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), kC3Waves * 64,
                                                     smem) != hipSuccess || per_cu < 1)
      per_cu = 1;
    num_cu = prop.multiProcessorCount;
  }
  const long long passes = (a.M + kC3Pixels - 1) / kC3Pixels;
  const int blocks = (int)std::min<long long>(passes, (long long)num_cu * per_cu);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kC3Waves * 64), smem, stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error(std::string("fss_conv3x3_c64_nhwc launch: ") + hipGetErrorString(e));
    return 3;
  }
  return 0;
}

}  // namespace fss

using namespace fss;

extern "C" int fss_conv3x3_c64_nhwc(const void* x, int x_ld, const void* w, const float* bias, void* y,
                                    int y_ld, int n, int h, int wd, int ho, int wo, int cout, int stride,
                                    int dilation, int pad_top, int pad_left, int act, int dtype,
                                    void* stream) {
  FSS_REQUIRE(x && w && y, "null argument");
  FSS_REQUIRE(dtype == FSS_BF16 || dtype == FSS_F16, "only bfloat16 and half precision are supported");
  FSS_REQUIRE(n > 0 && h > 0 && wd > 0 && ho > 0 && wo > 0, "bad shape");
  FSS_REQUIRE(cout == 64 || cout == 128, "cout must be 64 or 128");
  FSS_REQUIRE(stride >= 1 && stride <= 4 && dilation >= 1 && dilation <= 8, "bad stride / dilation");
  FSS_REQUIRE(pad_top >= 0 && pad_left >= 0, "negative padding");
  FSS_REQUIRE((long long)(ho - 1) * stride - pad_top < h && (long long)(wo - 1) * stride - pad_left < wd,
              "output size exceeds the input");
  FSS_REQUIRE(x_ld >= 64 && x_ld % 8 == 0 && y_ld >= cout && y_ld % 8 == 0,
              "pixel strides must be multiples of 8 and at least the channel count");
  FSS_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)y & 15) == 0, "unaligned pointer");
  FSS_REQUIRE(act == FSS_ACT_NONE || act == FSS_ACT_RELU || act == FSS_ACT_RELU6, "bad act");
  FSS_REQUIRE((long long)(h + 2 * dilation + pad_top) * wd * x_ld < 0x7fffffffLL && (long long)ho * wo < 0x7fffffffLL,
              "image too large (element offsets inside one frame are 32-bit)");
  Conv3Args a;
  a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.bias = bias; a.y = (bf16_t*)y;
  a.M = (long long)n * ho * wo;
  a.N = n; a.H = h; a.W = wd; a.Ho = ho; a.Wo = wo; a.stride = stride; a.dil = dilation;
  a.pad_t = pad_top; a.pad_l = pad_left; a.x_ld = x_ld; a.y_ld = y_ld; a.act = act;
  hipStream_t s = (hipStream_t)stream;
  // stride 1, dilation 1: horizontal taps shared between lanes
  const int shared = dilation == 1 && stride <= 2 && getenv("FSS_C3_PLAIN") == nullptr ? stride : 0;
  if (dtype == FSS_F16) {
    if (shared == 1) return cout == 64 ? launch_conv3<f16_t, 2, 1>(a, s) : launch_conv3<f16_t, 4, 1>(a, s);
    if (shared == 2) return cout == 64 ? launch_conv3<f16_t, 2, 2>(a, s) : launch_conv3<f16_t, 4, 2>(a, s);
    return cout == 64 ? launch_conv3<f16_t, 2, 0>(a, s) : launch_conv3<f16_t, 4, 0>(a, s);
  }
  if (shared == 1) return cout == 64 ? launch_conv3<bf16_t, 2, 1>(a, s) : launch_conv3<bf16_t, 4, 1>(a, s);
  if (shared == 2) return cout == 64 ? launch_conv3<bf16_t, 2, 2>(a, s) : launch_conv3<bf16_t, 4, 2>(a, s);
  return cout == 64 ? launch_conv3<bf16_t, 2, 0>(a, s) : launch_conv3<bf16_t, 4, 0>(a, s);
}
