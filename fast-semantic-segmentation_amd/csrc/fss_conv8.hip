// 256x256-tile, 8-wave, 8-phase form of the bf16 implicit-GEMM convolution for gfx950.
//
// Same mathematics, operand gathering, LDS byte layout (128-byte rows, XOR-swizzled 16-byte
// chunks), MFMA shape (v_mfma_f32_16x16x32_bf16, k ascending) and epilogue as conv_igemm_kernel
// in fss_conv.hip - results are bit-identical to its 16x16x32 configurations - but another
// schedule of the K loop:
//
//   * 8 waves as 2 (rows) x 4 (columns), each owning a 128x64 piece of the output (four 64x32
//     quadrants);
//   * a K tile (64 channels of one filter tap) lives in LDS as four half-tile images of 128 rows
//     x 128 bytes, cut by what a PHASE reads: j = 0: the weight rows of quadrant column 0 of all
//     four wave columns, j = 1: the pixel rows of quadrant row 0 of both wave rows, j = 2: weight
//     rows of quadrant column 1, j = 3: pixel rows of quadrant row 1.  Two K tiles of buffers
//     (128 KiB);
//   * a K tile is four phases, one per quadrant, in the order (0,0) (0,1) (1,1) (1,0).  A phase
//     reads the fragments it is the first to need (12 / 4 / 8 / 0 ds_read_b128), issues the
//     LDS-DMA of ONE half-tile (two pieces per lane) seven half-tiles ahead of the one it reads,
//     passes a barrier, runs its 16 MFMAs and passes a second barrier.  The DMA is waited for
//     once per K tile (vmcnt(6) in the fourth phase: everything but the three newest half-tiles
//     has landed, i.e. all of the next K tile), a phase before the first read of that data;
//   * every buffer is refilled two phases after its last read, except j = 0, refilled one phase
//     after: its four reads are issued first in their phase and retired (lgkmcnt) before that
//     phase's first barrier;
//   * the two wave rows run one barrier apart (the second wave row passes one extra barrier
//     before the loop, the first one after it), so that on every SIMD one wave issues MFMAs
//     while the other reads fragments and issues DMA.
#include <type_traits>

#include "fss_conv_args.h"

namespace fss {
namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

// 16 zero bytes every padding tap is fetched from.
__device__ __attribute__((aligned(16))) unsigned int g_zero_chunk8[4] = {0u, 0u, 0u, 0u};

constexpr int kBM = 256, kBN = 256, kNT = 512;
constexpr int kRowBytes = 128, kBK = 64;          // one tile row: 64 bf16 of K
constexpr int kHalf = 128 * kRowBytes;            // one half-tile image
constexpr int kLdsBytes = 8 * kHalf;              // two K tiles x four half-tiles

__device__ __forceinline__ void mma(f32x4_t& acc, const uint4& a, const uint4& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}

template <int V> using ic = std::integral_constant<int, V>;

template <typename TO>
__global__ void __launch_bounds__(kNT)
conv8_kernel(const ConvArgs a) {
  typedef bf16_t T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  // XCD-aware tile order, as in conv_igemm_kernel
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, rr = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < rr ? xcd * (q8 + 1) : rr * (q8 + 1) + (xcd - rr) * q8) + (bid >> 3);
  const int tile_n = wg % a.tiles_n, tile_m = wg / a.tiles_n;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wr = wave >> 2, wc = wave & 3;
  const int prow = tid >> 3;                            // image row of this lane's first piece
  const int chunk = (tid & 7) ^ ((prow >> 1) & 7);      // logical chunk it fetches (rows +64: same)

  const int HoWo = a.Ho * a.Wo;
  const int ntap = a.KH * a.KW;

  const bool pointwise = a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad_t == 0 && a.pad_l == 0 &&
                         a.Ho == a.H && a.Wo == a.W;
  // ---- DMA windows of this block (buffer resources; the offsets below are relative to them)
  const int m_first = tile_m * kBM;
  const long long pix_first = pointwise ? (long long)m_first : (long long)(m_first / HoWo) * a.H * a.W;
  const unsigned long long x_bytes = (unsigned long long)a.N * a.H * a.W * a.Cin * 2;
  const unsigned long long x_first = (unsigned long long)pix_first * a.Cin * 2;
  const fss_i32x4 rs_x = dma_resource(reinterpret_cast<const char*>(a.x) + x_first, x_bytes - x_first);
  const unsigned long long x2_first = (unsigned long long)m_first * a.Cin2 * 2;
  const fss_i32x4 rs_x2 = dma_resource(
      reinterpret_cast<const char*>(a.x2 != nullptr ? a.x2 : a.x) + x2_first,
      (unsigned long long)a.M * a.Cin2 * 2 - x2_first);
  const fss_i32x4 rs_w = dma_resource(
      reinterpret_cast<const char*>(a.w) + (size_t)tile_n * kBN * a.Kpad * 2, (size_t)kBN * a.Kpad * 2);
  // ---- gather state of this lane's four pixel rows: image row prow + 64 i of half-tile qm is
  // tile row 128 i + 64 qm + prow (i = the wave row that reads it); roff = byte offset of the
  // lane's chunk at tap (0, 0) in the input window, modulo 2^32
  unsigned roff[2][2];
  unsigned tapmask[2][2];
  unsigned r2off[2][2];
#pragma unroll
  for (int qm = 0; qm < 2; ++qm)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int trow = i * 128 + qm * 64 + prow;
      const int m = m_first + trow;
      const bool row_ok = m < a.M;
      int n = 0, oh = 0, ow = 0, ih0, iw0;
      long long pix;
      if (pointwise) {
        ih0 = iw0 = row_ok ? 0 : -(1 << 28);
        pix = row_ok ? m : 0;
      } else {
        if (row_ok) {
          n = m / HoWo;
          const int rem = m - n * HoWo;
          oh = rem / a.Wo;
          ow = rem - oh * a.Wo;
        }
        ih0 = row_ok ? oh * a.stride - a.pad_t : -(1 << 28);
        iw0 = row_ok ? ow * a.stride - a.pad_l : -(1 << 28);
        pix = ((long long)n * a.H + (row_ok ? ih0 : 0)) * a.W + (row_ok ? iw0 : 0);
      }
      roff[qm][i] = (unsigned)(((pix - pix_first) * a.Cin + chunk * 8) * 2);
      unsigned mask = 0;
      for (int r = 0, t = 0; r < a.KH; ++r)
        for (int sx = 0; sx < a.KW; ++sx, ++t) {
          const bool ok = ((unsigned)(ih0 + r * a.dil) < (unsigned)a.H) &&
                          ((unsigned)(iw0 + sx * a.dil) < (unsigned)a.W);
          mask |= (unsigned)ok << t;
        }
      if (a.Cin2 > 0) mask |= (unsigned)row_ok << ntap;   // the second input: "one more tap"
      tapmask[qm][i] = mask;
      r2off[qm][i] = (unsigned)((trow * a.Cin2 + chunk * 8) * 2);
    }
  // weight rows: image row prow + 64 i of half-tile qn is channel 64 wcol + 32 qn + (prow & 31)
  // of the N tile, wcol = (prow >> 5) + 2 i the wave column that reads it
  unsigned woff[2][2];
#pragma unroll
  for (int qn = 0; qn < 2; ++qn)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int nrow = ((prow >> 5) + 2 * i) * 64 + qn * 32 + (prow & 31);
      woff[qn][i] = (unsigned)(((size_t)nrow * a.Kpad + chunk * 8) * 2);
    }

  // ---- taps some row of the block reads for real; the others are skipped
  __shared__ unsigned wave_taps[8];
  const int ntaps_all = ntap + (a.Cin2 > 0 ? 1 : 0);
  unsigned bmask = 0;
  {
    const unsigned mine = tapmask[0][0] | tapmask[0][1] | tapmask[1][0] | tapmask[1][1];
    unsigned wmask = 0;
    for (int t = 0; t < ntaps_all; ++t)
      if (__ballot((mine >> t) & 1u) != 0ull) wmask |= 1u << t;
    if (lane == 0) wave_taps[wave] = wmask;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 8; ++w) bmask |= wave_taps[w];
    bmask = (unsigned)__builtin_amdgcn_readfirstlane((int)bmask);
  }
  const int KT = __popc(bmask & ((1u << ntap) - 1u)) * (a.Cin / kBK) + a.Cin2 / kBK;
  // tap walk of the K tile being STAGED (uniform): channel base kc of tap (kr, ks)
  int kc = 0, kr = 0, ks = 0, tap = 0, skt = 0;
  while (tap < ntaps_all && !((bmask >> tap) & 1u)) {
    ++tap;
    if (++ks == a.KW) { ks = 0; ++kr; }
  }
  auto advance = [&]() {
    kc += kBK;
    if (kc == (tap == ntap ? a.Cin2 : a.Cin)) {
      kc = 0;
      do {
        ++tap;
        if (++ks == a.KW) { ks = 0; ++kr; }
      } while (tap < ntaps_all && !((bmask >> tap) & 1u));
    }
    ++skt;
  };
  // issues the two DMA pieces of half-tile J of the K tile being staged into the buffers at
  // byte offset dofs; K tiles past the last one, padding taps and rows past M deposit zeros
  // (out-of-range offsets), so every wave keeps the same number of pieces in flight
  auto stage = [&](auto jtag, unsigned dofs) {
    constexpr int J = decltype(jtag)::value;
    const unsigned dst = smem_base + dofs + J * kHalf + wave * 1024;
    const bool in_range = skt < KT;
    if constexpr (J == 1 || J == 3) {
      constexpr int qm = J == 3 ? 1 : 0;
      const int tbit = min(tap, 31);
      const int tap_off = ((kr * a.dil * a.W + ks * a.dil) * a.Cin + kc) * 2;
      const bool second = a.Cin2 > 0 && tap == ntap;
      const fss_i32x4 rs = second ? rs_x2 : rs_x;                // uniform
      const unsigned add = second ? (unsigned)(kc * 2) : (unsigned)tap_off;
      const unsigned bit = in_range && tap < 31 ? 1u << tbit : 0u;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const unsigned base = second ? r2off[qm][i] : roff[qm][i];
        blds16((tapmask[qm][i] & bit) ? base + add : kDmaOutOfRange, rs, dst + i * 8192);
      }
    } else {
      constexpr int qn = J == 2 ? 1 : 0;
      const unsigned wk = (unsigned)(tap * a.Cin + kc) * 2u;
#pragma unroll
      for (int i = 0; i < 2; ++i) blds16(in_range ? woff[qn][i] + wk : kDmaOutOfRange, rs_w, dst + i * 8192);
    }
    if constexpr (J == 3) advance();
  };

  // ---- fragment addresses: lane (lrow, lg) reads row lrow of a 16-row fragment, 16 bytes of
  // k-group lg; logical chunk 4 s + lg of k step s
  const int lrow = lane & 15, lg = lane >> 4;
  const int swz = (lrow >> 1) & 7;
  const int foff0 = ((lg) ^ swz) * 16, foff1 = ((4 + lg) ^ swz) * 16;
  const char* pA = smem + (wr * 64 + lrow) * kRowBytes;
  const char* pB = smem + (wc * 32 + lrow) * kRowBytes;

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: K tile 0 and the first three half-tiles of K tile 1
  stage(ic<0>{}, 0u); stage(ic<1>{}, 0u); stage(ic<2>{}, 0u); stage(ic<3>{}, 0u);
  stage(ic<0>{}, 4u * kHalf); stage(ic<1>{}, 4u * kHalf); stage(ic<2>{}, 4u * kHalf);
  wait_vmcnt<6>();                                   // K tile 0 has landed (this wave's pieces)
  __builtin_amdgcn_s_barrier();                      // ... and everybody else's
  if (wr == 1) __builtin_amdgcn_s_barrier();         // the second wave row runs one barrier behind

  auto ktile = [&](auto dtag) {
    constexpr unsigned D = decltype(dtag)::value * 4u * kHalf;          // this K tile's buffers
    constexpr unsigned DN = (1 - decltype(dtag)::value) * 4u * kHalf;   // the next K tile's
    uint4 fa[4][2], fb0[2][2], fb1[2][2];
    // ---------------- phase 0: quadrant (0, 0)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      fb0[ni][0] = *reinterpret_cast<const uint4*>(pB + D + 0 * kHalf + ni * 16 * kRowBytes + foff0);
      fb0[ni][1] = *reinterpret_cast<const uint4*>(pB + D + 0 * kHalf + ni * 16 * kRowBytes + foff1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      fa[mi][0] = *reinterpret_cast<const uint4*>(pA + D + 1 * kHalf + mi * 16 * kRowBytes + foff0);
      fa[mi][1] = *reinterpret_cast<const uint4*>(pA + D + 1 * kHalf + mi * 16 * kRowBytes + foff1);
    }
    stage(ic<3>{}, DN);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");   // the four weight reads have returned
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) mma(acc[mi][ni], fa[mi][s], fb0[ni][s]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---------------- phase 1: quadrant (0, 1)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      fb1[ni][0] = *reinterpret_cast<const uint4*>(pB + D + 2 * kHalf + ni * 16 * kRowBytes + foff0);
      fb1[ni][1] = *reinterpret_cast<const uint4*>(pB + D + 2 * kHalf + ni * 16 * kRowBytes + foff1);
    }
    stage(ic<0>{}, D);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) mma(acc[mi][2 + ni], fa[mi][s], fb1[ni][s]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---------------- phase 2: quadrant (1, 1)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      fa[mi][0] = *reinterpret_cast<const uint4*>(pA + D + 3 * kHalf + mi * 16 * kRowBytes + foff0);
      fa[mi][1] = *reinterpret_cast<const uint4*>(pA + D + 3 * kHalf + mi * 16 * kRowBytes + foff1);
    }
    stage(ic<1>{}, D);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) mma(acc[4 + mi][2 + ni], fa[mi][s], fb1[ni][s]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---------------- phase 3: quadrant (1, 0); the next K tile has landed after this wait
    stage(ic<2>{}, D);
    wait_vmcnt<6>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) mma(acc[4 + mi][ni], fa[mi][s], fb0[ni][s]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };

  int kt = 0;
  for (; kt + 1 < KT; kt += 2) {
    ktile(ic<0>{});
    ktile(ic<1>{});
  }
  if (kt < KT) ktile(ic<0>{});
  if (wr == 0) __builtin_amdgcn_s_barrier();         // the first wave row catches up

  // ---- epilogue (as conv_igemm_kernel: 64x64-wide staging through LDS, 16-byte stores)
  constexpr int WTM = 128, WTN = 64, FM = 16, MI = 8, NI = 4, NREG = 4;
  constexpr int SROW = WTN + 4;
  constexpr int SLABR = 32;
  static_assert(8 * SLABR * SROW * 4 <= kLdsBytes, "epilogue staging does not fit");
  constexpr int VW = 16 / (int)sizeof(TO);
  constexpr int QPR = WTN / VW;
  constexpr int PASSES = SLABR * QPR / 64;
  constexpr int SLABS = WTM / SLABR;
  constexpr int TPS = SLABR / FM;
  static_assert(MI == SLABS * TPS, "slabs must cover the wave tile");
  TO* __restrict__ yg = reinterpret_cast<TO*>(a.y);
  const T* __restrict__ rg = reinterpret_cast<const T*>(a.res);
  const int wm = wr, wn = wc;
  const int n_base = tile_n * kBN + wn * WTN;
  const bool vec_ok = ((a.y_ld & (VW - 1)) == 0) && ((a.Cout & (VW - 1)) == 0) &&
                      ((reinterpret_cast<uintptr_t>(a.y) & 15) == 0) &&
                      (rg == nullptr ||
                       (((a.res_ld & (VW - 1)) == 0) && ((reinterpret_cast<uintptr_t>(a.res) & 15) == 0)));
  typedef typename std::conditional<VW == 8, uint4, uint2>::type rvec_t;
  rvec_t rpre[SLABS][PASSES];
  bool drained = false;
  if (rg != nullptr && vec_ok) {
#pragma unroll
    for (int sl = 0; sl < SLABS; ++sl)
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        const int gq = lane + p * 64;
        const int m = tile_m * kBM + wm * WTM + sl * SLABR + gq / QPR, n = n_base + (gq % QPR) * VW;
        const bool ok = m < a.M && n < a.Cout;
        const T* rp = ok ? rg + (size_t)m * a.res_ld + n : reinterpret_cast<const T*>(g_zero_chunk8);
        rpre[sl][p] = *reinterpret_cast<const rvec_t*>(rp);
      }
    wait_vmcnt<SLABS * PASSES>();
    drained = true;
  } else {
#pragma unroll
    for (int sl = 0; sl < SLABS; ++sl)
#pragma unroll
      for (int p = 0; p < PASSES; ++p) rpre[sl][p] = rvec_t{};
  }
  if (!drained) wait_vmcnt<0>();
  __syncthreads();

  float* stage_f = reinterpret_cast<float*>(smem) + wave * (SLABR * SROW);
  const float act_lo = a.act == FSS_ACT_NONE ? -INFINITY : 0.f;
  const float act_hi = a.act == FSS_ACT_RELU6 ? 6.f : INFINITY;
#pragma unroll
  for (int sl = 0; sl < SLABS; ++sl) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n_base + j * FM + lrow;
      const float bias = (a.bias != nullptr && n < a.Cout) ? a.bias[n] : 0.f;
#pragma unroll
      for (int ti = 0; ti < TPS; ++ti)
#pragma unroll
        for (int r = 0; r < NREG; ++r)
          stage_f[(ti * FM + 4 * lg + r) * SROW + j * FM + lrow] = acc[sl * TPS + ti][j][r] + bias;
    }
    wave_lds_fence();   // the slab is private to this wave: LDS order within a wave suffices
    const int m_base = tile_m * kBM + wm * WTM + sl * SLABR;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int g = lane + p * 64;
      const int row = g / QPR, cv = (g % QPR) * VW;
      const int m = m_base + row, n = n_base + cv;
      if (m < a.M && n < a.Cout) {
        float v[VW];
#pragma unroll
        for (int q = 0; q < VW; q += 4) {
          const float4 v4 = *reinterpret_cast<const float4*>(stage_f + row * SROW + cv + q);
          v[q] = v4.x; v[q + 1] = v4.y; v[q + 2] = v4.z; v[q + 3] = v4.w;
        }
        TO* dst = yg + (size_t)m * a.y_ld + n;
        if (vec_ok) {
          if (rg != nullptr) {
            const rvec_t rv = rpre[sl][p];
            const unsigned* rw = reinterpret_cast<const unsigned*>(&rv);
#pragma unroll
            for (int q = 0; q < VW / 2; ++q) {
              v[2 * q] += __uint_as_float(rw[q] << 16);
              v[2 * q + 1] += __uint_as_float(rw[q] & 0xffff0000u);
            }
          }
#pragma unroll
          for (int e = 0; e < VW; ++e) v[e] = fminf(fmaxf(v[e], act_lo), act_hi);
          if constexpr (sizeof(TO) == 4) {
            *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
            store16_nt(dst, pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                       pack_bf16x2(v[6], v[7]));
          }
        } else {
#pragma unroll
          for (int e = 0; e < VW; ++e) {
            if (n + e < a.Cout) {
              float t = v[e];
              if (rg != nullptr) t += ElemTraits<T>::load(rg + (size_t)m * a.res_ld + n + e);
              ElemTraits<TO>::store(dst + e, fminf(fmaxf(t, act_lo), act_hi));
            }
          }
        }
      }
    }
    wave_lds_fence();   // the slab is private to this wave: LDS order within a wave suffices
  }
}

}  // namespace

int launch_conv8_bf16(ConvArgs& a, hipStream_t stream) {
  // covers: bf16 in and out, cin (and cin2) in whole K tiles, at most 30 taps, no split-K
  if (a.Cin % kBK != 0 || a.Cin2 % kBK != 0 || a.KH * a.KW > 30 || a.splits != 1) return -1;
  a.tiles_m = ceil_div(a.M, kBM);
  a.tiles_n = ceil_div(a.Cout, kBN);
  const long long blocks = (long long)a.tiles_m * a.tiles_n;
  if (blocks <= 0 || blocks > 0x7fffffffLL) {
    set_error("fss_conv2d_nhwc: grid out of range");
    return 2;
  }
  auto kern = conv8_kernel<bf16_t>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    if (e != hipSuccess) {
      set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
      return 1;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kNT), kLdsBytes, stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error(std::string("conv8 launch: ") + hipGetErrorString(e));
    return 3;
  }
  return 0;
}

}  // namespace fss
