// The implicit-GEMM convolution kernel template, its tile configurations and launchers.
// Included by the translation units that instantiate it for their tensor types
// (fss_conv.hip: float, bfloat16, split-bf16; fss_conv_f16.hip: half precision).
#pragma once
#include <type_traits>

#include "fss_conv_args.h"

namespace fss {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;

// 16 zero bytes every padding tap is fetched from.
static __device__ __attribute__((aligned(16))) unsigned int g_zero_chunk[4] = {0u, 0u, 0u, 0u};

typedef __attribute__((ext_vector_type(4))) float f32x4;

// MFMA shapes.  FM = edge of the square output tile of one instruction.
//   FM = 32: lane (r = lane&31, g = lane>>5) supplies row r, 16 bytes of k-group g (2 groups);
//            accumulator reg q holds C[(q&3) + 8*(q>>2) + 4*g][r].
//   FM = 16: lane (r = lane&15, g = lane>>4) supplies row r, 16 bytes of k-group g (4 groups);
//            accumulator reg q holds C[4*g + q][r].
// A float chunk is 4 k values, consumed by 4 f32 MFMAs with the same k permutation
// on both operands.
// Split-bfloat16 mode (tensors of f32s_t slots, see fss_common.h): each product a*b is formed on
// the bf16 matrix pipe as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi, which keeps about 16 mantissa bits
// per operand: float-class results at three bf16 MFMAs per step instead of the 16x slower float
// MFMAs.  Activations and weights both arrive split (per 8 k values 8 bf16 high parts, then 8
// bf16 remainders), so the main loop is the bf16 loop with twice the bytes per k and three
// MFMAs per fragment pair; the epilogue splits each output value once.
template <typename T, int FM> struct Mma;
template <> struct Mma<bf16_t, 32> {
  typedef f32x16 acc_t;
  static constexpr int NREG = 16;
  static constexpr bool PAIRED = false;
  static __device__ __forceinline__ int row_of(int q, int g) { return (q & 3) + 8 * (q >> 2) + 4 * g; }
  static __device__ __forceinline__ void run(acc_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                  __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
  }
};
template <> struct Mma<bf16_t, 16> {
  typedef f32x4 acc_t;
  static constexpr int NREG = 4;
  static constexpr bool PAIRED = false;
  static __device__ __forceinline__ int row_of(int q, int g) { return 4 * g + q; }
  static __device__ __forceinline__ void run(acc_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                  __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
  }
};
// Half precision: the same tile geometry and LDS byte layout as bfloat16.
template <> struct Mma<f16_t, 32> : Mma<bf16_t, 32> {
  static __device__ __forceinline__ void run(acc_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a),
                                                 __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
  }
};
template <> struct Mma<f16_t, 16> : Mma<bf16_t, 16> {
  static __device__ __forceinline__ void run(acc_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a),
                                                 __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
  }
};
// half-precision activations against two-f16 weights: the f16 MFMA, once per weight part
template <> struct Mma<f16w_t, 16> : Mma<f16_t, 16> {};
template <> struct Mma<float, 32> {
  typedef f32x16 acc_t;
  static constexpr int NREG = 16;
  static constexpr bool PAIRED = false;
  static __device__ __forceinline__ int row_of(int q, int g) { return (q & 3) + 8 * (q >> 2) + 4 * g; }
  static __device__ __forceinline__ void run(acc_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  }
};
template <> struct Mma<float, 16> {
  typedef f32x4 acc_t;
  static constexpr int NREG = 4;
  static constexpr bool PAIRED = false;
  static __device__ __forceinline__ int row_of(int q, int g) { return 4 * g + q; }
  static __device__ __forceinline__ void run(acc_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  }
};

// Split mode: the float chunks of two consecutive k steps (8 values per lane and operand, the
// same k positions on both operands) feed one bf16 MFMA step per product term; the two small
// terms are added first.
template <> struct Mma<f32s_t, 32> : Mma<bf16_t, 32> {
  static constexpr bool PAIRED = true;
  static __device__ __forceinline__ void run3(acc_t& acc, const uint4& ahi, const uint4& alo,
                                              const uint4& bhi, const uint4& blo) {
    Mma<bf16_t, 32>::run(acc, alo, bhi);
    Mma<bf16_t, 32>::run(acc, ahi, blo);
    Mma<bf16_t, 32>::run(acc, ahi, bhi);
  }
};
template <> struct Mma<f32s_t, 16> : Mma<bf16_t, 16> {
  static constexpr bool PAIRED = true;
  static __device__ __forceinline__ void run3(acc_t& acc, const uint4& ahi, const uint4& alo,
                                              const uint4& bhi, const uint4& blo) {
    Mma<bf16_t, 16>::run(acc, alo, bhi);
    Mma<bf16_t, 16>::run(acc, ahi, blo);
    Mma<bf16_t, 16>::run(acc, ahi, bhi);
  }
};

// ALIGNED: cin is a multiple of the K tile, so a tile never straddles two filter
// taps and the tap walk is wave-uniform (scalar) instead of per lane.
// scores per pixel the classifier-head epilogue can produce (row stride of its staging)
constexpr int kHeadMax = 20;

template <typename T, typename TO, int BM, int BN, int WAVES_M, int WAVES_N, int BKB, int NSTAGE,
          int FM, bool ALIGNED, bool HEAD = false>
__global__ void __launch_bounds__(WAVES_M* WAVES_N * 64)
conv_igemm_kernel(const ConvArgs a) {
  typedef Mma<T, FM> MMA;
  constexpr int LG = 64 / FM;               // lane groups = 16-byte k-groups per MFMA step
  constexpr int KSTEPS = BKB / (16 * LG);   // MFMA steps per K tile
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int EPC = 16 / (int)sizeof(T);  // elements per 16-byte chunk
  constexpr int CPR = BKB / 16;             // chunks per tile row (4 or 8)
  constexpr int BK = CPR * EPC;             // K elements per tile
  constexpr int A_CHUNKS = BM * CPR, B_CHUNKS = BN * CPR;
  constexpr int A_IT = A_CHUNKS / NT, B_IT = B_CHUNKS / NT;
  // WIDE: the B operand comes as two tiles per stage, the high parts of the weights and their
  // remainders (k_pad elements further along the packed row); every MFMA step runs once per part
  constexpr bool WIDE = std::is_same<T, f16w_t>::value;
  constexpr int BPL = WIDE ? 2 : 1;
  constexpr int LOADS = A_IT + BPL * B_IT;  // DMA pieces per wave per stage
  constexpr int RSTEP = NT / CPR;           // row step between a thread's chunks
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MI = WTM / FM, NI = WTN / FM;
  constexpr int STAGE_BYTES = (BM + BPL * BN) * BKB;
  // XOR term of the chunk index of a tile row.  ds_read_b128 serves a wave in four fixed
  // groups of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31} and the same +32); with
  // these terms the 16 (row, chunk) pairs of every group fall on 16 different 16-byte
  // bank windows, for both MFMA shapes (32-row fragments: one chunk per group; 16-row
  // fragments: two chunks per group).  Invariant under row += 16.
  auto swizzle_of = [](int row) { return CPR == 4 ? ((0 - (row >> 2)) & 3) : ((row >> 1) & 7); };
  static_assert(CPR == 4 || CPR == 8, "BKB must be 64 or 128");
  static_assert(A_CHUNKS % NT == 0 && B_CHUNKS % NT == 0, "tile must split evenly over threads");
  static_assert(RSTEP % 16 == 0, "row step must keep the swizzle term lane-invariant");
  static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be a multiple of 32x32");
  static_assert(NSTAGE >= 2 && (NSTAGE - 2) * LOADS <= 63, "ring too deep for vmcnt");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned smem_base =
      (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2), so give
  // each XCD a contiguous range of tiles; n-tiles of one m-tile are adjacent.
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7;
  const int wg_all = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
  const int zsplit = wg_all / (a.tiles_m * a.tiles_n);       // 0 unless this is a split-K launch
  const int wg = wg_all - zsplit * (a.tiles_m * a.tiles_n);
  const int tile_n = wg % a.tiles_n, tile_m = wg / a.tiles_n;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int row0 = tid / CPR;
  // this lane fills physical chunk (tid % CPR) of its rows with logical chunk `chunk`
  const int chunk = (tid % CPR) ^ swizzle_of(row0);

  const int HoWo = a.Ho * a.Wo;

  // 1x1, stride 1, unpadded: the A matrix is the input tensor itself
  const bool pointwise = a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad_t == 0 && a.pad_l == 0 &&
                         a.Ho == a.H && a.Wo == a.W;
  // ---- DMA windows of this block (buffer resources; offsets below are relative to them):
  // the input from the first image (pointwise: the first pixel) its rows touch, the second
  // input from its first row, the weights from the first row of its N tile
  const int m_first = tile_m * BM;
  const long long pix_first = pointwise ? (long long)m_first : (long long)(m_first / HoWo) * a.H * a.W;
  const unsigned long long x_bytes = (unsigned long long)a.N * a.H * a.W * a.Cin * sizeof(T);
  const unsigned long long x_first = (unsigned long long)pix_first * a.Cin * sizeof(T);
  const fss_i32x4 rs_x = dma_resource(reinterpret_cast<const char*>(a.x) + x_first, x_bytes - x_first);
  const unsigned long long x2_first = (unsigned long long)m_first * a.Cin2 * sizeof(T);
  const fss_i32x4 rs_x2 = dma_resource(
      reinterpret_cast<const char*>(a.x2 != nullptr ? a.x2 : a.x) + x2_first,
      (unsigned long long)a.M * a.Cin2 * sizeof(T) - x2_first);
  const fss_i32x4 rs_w = dma_resource(
      reinterpret_cast<const char*>(a.w) + (size_t)tile_n * BN * a.Kpad * BPL * sizeof(T),
      (size_t)BN * a.Kpad * BPL * sizeof(T));
  // ---- per-row gather state
  // ALIGNED: roff = byte offset of this lane's chunk at tap (0,0) in the input window (modulo
  //          2^32: exact once a real tap's offset is added), tapmask bit t = tap t of this row
  //          reads a real pixel.  Otherwise: explicit coordinates.
  unsigned roff[A_IT];
  unsigned long long tapmask[A_IT];
  int ih0[A_IT], iw0[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int m = tile_m * BM + row0 + i * RSTEP;
    int n = 0, oh = 0, ow = 0;
    const bool row_ok = m < a.M;
    long long pix;
    if (pointwise) {
      // output pixel m reads input pixel m: no coordinates needed (the tap is always
      // inside the image; (0, 0) stands in for the real position)
      ih0[i] = iw0[i] = row_ok ? 0 : -(1 << 28);
      pix = row_ok ? m : 0;
    } else {
      if (row_ok) {
        n = m / HoWo;
        const int rem = m - n * HoWo;
        oh = rem / a.Wo;
        ow = rem - oh * a.Wo;
      }
      ih0[i] = row_ok ? oh * a.stride - a.pad_t : -(1 << 28);
      iw0[i] = row_ok ? ow * a.stride - a.pad_l : -(1 << 28);
      pix = ((long long)n * a.H + (row_ok ? ih0[i] : 0)) * a.W + (row_ok ? iw0[i] : 0);
    }
    roff[i] = (unsigned)(((pix - pix_first) * a.Cin + chunk * EPC) * (long long)sizeof(T));
    tapmask[i] = 0;
    if constexpr (ALIGNED) {
      for (int r = 0, t = 0; r < a.KH; ++r)
        for (int sx = 0; sx < a.KW; ++sx, ++t) {
          const bool ok = ((unsigned)(ih0[i] + r * a.dil) < (unsigned)a.H) &&
                          ((unsigned)(iw0[i] + sx * a.dil) < (unsigned)a.W);
          tapmask[i] |= (unsigned long long)ok << t;
        }
      // the second input counts as tap number KH*KW: real for every row of the matrix
      if (a.Cin2 > 0) tapmask[i] |= (unsigned long long)row_ok << (a.KH * a.KW);
    }
  }
  // second input: byte offset of this lane's chunk in row row0 of the block's window of x2; the
  // rows of one lane are RSTEP pixels apart
  const unsigned r2off0 = (unsigned)((row0 * a.Cin2 + chunk * EPC) * (int)sizeof(T));
  const unsigned r2step = (unsigned)(RSTEP * a.Cin2 * (int)sizeof(T));
  unsigned woff[B_IT];
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    int wrow = row0 + i * RSTEP;             // row of the B tile = output channel of the N tile
    if constexpr (HEAD) {
      // classifier-head form: tile row 16 j + 4 g + q of a wave's slice holds channel
      // 32 (j >> 1) + 8 g + 4 (j & 1) + q, so that the 8 accumulator values a lane owns in the
      // tile pair (2 s, 2 s + 1) are 8 CONSECUTIVE channels: its residual and classifier
      // operands are then single 16-byte loads
      const int l = wrow % (BN / WAVES_N);
      wrow += (((l >> 5) << 5) + (((l >> 2) & 3) << 3) + (((l >> 4) & 1) << 2) + (l & 3)) - l;
    }
    woff[i] = (unsigned)(((size_t)wrow * a.Kpad * BPL + chunk * EPC) * sizeof(T));
  }

  // tap walk.  ALIGNED: (kc, tap = kr*KW + ks) = channel base / tap of the whole tile
  // (uniform); taps that read padding for EVERY row of this block are skipped, and so
  // is the zero tail of the packed K axis, so the block runs KT <= Kpad/BK tiles.
  // Otherwise: channel kc of tap (kr, ks) of this lane's logical chunk.
  int kc = ALIGNED ? 0 : chunk * EPC, kr = 0, ks = 0, tap = 0;
  unsigned long long bmask = 0;              // ALIGNED: taps some row of the block reads for real
  int KT = a.Kpad / BK;
  if constexpr (ALIGNED) {
    __shared__ unsigned long long wave_taps[WAVES_M * WAVES_N];
    unsigned long long mine = 0;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) mine |= tapmask[i];
    const int ntaps = a.KH * a.KW + (a.Cin2 > 0 ? 1 : 0);
    unsigned long long wmask = 0;
    for (int t = 0; t < ntaps; ++t)
      if (__ballot((mine >> t) & 1ull) != 0ull) wmask |= 1ull << t;
    if (lane == 0) wave_taps[wave] = wmask;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < WAVES_M * WAVES_N; ++w) bmask |= wave_taps[w];
    bmask = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(bmask >> 32)) << 32) |
            (unsigned)__builtin_amdgcn_readfirstlane((unsigned)bmask);
    KT = __popcll(bmask & ((1ull << (a.KH * a.KW)) - 1ull)) * (a.Cin / BK) + a.Cin2 / BK;
    while (tap < ntaps && !((bmask >> tap) & 1ull)) {
      ++tap;
      if (++ks == a.KW) { ks = 0; ++kr; }
    }
  } else {
    while (kc >= a.Cin) {
      kc -= a.Cin;
      if (++ks == a.KW) { ks = 0; ++kr; }
    }
  }

  // the block's share of its K tiles: all of them, or slice zsplit of a split-K launch
  int kt0 = 0, kt1 = KT;
  if (a.splits > 1) {
    kt0 = (int)((long long)KT * zsplit / a.splits);
    kt1 = (int)((long long)KT * (zsplit + 1) / a.splits);
  }
  // moves the tap walk one K tile on
  auto advance = [&]() {
    kc += BK;
    if constexpr (ALIGNED) {
      const int ntaps_all = a.KH * a.KW + (a.Cin2 > 0 ? 1 : 0);
      if (kc == (tap == a.KH * a.KW ? a.Cin2 : a.Cin)) {
        kc = 0;
        do {
          ++tap;
          if (++ks == a.KW) { ks = 0; ++kr; }
        } while (tap < ntaps_all && !((bmask >> tap) & 1ull));
      }
    } else {
      while (kc >= a.Cin) {
        kc -= a.Cin;
        if (++ks == a.KW) { ks = 0; ++kr; }
      }
    }
  };
  for (int t = 0; t < kt0; ++t) advance();
  // issues the LOADS DMA pieces of K tile `kt` into ring slot `slot`; tiles past
  // the block's last one fetch zeros so every wave always has the same number of pieces in flight
  auto issue = [&](int kt, int slot) {
    const bool in_range = kt < kt1;
    const unsigned dstA = smem_base + slot * STAGE_BYTES + wave * 1024;
    const unsigned dstB = dstA + BM * BKB;
    if constexpr (ALIGNED) {
      const int tbit = min(tap, 63);                    // bit 63 is never set
      // byte offset of the tap inside one image: fits 32 bits (checked by the host)
      const int tap_off = ((kr * a.dil * a.W + ks * a.dil) * a.Cin + kc) * (int)sizeof(T);
      const bool second = a.Cin2 > 0 && tap == a.KH * a.KW;          // uniform
      const unsigned off2 = r2off0 + (unsigned)(kc * (int)sizeof(T));
      const fss_i32x4 rs = second ? rs_x2 : rs_x;                       // uniform
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const bool ok = in_range && ((tapmask[i] >> tbit) & 1ull);
        const unsigned real = second ? off2 + i * r2step : roff[i] + (unsigned)tap_off;
        blds16(ok ? real : kDmaOutOfRange, rs, dstA + i * (NT * 16));
      }
    } else {
      const int dr = kr * a.dil, ds = ks * a.dil;
      const bool k_ok = in_range && kr < a.KH;
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const int ih = ih0[i] + dr, iw = iw0[i] + ds;
        const bool ok = k_ok && ((unsigned)ih < (unsigned)a.H) && ((unsigned)iw < (unsigned)a.W);
        const int off = ((dr * a.W + ds) * a.Cin + (kc - chunk * EPC)) * (int)sizeof(T);
        blds16(ok ? roff[i] + (unsigned)off : kDmaOutOfRange, rs_x, dstA + i * (NT * 16));
      }
    }
    const bool w_ok = in_range;
    // k index of the tile in the packed weights
    const size_t wk = ALIGNED ? (size_t)(tap * a.Cin + kc) * sizeof(T) : (size_t)kt * BKB;
#pragma unroll
    for (int i = 0; i < B_IT; ++i)
      blds16(w_ok ? woff[i] + (unsigned)wk : kDmaOutOfRange, rs_w, dstB + i * (NT * 16));
    if constexpr (WIDE) {
      const unsigned lo_off = (unsigned)wk + (unsigned)(a.Kpad * (int)sizeof(T));
#pragma unroll
      for (int i = 0; i < B_IT; ++i)
        blds16(w_ok ? woff[i] + lo_off : kDmaOutOfRange, rs_w, dstB + BN * BKB + i * (NT * 16));
    }
    advance();
  };

  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int lrow = lane & (FM - 1), lg = lane / FM;
  const int swz = swizzle_of(lrow);
  // byte offset, inside a row, of the fragment this lane reads at k step s
  int frag_off[KSTEPS];
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) frag_off[s] = ((LG * s + lg) ^ swz) * 16;

  typename MMA::acc_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < MMA::NREG; ++r) acc[i][j][r] = 0.f;

  // ---- epilogue geometry (needed now: the residual is fetched before the main loop)
  constexpr int SROW = WTN + 4;              // staging row stride in floats
  // rows staged per wave at a time: 32, or 16 when 32 would not fit in the ring
  constexpr int SLABR = (WAVES_M * WAVES_N * 32 * SROW * 4 <= NSTAGE * STAGE_BYTES) ? 32 : 16;
  static_assert(WAVES_M * WAVES_N * SLABR * SROW * 4 <= NSTAGE * STAGE_BYTES,
                "epilogue staging does not fit in the ring");
  static_assert(SLABR % FM == 0, "a staging slab must hold whole MFMA tiles");
  // outputs per lane per pass: one 16-byte store, or the two of a split-bfloat16 group
  constexpr bool SPLIT_OUT = std::is_same<TO, f32s_t>::value;
  constexpr int VW = SPLIT_OUT ? 8 : 16 / (int)sizeof(TO);
  constexpr int QPR = WTN / VW;              // lane pieces per staged row
  constexpr int PASSES = SLABR * QPR / 64;   // pieces per lane per slab
  static_assert(SLABR * QPR % 64 == 0, "slab must split evenly over the lanes");
  constexpr int SLABS = WTM / SLABR;
  TO* __restrict__ yg = reinterpret_cast<TO*>(a.y) + (size_t)zsplit * a.M * a.y_ld;
  // the residual has the input's type; with half-precision input, the OUTPUT's (a float residual
  // stream can then take the result of a half-precision stretch of the graph)
  typedef typename std::conditional<std::is_same<T, f16_t>::value || WIDE, TO, T>::type TR;
  const TR* __restrict__ rg = reinterpret_cast<const TR*>(a.res);
  const int n_base = tile_n * BN + wn * WTN;
  const bool vec_ok = ((a.y_ld & (VW - 1)) == 0) && ((a.Cout & (VW - 1)) == 0) &&
                      ((reinterpret_cast<uintptr_t>(a.y) & 15) == 0) &&
                      (rg == nullptr ||
                       (((a.res_ld & (VW - 1)) == 0) && ((reinterpret_cast<uintptr_t>(a.res) & 15) == 0)));
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s) issue(kt0 + s, s);

  int slot = 0, fill = NSTAGE - 1;
  for (int kt = kt0; kt < kt1; ++kt) {
    wait_vmcnt<(NSTAGE - 2) * LOADS>();      // this wave's pieces of tile kt have landed
    __builtin_amdgcn_s_barrier();            // ... and everybody else's; slot `fill` is free
    issue(kt + NSTAGE - 1, fill);
    const char* pa = smem + slot * STAGE_BYTES + (wm * WTM + lrow) * BKB;
    const char* pb = smem + slot * STAGE_BYTES + BM * BKB + (wn * WTN + lrow) * BKB;
    if constexpr (MMA::PAIRED) {
      // split mode: per MFMA step a lane group owns the 8 consecutive k values of one chunk
      // pair (2q, 2q + 1) of both operands: their 8 bf16 high parts (chunk 2q) and their 8 bf16
      // remainders (chunk 2q + 1)
      static_assert(KSTEPS % 2 == 0, "split mode pairs the chunks of a tile row");
      // Which chunk pair a lane group takes (the same on both operands, so any order is right).
      // With four lane groups the order 0, 3, 1, 2 keeps the fragment reads free of bank
      // conflicts: ds_read_b128 serves lane groups (0, 1) and (2, 3) together, eight rows of
      // each, and the swizzle separates their 16-byte windows only when the chunk indices of the
      // two groups agree, or are complementary, in their upper two bits - true for chunks
      // (lg, lg + 1) of the bf16 loop, not for the even chunks (2 lg, 2 lg + 2) read here.
      const int qg = LG == 4 ? ((lg & 1) ? 3 - (lg >> 1) : (lg >> 1)) : lg;
#pragma unroll
      for (int s = 0; s < KSTEPS / 2; ++s) {
        const int o0 = ((2 * (LG * s + qg)) ^ swz) * 16, o1 = ((2 * (LG * s + qg) + 1) ^ swz) * 16;
        uint4 ahi[MI], alo[MI], bhi[NI], blo[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          ahi[i] = *reinterpret_cast<const uint4*>(pa + i * FM * BKB + o0);
          alo[i] = *reinterpret_cast<const uint4*>(pa + i * FM * BKB + o1);
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          bhi[j] = *reinterpret_cast<const uint4*>(pb + j * FM * BKB + o0);
          blo[j] = *reinterpret_cast<const uint4*>(pb + j * FM * BKB + o1);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            if constexpr (HEAD) MMA::run3(acc[i][j], bhi[j], blo[j], ahi[i], alo[i]);
            else MMA::run3(acc[i][j], ahi[i], alo[i], bhi[j], blo[j]);
          }
      }
    } else {
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      uint4 fa[MI], fb[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i)
        fa[i] = *reinterpret_cast<const uint4*>(pa + i * FM * BKB + frag_off[s]);
#pragma unroll
      for (int j = 0; j < NI; ++j)
        fb[j] = *reinterpret_cast<const uint4*>(pb + j * FM * BKB + frag_off[s]);
      if constexpr (WIDE) {
        uint4 fl[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j)
          fl[j] = *reinterpret_cast<const uint4*>(pb + BN * BKB + j * FM * BKB + frag_off[s]);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) {                                       // the small terms first
            if constexpr (HEAD) MMA::run(acc[i][j], fl[j], fa[i]);
            else MMA::run(acc[i][j], fa[i], fl[j]);
          }
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          if constexpr (HEAD) MMA::run(acc[i][j], fb[j], fa[i]);
          else MMA::run(acc[i][j], fa[i], fb[j]);
        }
    }
    }
    slot = slot + 1 == NSTAGE ? 0 : slot + 1;
    fill = fill + 1 == NSTAGE ? 0 : fill + 1;
  }
  if constexpr (HEAD) {
    // acc[i][j] is D[channel][pixel]: lane (lrow, lg) holds, for pixel 16 i + lrow of this wave's
    // tile, tile rows 16 j + 4 lg + q = channels 32 (j >> 1) + 8 lg + 4 (j & 1) + q (the weight
    // rows were fetched in that order).  Second product, per 32 channels (two D tiles): k index
    // 8 lg + e of both operands = channel 32 s + 8 lg + e.
    static_assert(FM == 16 && NI % 2 == 0 && (sizeof(T) == 2 || MMA::PAIRED),
                  "head epilogue: 16x16x32 bf16 tiles (bf16 tensors, or float tensors in the split mode)");
    // Split mode: residuals are float (two 16-byte loads per 8 channels), the activated values
    // are split like every other activation and the classifier weights arrive as two bf16
    // planes [32][cout] (high parts, then remainders): three MFMAs per step again.
    // (the same epilogue serves half-precision activations against two-f16 weights: its residuals,
    // interpolation sources and second product are those of the split mode)
    constexpr bool SPLIT = MMA::PAIRED || WIDE;
    constexpr unsigned ESZ = SPLIT ? 4u : (unsigned)sizeof(T);     // bytes per residual element
    // 8 consecutive channels as they lie in memory, and as floats
    struct Raw8 { uint4 a, b; };
    auto raw8 = [](const char* p) {
      Raw8 r;
      r.a = *reinterpret_cast<const uint4*>(p);
      if constexpr (SPLIT) r.b = *reinterpret_cast<const uint4*>(p + 16);
      else r.b = r.a;
      return r;
    };
    auto floats8 = [](const Raw8& r, float (&f)[8]) {
      if constexpr (SPLIT) {
        join8(r.a, r.b, f);
      } else {
        f[0] = __uint_as_float(r.a.x << 16); f[1] = __uint_as_float(r.a.x & 0xffff0000u);
        f[2] = __uint_as_float(r.a.y << 16); f[3] = __uint_as_float(r.a.y & 0xffff0000u);
        f[4] = __uint_as_float(r.a.z << 16); f[5] = __uint_as_float(r.a.z & 0xffff0000u);
        f[6] = __uint_as_float(r.a.w << 16); f[7] = __uint_as_float(r.a.w & 0xffff0000u);
      }
    };
    static_assert(WAVES_N * BM * kHeadMax * 4 <= NSTAGE * STAGE_BYTES, "head staging does not fit");
    wait_vmcnt<0>();
    __syncthreads();                         // everyone is done with the ring
    const float act_lo = a.act == FSS_ACT_NONE ? -INFINITY : 0.f;
    const float act_hi = a.act == FSS_ACT_RELU6 ? 6.f : INFINITY;
    const bf16_t* __restrict__ wh = reinterpret_cast<const bf16_t*>(a.w_head);
    const int n_base = tile_n * BN + wn * WTN;
    f32x4 acc2[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) acc2[i][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Zoomed residual: per 16-pixel group i, the byte offset from zy of this lane's 8 channels
    // (of the wave's first 32) in the top-left source pixel of its 2 x 2 neighbourhood; the
    // offset is a multiple of 16, and its bits 0 / 1 say that the output row / column is odd,
    // i.e. has a second source one row / column further.  32-bit arithmetic: the host checks
    // that zy is below 4 GB.  zborder[i]: some lane of the wave has a pixel next to the frame
    // border, whose sources come from the correction tensors (wave-uniform, rare).  zeven[i]:
    // every lane's output row is even, so the second source row repeats the first.
    const char* __restrict__ zyb = reinterpret_cast<const char*>(a.zy);
    const unsigned zpix = (unsigned)a.Cout * ESZ, zrow = (unsigned)a.zW * zpix;
    unsigned zo[MI] = {};
    bool zborder[MI] = {}, zeven[MI] = {};
    if (a.zy != nullptr) {
      // (frame, row, column) of this lane's pixel in group 0 by division, in the further groups
      // by stepping 16 pixels on (the output is at least 7 wide and high: at most 3 row wraps
      // and one frame wrap per step); pixels past the end of the batch stand in for the last one
      const int m0 = tile_m * BM + wm * WTM + lrow;
      int f = (int)((unsigned)m0 / (unsigned)HoWo);
      const int rem = m0 - f * HoWo;
      int p = (int)((unsigned)rem / (unsigned)a.Wo), q = rem - p * a.Wo;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        if (i > 0) {
          q += 16;
#pragma unroll
          for (int k = 0; k < 3; ++k)
            if (q >= a.Wo) { q -= a.Wo; ++p; }
          if (p >= a.Ho) { p -= a.Ho; ++f; }
        }
        const bool in = m0 + 16 * i < a.M;
        const int fc = in ? f : a.N - 1, pc = in ? p : a.Ho - 1, qc = in ? q : a.Wo - 1;
        zo[i] = ((unsigned)(fc * a.zH + (pc >> 1)) * zrow + (unsigned)(qc >> 1) * zpix +
                 (unsigned)(n_base + 8 * lg) * ESZ) | (unsigned)(pc & 1) | (unsigned)((qc & 1) << 1);
        zborder[i] = __ballot(pc == 1 || pc == a.Ho - 2 || qc == 1 || qc == a.Wo - 2) != 0ull;
        zeven[i] = __ballot(pc & 1) == 0ull;   // every lane's pixel has one source row
      }
    }
#pragma unroll
    for (int s = 0; s < NI / 2; ++s) {
      const int c0 = n_base + 32 * s;
      if (c0 >= a.Cout) break;               // wave-uniform: Cout is a multiple of 32
      uint4 wa[2], wal[SPLIT ? 2 : 1];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        wa[ct] = *reinterpret_cast<const uint4*>(wh + (size_t)(ct * 16 + lrow) * a.Cout + c0 + 8 * lg);
        if constexpr (SPLIT)
          wal[ct] = *reinterpret_cast<const uint4*>(wh + (size_t)(32 + ct * 16 + lrow) * a.Cout + c0 + 8 * lg);
      }
      float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
      if (a.bias != nullptr) {
        b0 = *reinterpret_cast<const float4*>(a.bias + c0 + 8 * lg);
        b1 = *reinterpret_cast<const float4*>(a.bias + c0 + 8 * lg + 4);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        f32x4 d0 = acc[i][2 * s], d1 = acc[i][2 * s + 1];
        if (a.zy != nullptr) {
          // residual = corrected x2 zoom of zy at this lane's pixel: the mean of 1, 2 or 4
          // sources, formed branch-free as a quarter of (u0 + u1) + (u2 + u3) over the 2 x 2
          // neighbourhood (a source counted twice or four times where a step is 0), so that the
          // loads are independent and in flight together.  Where the whole wave has even rows,
          // u2 = u0 and u3 = u1: half of (u0 + u1) is the same number, from two loads.
          if constexpr (!SPLIT) {
            uint4 u[4];
            const unsigned zbase = (zo[i] & ~15u) + (unsigned)(64 * s);
            const unsigned zdr = (zo[i] & 1u) ? zrow : 0u, zdc = (zo[i] & 2u) ? zpix : 0u;
            const bool two = zeven[i] && !zborder[i];
            if (two) {
              u[0] = *reinterpret_cast<const uint4*>(zyb + zbase);
              u[1] = *reinterpret_cast<const uint4*>(zyb + (zbase + zdc));
              u[2] = u[3] = make_uint4(0u, 0u, 0u, 0u);
            } else if (!zborder[i]) {
#pragma unroll
              for (int nb = 0; nb < 4; ++nb)
                u[nb] = *reinterpret_cast<const uint4*>(zyb + (zbase + ((nb & 2) ? zdr : 0u) + ((nb & 1) ? zdc : 0u)));
            } else {
              // next to the frame border a source row / column is replaced by its corrected copy
              const int m = min(tile_m * BM + wm * WTM + i * 16 + lrow, a.M - 1);
              const int f = m / HoWo, rem = m - f * HoWo;
              const int p = rem / a.Wo, q = rem - p * a.Wo;
              const unsigned lane_c = (unsigned)(c0 + 8 * lg) * (unsigned)sizeof(T);
#pragma unroll
              for (int nb = 0; nb < 4; ++nb) {
                const int zi = (p >> 1) + ((nb >> 1) & p), zj = (q >> 1) + (nb & q & 1);
                const int rf = (p == 1 && zi == 1) ? 1 : (p == a.Ho - 2 && zi == a.zH - 2) ? 2 : 0;
                const int cf = (q == 1 && zj == 1) ? 1 : (q == a.Wo - 2 && zj == a.zW - 2) ? 2 : 0;
                const char* base = zyb;
                unsigned off = zbase + ((nb & 2) ? zdr : 0u) + ((nb & 1) ? zdc : 0u);
                if (rf != 0 && cf == 0) {
                  base = reinterpret_cast<const char*>(a.zytb);
                  off = (unsigned)(((rf - 1) * a.N + f) * a.zW + zj) * zpix + lane_c;
                }
                if (rf == 0 && cf != 0) {
                  base = reinterpret_cast<const char*>(a.zylr);
                  off = (unsigned)(((cf - 1) * a.N + f) * a.zH + zi) * zpix + lane_c;
                }
                if (rf != 0 && cf != 0) {
                  base = reinterpret_cast<const char*>(a.zycn);
                  off = (unsigned)(((rf - 1) * 2 + (cf - 1)) * a.N + f) * zpix + lane_c;
                }
                u[nb] = *reinterpret_cast<const uint4*>(base + off);
              }
            }
            auto lo = [](unsigned v) { return __uint_as_float(v << 16); };
            auto hi = [](unsigned v) { return __uint_as_float(v & 0xffff0000u); };
            float r0[4], r1[4];
            r0[0] = lo(u[0].x) + lo(u[1].x); r0[1] = hi(u[0].x) + hi(u[1].x);
            r0[2] = lo(u[0].y) + lo(u[1].y); r0[3] = hi(u[0].y) + hi(u[1].y);
            r1[0] = lo(u[0].z) + lo(u[1].z); r1[1] = hi(u[0].z) + hi(u[1].z);
            r1[2] = lo(u[0].w) + lo(u[1].w); r1[3] = hi(u[0].w) + hi(u[1].w);
            if (two) {
#pragma unroll
              for (int e = 0; e < 4; ++e) { d0[e] += r0[e] * 0.5f; d1[e] += r1[e] * 0.5f; }
            } else {
              r0[0] += lo(u[2].x) + lo(u[3].x); r0[1] += hi(u[2].x) + hi(u[3].x);
              r0[2] += lo(u[2].y) + lo(u[3].y); r0[3] += hi(u[2].y) + hi(u[3].y);
              r1[0] += lo(u[2].z) + lo(u[3].z); r1[1] += hi(u[2].z) + hi(u[3].z);
              r1[2] += lo(u[2].w) + lo(u[3].w); r1[3] += hi(u[2].w) + hi(u[3].w);
#pragma unroll
              for (int e = 0; e < 4; ++e) { d0[e] += r0[e] * 0.25f; d1[e] += r1[e] * 0.25f; }
            }
          } else {
            // float sources (32 bytes per 8 channels): the addresses of the (up to) four sources
            // first
            const unsigned zbase = (zo[i] & ~15u) + (unsigned)(32 * s) * ESZ;
            const unsigned zdr = (zo[i] & 1u) ? zrow : 0u, zdc = (zo[i] & 2u) ? zpix : 0u;
            const bool two = zeven[i] && !zborder[i];
            const char* src[4];
            if (!zborder[i]) {
#pragma unroll
              for (int nb = 0; nb < 4; ++nb)
                src[nb] = zyb + (zbase + ((nb & 2) ? zdr : 0u) + ((nb & 1) ? zdc : 0u));
            } else {
              // next to the frame border a source row / column is replaced by its corrected copy
              const int m = min(tile_m * BM + wm * WTM + i * 16 + lrow, a.M - 1);
              const int f = m / HoWo, rem = m - f * HoWo;
              const int p = rem / a.Wo, q = rem - p * a.Wo;
              const unsigned lane_c = (unsigned)(c0 + 8 * lg) * ESZ;
#pragma unroll
              for (int nb = 0; nb < 4; ++nb) {
                const int zi = (p >> 1) + ((nb >> 1) & p), zj = (q >> 1) + (nb & q & 1);
                const int rf = (p == 1 && zi == 1) ? 1 : (p == a.Ho - 2 && zi == a.zH - 2) ? 2 : 0;
                const int cf = (q == 1 && zj == 1) ? 1 : (q == a.Wo - 2 && zj == a.zW - 2) ? 2 : 0;
                const char* base = zyb;
                unsigned off = zbase + ((nb & 2) ? zdr : 0u) + ((nb & 1) ? zdc : 0u);
                if (rf != 0 && cf == 0) {
                  base = reinterpret_cast<const char*>(a.zytb);
                  off = (unsigned)(((rf - 1) * a.N + f) * a.zW + zj) * zpix + lane_c;
                }
                if (rf == 0 && cf != 0) {
                  base = reinterpret_cast<const char*>(a.zylr);
                  off = (unsigned)(((cf - 1) * a.N + f) * a.zH + zi) * zpix + lane_c;
                }
                if (rf != 0 && cf != 0) {
                  base = reinterpret_cast<const char*>(a.zycn);
                  off = (unsigned)(((rf - 1) * 2 + (cf - 1)) * a.N + f) * zpix + lane_c;
                }
                src[nb] = base + off;
              }
            }
            // the sources are plain floats here (nothing else reads them, so their producer skips
            // the split): 8 channels = two 16-byte loads per source, summed as they arrive
            float r8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
              const float4 t0 = *reinterpret_cast<const float4*>(src[0] + 16 * hh);
              const float4 t1 = *reinterpret_cast<const float4*>(src[1] + 16 * hh);
              float4 t = make_float4(t0.x + t1.x, t0.y + t1.y, t0.z + t1.z, t0.w + t1.w);
              if (!two) {
                const float4 t2 = *reinterpret_cast<const float4*>(src[2] + 16 * hh);
                const float4 t3 = *reinterpret_cast<const float4*>(src[3] + 16 * hh);
                t.x += t2.x + t3.x; t.y += t2.y + t3.y; t.z += t2.z + t3.z; t.w += t2.w + t3.w;
              }
              r8[4 * hh] = t.x; r8[4 * hh + 1] = t.y; r8[4 * hh + 2] = t.z; r8[4 * hh + 3] = t.w;
            }
            const float zsc = two ? 0.5f : 0.25f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { d0[e] += r8[e] * zsc; d1[e] += r8[4 + e] * zsc; }
          }
        } else if (a.res != nullptr) {
          if constexpr (!SPLIT) {
            const int m = tile_m * BM + wm * WTM + i * 16 + lrow;
            if (m < a.M) {
              const uint4 rv = *reinterpret_cast<const uint4*>(
                  reinterpret_cast<const T*>(a.res) + (size_t)m * a.res_ld + c0 + 8 * lg);
              d0[0] += __uint_as_float(rv.x << 16); d0[1] += __uint_as_float(rv.x & 0xffff0000u);
              d0[2] += __uint_as_float(rv.y << 16); d0[3] += __uint_as_float(rv.y & 0xffff0000u);
              d1[0] += __uint_as_float(rv.z << 16); d1[1] += __uint_as_float(rv.z & 0xffff0000u);
              d1[2] += __uint_as_float(rv.w << 16); d1[3] += __uint_as_float(rv.w & 0xffff0000u);
            }
          } else {
            const int m = tile_m * BM + wm * WTM + i * 16 + lrow;
            if (m < a.M) {
              float f0[8];
              floats8(raw8(reinterpret_cast<const char*>(a.res) + ((size_t)m * a.res_ld + c0 + 8 * lg) * ESZ), f0);
#pragma unroll
              for (int e = 0; e < 4; ++e) { d0[e] += f0[e]; d1[e] += f0[4 + e]; }
            }
          }
        }
        auto cl = [&](float v) { return __builtin_amdgcn_fmed3f(v, act_lo, act_hi); };
        const float v8[8] = {cl(d0[0] + b0.x), cl(d0[1] + b0.y), cl(d0[2] + b0.z), cl(d0[3] + b0.w),
                             cl(d1[0] + b1.x), cl(d1[1] + b1.y), cl(d1[2] + b1.z), cl(d1[3] + b1.w)};
        if constexpr (SPLIT) {
          uint4 fh, fl;
          split8(v8, fh, fl);
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) Mma<f32s_t, 16>::run3(acc2[i][ct], wa[ct], wal[ct], fh, fl);
        } else {
          const uint4 fb = make_uint4(pack_bf16x2(v8[0], v8[1]), pack_bf16x2(v8[2], v8[3]),
                                      pack_bf16x2(v8[4], v8[5]), pack_bf16x2(v8[6], v8[7]));
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) MMA::run(acc2[i][ct], wa[ct], fb);
        }
      }
    }
    // acc2[i][ct][q] = partial score of class 16 ct + 4 lg + q, pixel 16 i + lrow, over this
    // wave's channels: one staging plane per channel slice, summed below in a fixed order
    float* hs = reinterpret_cast<float*>(smem);
    static_assert(kHeadMax == 20, "head staging: 16 classes from the first tile, 4 from the second");
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      // lane group lg holds classes 4 lg .. 4 lg + 3 (first tile) and 16 + 4 lg .. (second tile,
      // of which only group 0's exist)
      float* row = hs + ((wn * BM) + wm * WTM + i * 16 + lrow) * kHeadMax;
      *reinterpret_cast<float4*>(row + 4 * lg) =
          make_float4(acc2[i][0][0], acc2[i][0][1], acc2[i][0][2], acc2[i][0][3]);
      if (lg == 0)
        *reinterpret_cast<float4*>(row + 16) =
            make_float4(acc2[i][1][0], acc2[i][1][1], acc2[i][1][2], acc2[i][1][3]);
    }
    __syncthreads();
    TO* __restrict__ yh = reinterpret_cast<TO*>(a.y);
    const int nc = a.cout_head;
    // one work item = 4 consecutive classes of one pixel, channel slices summed in index order
    for (int e = tid; e < BM * (kHeadMax / 4); e += NT) {
      const int px = e / (kHeadMax / 4), c = (e - px * (kHeadMax / 4)) * 4;
      const int m = tile_m * BM + px;
      if (m >= a.M) break;                   // px grows with e
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = (a.bias_head != nullptr && c + k < nc) ? a.bias_head[c + k] : 0.f;
#pragma unroll
      for (int w = 0; w < WAVES_N; ++w) {
        const float4 t = *reinterpret_cast<const float4*>(hs + (w * BM + px) * kHeadMax + c);
        v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
      }
      TO* dst = yh + (size_t)m * a.y_ld + c;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (c + k < nc) ElemTraits<TO>::store(dst + k, v[k]);
    }
    return;
  }
  // Residual of this lane's output pieces (bf16: 4 values = 8 bytes per piece): loaded
  // here, where the gather registers are dead, so that its latency overlaps the drain
  // of the ring, the barrier and the staging of the first slab.  The loads are younger
  // than every DMA piece, so waiting for all but them drains the ring.
  constexpr bool RES_PREFETCH = sizeof(TR) == 2;
  // VW bf16 residual values of one piece: 16 bytes (bf16 output) or 8 bytes (float output)
  typedef typename std::conditional<VW == 8, uint4, uint2>::type rvec_t;
  rvec_t rpre[RES_PREFETCH ? SLABS : 1][RES_PREFETCH ? PASSES : 1];
  bool drained = false;
  if constexpr (RES_PREFETCH) {
    if (rg != nullptr && vec_ok) {
#pragma unroll
      for (int sl = 0; sl < SLABS; ++sl)
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
          const int gq = lane + p * 64;
          const int m = tile_m * BM + wm * WTM + sl * SLABR + gq / QPR, n = n_base + (gq % QPR) * VW;
          const bool ok = m < a.M && n < a.Cout;
          const TR* rp = ok ? rg + (size_t)m * a.res_ld + n : reinterpret_cast<const TR*>(g_zero_chunk);
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
  }
  if (!drained) wait_vmcnt<0>();             // drain the zero-fill tail pieces
  __syncthreads();                           // everyone is done with the ring

  // ---- epilogue: per wave, 32 rows at a time through LDS (MMA::row_of maps an
  // accumulator register to its row inside the FM x FM tile; the column is lrow).
  float* stage = reinterpret_cast<float*>(smem) + wave * (SLABR * SROW);
  // branch-free activation: none / relu / relu6 as a clamp to [lo, hi]
  // (half-precision outputs saturate instead of overflowing to infinity)
  constexpr float kOutMax = std::is_same<TO, f16_t>::value ? 65504.f : INFINITY;
  const float act_lo = a.act == FSS_ACT_NONE ? -kOutMax : 0.f;
  const float act_hi = a.act == FSS_ACT_RELU6 ? 6.f : kOutMax;
  constexpr int TPS = SLABR / FM;            // MFMA tile rows per slab
#pragma unroll
  for (int sl = 0; sl < SLABS; ++sl) {
    // split-bfloat16 residual of this slab's pieces: requested before the slab is staged, so
    // that the loads of all passes are in flight together and behind the LDS traffic (clamped
    // addresses: a piece past the end of the tensor reads a valid one and is never stored)
    constexpr bool RES_SLAB = std::is_same<TR, f32s_t>::value && VW == 8;
    uint4 rs[RES_SLAB ? PASSES : 1][2];
    if constexpr (RES_SLAB) {
      if (rg != nullptr && vec_ok) {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
          const int g = lane + p * 64;
          const int m = min(tile_m * BM + wm * WTM + sl * SLABR + g / QPR, a.M - 1);
          const int n = min(n_base + (g % QPR) * VW, a.Cout - VW);
          const uint4* rq = reinterpret_cast<const uint4*>(rg + (size_t)m * a.res_ld + n);
          rs[p][0] = rq[0];
          rs[p][1] = rq[1];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n_base + j * FM + lrow;
      const float bias = (a.bias != nullptr && n < a.Cout) ? a.bias[n] : 0.f;
#pragma unroll
      for (int ti = 0; ti < TPS; ++ti)
#pragma unroll
        for (int r = 0; r < MMA::NREG; ++r)
          stage[(ti * FM + MMA::row_of(r, lg)) * SROW + j * FM + lrow] = acc[sl * TPS + ti][j][r] + bias;
    }
    wave_lds_fence();   // the slab is private to this wave: LDS order within a wave suffices
    const int m_base = tile_m * BM + wm * WTM + sl * SLABR;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int g = lane + p * 64;
      const int row = g / QPR, cv = (g % QPR) * VW;
      const int m = m_base + row, n = n_base + cv;
      if (m < a.M && n < a.Cout) {
        float v[VW];
#pragma unroll
        for (int q = 0; q < VW; q += 4) {
          const float4 v4 = *reinterpret_cast<const float4*>(stage + row * SROW + cv + q);
          v[q] = v4.x; v[q + 1] = v4.y; v[q + 2] = v4.z; v[q + 3] = v4.w;
        }
        TO* dst = yg + (size_t)m * a.y_ld + n;
        if (vec_ok) {
          if (rg != nullptr) {
            if constexpr (std::is_same<TR, f32s_t>::value) {
              if constexpr (VW == 8) {
                float r8[8];
                join8(rs[p][0], rs[p][1], r8);
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] += r8[q];
              } else {
#pragma unroll
                for (int q = 0; q < VW; ++q) v[q] += ElemTraits<TR>::load(rg + (size_t)m * a.res_ld + n + q);
              }
            } else if constexpr (sizeof(TR) == 4) {
#pragma unroll
              for (int q = 0; q < VW; q += 4) {
                const float4 r4 = *reinterpret_cast<const float4*>(rg + (size_t)m * a.res_ld + n + q);
                v[q] += r4.x; v[q + 1] += r4.y; v[q + 2] += r4.z; v[q + 3] += r4.w;
              }
            } else {
              const rvec_t rv = rpre[sl][p];
              const unsigned* rw = reinterpret_cast<const unsigned*>(&rv);
#pragma unroll
              for (int q = 0; q < VW / 2; ++q) {
                v[2 * q] += Half16<TR>::lo(rw[q]);
                v[2 * q + 1] += Half16<TR>::hi(rw[q]);
              }
            }
          }
#pragma unroll
          for (int e = 0; e < VW; ++e) v[e] = fminf(fmaxf(v[e], act_lo), act_hi);
          if constexpr (SPLIT_OUT) {
            uint4 hi, lo;
            split8(v, hi, lo);
            store16_nt(dst, hi.x, hi.y, hi.z, hi.w);
            store16_nt(dst + 4, lo.x, lo.y, lo.z, lo.w);
          } else if constexpr (sizeof(TO) == 4) {
            store16_nt(dst, __float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
          } else {
            typedef Half16<TO> HO;
            store16_nt(dst, HO::pack2(v[0], v[1]), HO::pack2(v[2], v[3]), HO::pack2(v[4], v[5]),
                       HO::pack2(v[6], v[7]));
          }
        } else {
#pragma unroll
          for (int e = 0; e < VW; ++e) {
            if (n + e < a.Cout) {
              float t = v[e];
              if (rg != nullptr) t += ElemTraits<TR>::load(rg + (size_t)m * a.res_ld + n + e);
              ElemTraits<TO>::store(dst + e, fminf(fmaxf(t, act_lo), act_hi));
            }
          }
        }
      }
    }
    wave_lds_fence();   // the slab is private to this wave: LDS order within a wave suffices
  }
}

template <typename T, typename TO, int BM, int BN, int WAVES_M, int WAVES_N, int BKB, int NSTAGE,
          int FM, bool ALIGNED, bool HEAD = false>
static int launch_conv_impl(ConvArgs& a, hipStream_t stream) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr size_t smem = (size_t)NSTAGE * (BM + (std::is_same<T, f16w_t>::value ? 2 : 1) * BN) * BKB;
  a.tiles_m = ceil_div(a.M, BM);
  a.tiles_n = ceil_div(a.Cout, BN);
  const long long blocks = (long long)a.tiles_m * a.tiles_n * a.splits;
  if (blocks <= 0 || blocks > 0x7fffffffLL) {
    set_error("fss_conv2d_nhwc: grid out of range");
    return 2;
  }
  auto kern = conv_igemm_kernel<T, TO, BM, BN, WAVES_M, WAVES_N, BKB, NSTAGE, FM, ALIGNED, HEAD>;
  if (smem > 64 * 1024) {
    static bool attr_set = false;   // per instantiation
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) {
        set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
        return 1;
      }
      attr_set = true;
    }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NT), smem, stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error(std::string("conv_igemm launch: ") + hipGetErrorString(e));
    return 3;
  }
  return 0;
}

template <typename T, typename TO, int BM, int BN, int WAVES_M, int WAVES_N, int BKB, int NSTAGE, int FM>
static int launch_conv(ConvArgs& a, hipStream_t stream) {
  constexpr int BK = BKB / (int)sizeof(T);
  if constexpr (std::is_same<T, f32s_t>::value && (BKB / (16 * (64 / FM))) % 2 != 0) {
    set_error("fss_conv2d_nhwc: FSS_F32_BF16X3 needs a configuration with an even number of k steps per "
              "tile (bytes of K per row a multiple of 2048 / mfma_edge)");
    return 2;
  } else {
  if (a.Cin % BK == 0 && a.Cin2 % BK == 0 && a.KH * a.KW <= 62)
    return launch_conv_impl<T, TO, BM, BN, WAVES_M, WAVES_N, BKB, NSTAGE, FM, true>(a, stream);
  if (a.Cin2 > 0) {
    set_error("fss_conv2d_dual_nhwc: cin and cin2 must be multiples of the K tile");
    return 2;
  }
  return launch_conv_impl<T, TO, BM, BN, WAVES_M, WAVES_N, BKB, NSTAGE, FM, false>(a, stream);
  }
}

// Tile configurations.  Every configuration is legal for every layer (tiles
// larger than the problem only waste lanes).  Configurations with the same MFMA
// shape accumulate each output element over k in the same order, so their results
// are bit-identical; the two shapes differ in the last float bits.  Which one is
// fastest depends on M, N, K and is best measured (fss.plan times the candidates
// once per distinct layer shape).  The 16x16 shape does the same FLOP per cycle as
// the 32x32 one; the chip holds a higher clock on it under sustained load.
//   columns: BM, BN, waves M x N, bytes of K per tile row, ring stages, MFMA tile edge
#define FSS_CONV_CONFIGS(X)        \
  X(1, 128, 128, 2, 2, 64, 4, 32)   \
  X(2, 128, 128, 2, 2, 64, 3, 32)   \
  X(3, 128, 128, 2, 2, 128, 2, 32)  \
  X(4, 256, 128, 4, 2, 128, 2, 32)  \
  X(5, 128, 64, 2, 2, 128, 2, 32)   \
  X(6, 128, 64, 2, 2, 64, 4, 32)    \
  X(7, 128, 64, 2, 2, 64, 3, 32)    \
  X(8, 128, 64, 2, 2, 64, 2, 32)    \
  X(9, 64, 64, 2, 2, 128, 2, 32)    \
  X(10, 64, 64, 2, 2, 64, 4, 32)    \
  X(11, 128, 32, 2, 1, 64, 4, 32)   \
  X(12, 128, 32, 2, 1, 64, 2, 32)   \
  X(13, 256, 128, 4, 2, 64, 3, 32)  \
  X(14, 128, 128, 2, 2, 64, 4, 16)  \
  X(15, 128, 128, 2, 2, 64, 3, 16)  \
  X(16, 128, 128, 2, 2, 128, 2, 16) \
  X(17, 256, 128, 4, 2, 128, 2, 16) \
  X(18, 128, 64, 2, 2, 128, 2, 16)  \
  X(19, 128, 64, 2, 2, 64, 4, 16)   \
  X(20, 128, 64, 2, 2, 64, 3, 16)   \
  X(21, 128, 64, 2, 2, 64, 2, 16)   \
  X(22, 64, 64, 2, 2, 128, 2, 16)   \
  X(23, 64, 64, 2, 2, 64, 4, 16)    \
  X(24, 128, 32, 2, 1, 64, 4, 16)   \
  X(25, 128, 32, 2, 1, 64, 2, 16)   \
  X(26, 256, 128, 4, 2, 64, 3, 16)  \
  X(27, 256, 256, 4, 4, 64, 3, 16)  \
  X(28, 256, 256, 4, 4, 128, 2, 16) \
  X(29, 128, 128, 2, 4, 128, 2, 16) \
  X(30, 128, 128, 4, 2, 128, 2, 16) \
  X(31, 128, 128, 2, 4, 64, 3, 16)  \
  X(32, 128, 64, 4, 2, 128, 2, 16)  \
  X(33, 128, 256, 2, 4, 64, 2, 16)  \
  X(34, 128, 256, 2, 4, 64, 3, 16)
// configuration 35: the 8-wave, 8-phase 256x256 kernel of fss_conv8.hip (bf16 in and out, cin in
// whole K tiles); for every other layer it stands for configuration 28, whose results are the same
static constexpr int kConv8Config = 35;
static constexpr int kNumConvConfigs = 35;

// automatic choice when the caller does not name a configuration
static int default_config(const ConvArgs& a) {
  if (a.Cout <= 32) return 11;
  if (a.K <= 256) return a.Cout <= 64 ? 6 : 8;          // short K: memory bound, favour occupancy
  if (a.Cout <= 64) return 6;
  const long long tiles = (long long)ceil_div(a.M, 128) * ceil_div(a.Cout, 128);
  if (tiles >= 1024) return 3;
  if (tiles >= 192) return 2;
  return 9;
}

template <typename T, typename TO>
static int dispatch_conv(ConvArgs& a, int config, hipStream_t stream) {
  // half-precision tensors (in or out) come with the 16x16 MFMA configurations only
  constexpr bool kWide = std::is_same<T, f16w_t>::value;
  constexpr bool kHalf = std::is_same<T, f16_t>::value || std::is_same<TO, f16_t>::value || kWide;
  if (config == 0) {
    if (!kHalf) config = default_config(a);
    else if (std::is_same<T, f32s_t>::value) config = a.Cout <= 64 ? 18 : 16;   // even k steps per tile
    else if (kWide) config = a.Cout <= 32 ? 24 : a.Cout <= 64 ? 19 : 14;                // rings that fit
    else config = a.Cout <= 32 ? 24 : a.Cout <= 64 ? 19 : 16;
  }
  if (config == kConv8Config) {
    if constexpr (std::is_same<T, bf16_t>::value && std::is_same<TO, bf16_t>::value) {
      const int rc = launch_conv8_bf16(a, stream);
      if (rc >= 0) return rc;
    }
    config = 28;
  }
  switch (config) {
#define FSS_CASE(ID, BM, BN, WM, WN, BKB, NS, FM)                                        \
  case ID:                                                                               \
    if constexpr ((!kHalf || FM == 16) && (!kWide || NS * (BM + 2 * BN) * BKB < 160 * 1024)) \
      return launch_conv<T, TO, BM, BN, WM, WN, BKB, NS, FM>(a, stream);                 \
    else break;
    FSS_CONV_CONFIGS(FSS_CASE)
#undef FSS_CASE
    default: break;
  }
  set_error(kWide ? "fss_conv2d_nhwc: FSS_F16_W2 needs a 16x16-MFMA configuration whose ring fits the LDS "
                    "with two weight tiles per stage"
            : kHalf ? "fss_conv2d_nhwc: FSS_F16 tensors need a 16x16-MFMA configuration"
                    : "fss_conv2d_nhwc: unknown config");
  return 2;
}

// The half-precision instantiations live in a translation unit of their own (fss_conv_f16.hip):
// in_dtype FSS_F16 with out_dtype FSS_F16 / FSS_F32_BF16X3 / FSS_F32, or in_dtype FSS_F32_BF16X3
// with out_dtype FSS_F16.
int dispatch_conv_half(ConvArgs& a, int in_dtype, int out_dtype, int config, hipStream_t stream);
// ... and the classifier-head form for FSS_F16_W2 input
int dispatch_conv_head_wide(ConvArgs& a, int config, hipStream_t stream);

// Classifier-head form: the 16x16x32 configurations whose N tile holds every output channel
// of the layer (BN >= 128 and cout <= BN); -1 = this configuration cannot.
template <typename T, int BM, int BN, int WM, int WN, int BKB, int NS, int FM>
static int launch_conv_head(ConvArgs& a, hipStream_t stream) {
  // split mode: only the configurations with an even number of k steps per tile; two-f16
  // weights: only those whose ring fits the LDS with two weight tiles per stage
  constexpr bool kFits = std::is_same<T, bf16_t>::value ||
                         (std::is_same<T, f32s_t>::value && (BKB / 64) % 2 == 0) ||
                         (std::is_same<T, f16w_t>::value && NS * (BM + 2 * BN) * BKB < 160 * 1024);
  if constexpr (FM == 16 && BN >= 128 && kFits) {
    if (a.Cout > BN) return -1;
    return launch_conv_impl<T, float, BM, BN, WM, WN, BKB, NS, FM, true, true>(a, stream);
  } else {
    return -1;
  }
}

template <typename T>
static int dispatch_conv_head(ConvArgs& a, int config, hipStream_t stream) {
  if (config == 0) {
    if (std::is_same<T, f16w_t>::value) config = a.Cout > 128 ? 27 : 31;
    else config = a.Cout > 128 ? 28 : 29;
  }
  if (config == kConv8Config) config = 28;
  int rc = -1;
  switch (config) {
#define FSS_CASE(ID, BM, BN, WM, WN, BKB, NS, FM) \
  case ID: rc = launch_conv_head<T, BM, BN, WM, WN, BKB, NS, FM>(a, stream); break;
    FSS_CONV_CONFIGS(FSS_CASE)
#undef FSS_CASE
    default: break;
  }
  if (rc < 0) {
    set_error("fss_conv2d_dual_head_nhwc: needs a 16x16x32 configuration with an N tile >= cout "
              "(FSS_F32_BF16X3: and 128 bytes of K per tile row)");
    return 2;
  }
  return rc;
}

}  // namespace fss
