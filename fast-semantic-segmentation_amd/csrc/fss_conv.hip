// Implicit-GEMM convolution for gfx950 (MI355X) on the matrix cores.
//
//   C[M][N] = A[M][K] * B[K][N],  M = n*ho*wo (output pixels), N = cout,
//   K = kh*kw*cin with k = (r*kw + s)*cin + c.
//
// A is never materialised: every 16-byte chunk of an A row is cin-contiguous
// in the NHWC input, so a thread gathers it with one predicated 16-byte load
// (zero fill for padding taps, rows past M and k past K).  B is the packed
// weight matrix Wp[cout_pad][k_pad] (K contiguous), so both operands land in
// LDS as rows of BKB data bytes + 16 pad bytes; that row stride keeps the
// ds_read_b128 fragment reads bank-conflict free.  One LDS byte layout serves
// both dtypes: lane (r = lane&31, h = lane>>5) reads 16 bytes at 32*s + 16*h of
// row r, which is 8 consecutive k for v_mfma_f32_32x32x16_bf16 and 4 k values
// (consumed by 4 x v_mfma_f32_32x32x2_f32, with the same k permutation on both
// operands) for float.
//
// Global -> register -> LDS staging is double buffered: the loads of tile t+1
// are issued before the MFMAs of tile t and written to the other LDS buffer
// after them, one barrier per K tile.
//
// Epilogue: + bias (folded batch norm shift / conv bias), + residual,
// activation, store with an output pixel stride (so a branch can write into a
// channel slice of a concat buffer).
#include "fss_common.h"

namespace fss {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

struct ConvArgs {
  const void* x;
  const void* w;
  const float* bias;
  const void* res;
  void* y;
  int N, H, W, Cin, Cout, KH, KW, stride, dil, pad_t, pad_l, Ho, Wo, act;
  int M, K, Kpad, y_ld, res_ld, tiles_m, tiles_n;
};

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static __device__ __forceinline__ void run(f32x16& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                  __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static __device__ __forceinline__ void run(f32x16& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  }
};

template <typename T, typename TO, int BM, int BN, int WAVES_M, int WAVES_N, int BKB>
__global__ void __launch_bounds__(WAVES_M* WAVES_N * 64)
conv_igemm_kernel(const ConvArgs a) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int EPC = 16 / (int)sizeof(T);  // elements per 16-byte chunk
  constexpr int CPR = BKB / 16;             // chunks per tile row
  constexpr int BK = CPR * EPC;             // K elements per tile
  constexpr int LROW = BKB + 16;            // LDS row stride in bytes
  constexpr int A_CHUNKS = BM * CPR, B_CHUNKS = BN * CPR;
  constexpr int A_IT = (A_CHUNKS + NT - 1) / NT, B_IT = (B_CHUNKS + NT - 1) / NT;
  constexpr int RSTEP = NT / CPR;           // row step between a thread's chunks
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MI = WTM / 32, NI = WTN / 32;
  static_assert(NT % CPR == 0, "thread count must be a multiple of chunks per row");
  static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be a multiple of 32x32");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;                       // [2][BM][LROW]
  char* sB = smem + 2 * BM * LROW;       // [2][BN][LROW]

  // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2), so give
  // each XCD a contiguous range of tiles; n-tiles of one m-tile are adjacent.
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
  const int tile_n = wg % a.tiles_n, tile_m = wg / a.tiles_n;

  const int tid = threadIdx.x;
  const int chunk = tid % CPR;
  const int row0 = tid / CPR;

  const T* __restrict__ xg = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ wgt = reinterpret_cast<const T*>(a.w);

  // per-row gather state
  const T* rbase[A_IT];
  int ih0[A_IT], iw0[A_IT];
  const int HoWo = a.Ho * a.Wo;
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int row = row0 + i * RSTEP;
    const int m = tile_m * BM + row;
    if (row < BM && m < a.M) {
      const int n = m / HoWo;
      const int rem = m - n * HoWo;
      const int oh = rem / a.Wo;
      const int ow = rem - oh * a.Wo;
      ih0[i] = oh * a.stride - a.pad_t;
      iw0[i] = ow * a.stride - a.pad_l;
      rbase[i] = xg + (size_t)n * a.H * a.W * a.Cin;
    } else {
      ih0[i] = -(1 << 28);
      iw0[i] = -(1 << 28);
      rbase[i] = xg;
    }
  }
  // position of this thread's chunk inside the filter: channel kc of tap (kr, ks)
  int kc = chunk * EPC, kr = 0, ks = 0;
  while (kc >= a.Cin) {
    kc -= a.Cin;
    if (++ks == a.KW) { ks = 0; ++kr; }
  }

  const T* wrow[B_IT];
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    const int row = row0 + i * RSTEP;
    wrow[i] = wgt + (size_t)(tile_n * BN + (row < BN ? row : 0)) * a.Kpad + chunk * EPC;
  }

  uint4 ra[A_IT], rb[B_IT];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int ih = ih0[i] + kr * a.dil;
      const int iw = iw0[i] + ks * a.dil;
      const bool ok = (kr < a.KH) && ((unsigned)ih < (unsigned)a.H) && ((unsigned)iw < (unsigned)a.W);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ok) v = *reinterpret_cast<const uint4*>(rbase[i] + ((size_t)ih * a.W + iw) * a.Cin + kc);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i)
      rb[i] = *reinterpret_cast<const uint4*>(wrow[i] + (size_t)kt * BK);
    kc += BK;
    while (kc >= a.Cin) {
      kc -= a.Cin;
      if (++ks == a.KW) { ks = 0; ++kr; }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int row = row0 + i * RSTEP;
      if (A_CHUNKS % NT == 0 || row < BM)
        *reinterpret_cast<uint4*>(sA + (buf * BM + row) * LROW + chunk * 16) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int row = row0 + i * RSTEP;
      if (B_CHUNKS % NT == 0 || row < BN)
        *reinterpret_cast<uint4*>(sB + (buf * BN + row) * LROW + chunk * 16) = rb[i];
    }
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int l31 = lane & 31, lh = lane >> 5;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int KT = a.Kpad / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < KT) load_tile(kt + 1);
    const char* pa = sA + (cur * BM + wm * WTM + l31) * LROW + lh * 16;
    const char* pb = sB + (cur * BN + wn * WTN + l31) * LROW + lh * 16;
#pragma unroll
    for (int s = 0; s < BKB / 32; ++s) {
      uint4 fa[MI], fb[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i)
        fa[i] = *reinterpret_cast<const uint4*>(pa + i * 32 * LROW + s * 32);
#pragma unroll
      for (int j = 0; j < NI; ++j)
        fb[j] = *reinterpret_cast<const uint4*>(pb + j * 32 * LROW + s * 32);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) Mma<T>::run(acc[i][j], fa[i], fb[j]);
    }
    if (kt + 1 < KT) store_tile(cur ^ 1);
    __syncthreads();
  }

  // epilogue.  Accumulator register r of lane l holds
  // C[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31] of its 32x32 tile.
  TO* __restrict__ yg = reinterpret_cast<TO*>(a.y);
  const T* __restrict__ rg = reinterpret_cast<const T*>(a.res);
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = tile_n * BN + wn * WTN + j * 32 + l31;
    const bool n_ok = n < a.Cout;
    const float bias = (n_ok && a.bias) ? a.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = tile_m * BM + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (n_ok && m < a.M) {
          float v = acc[i][j][r] + bias;
          if (rg) v += ElemTraits<T>::load(rg + (size_t)m * a.res_ld + n);
          v = apply_act(v, a.act);
          ElemTraits<TO>::store(yg + (size_t)m * a.y_ld + n, v);
        }
      }
    }
  }
}

template <typename T, typename TO, int BM, int BN, int WAVES_M, int WAVES_N, int BKB>
static int launch_conv(ConvArgs& a, hipStream_t stream) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int LROW = BKB + 16;
  constexpr size_t smem = 2 * (size_t)(BM + BN) * LROW;
  a.tiles_m = ceil_div(a.M, BM);
  a.tiles_n = ceil_div(a.Cout, BN);
  const long long blocks = (long long)a.tiles_m * a.tiles_n;
  if (blocks <= 0 || blocks > 0x7fffffffLL) {
    set_error("fss_conv2d_nhwc: grid out of range");
    return 2;
  }
  auto kern = conv_igemm_kernel<T, TO, BM, BN, WAVES_M, WAVES_N, BKB>;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) {
      set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
      return 1;
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

template <typename T, typename TO>
static int dispatch_conv(ConvArgs& a, hipStream_t stream) {
  // Tile choice: widest N tile the layer fills; drop to 64-row tiles when the
  // 128-row grid would leave most of the 256 CUs idle.
  const int n_cols = a.Cout;
  if (n_cols <= 32) return launch_conv<T, TO, 128, 32, 4, 1, 64>(a, stream);
  if (n_cols <= 64) return launch_conv<T, TO, 128, 64, 2, 2, 64>(a, stream);
  const long long big = (long long)ceil_div(a.M, 128) * ceil_div(n_cols, 128);
  if (big >= 192) return launch_conv<T, TO, 128, 128, 2, 2, 64>(a, stream);
  return launch_conv<T, TO, 64, 64, 2, 2, 64>(a, stream);
}

}  // namespace fss

using namespace fss;

extern "C" int fss_conv_pack_dims(int cin, int kh, int kw, int cout, int dtype, int* k_pad,
                                  int* cout_pad) {
  FSS_REQUIRE(k_pad && cout_pad, "null argument");
  FSS_REQUIRE(dtype == FSS_F32 || dtype == FSS_BF16, "bad dtype");
  FSS_REQUIRE(cin > 0 && kh > 0 && kw > 0 && cout > 0, "bad shape");
  const int k = cin * kh * kw;
  *k_pad = (k + 63) / 64 * 64;          // a multiple of every K tile in use
  *cout_pad = (cout + 127) / 128 * 128; // a multiple of every N tile in use
  return 0;
}

extern "C" int fss_conv2d_nhwc(const fss_conv_desc* d, const void* x, const void* w_packed,
                               const float* bias, const void* residual, void* y, void* stream) {
  FSS_REQUIRE(d && x && w_packed && y, "null argument");
  FSS_REQUIRE(d->in_dtype == FSS_F32 || d->in_dtype == FSS_BF16, "bad in_dtype");
  FSS_REQUIRE(d->out_dtype == FSS_F32 || d->out_dtype == FSS_BF16, "bad out_dtype");
  FSS_REQUIRE(!(d->in_dtype == FSS_F32 && d->out_dtype == FSS_BF16),
              "f32 input with bf16 output is not supported");
  const int epc = d->in_dtype == FSS_F32 ? 4 : 8;
  FSS_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->cout > 0, "bad shape");
  FSS_REQUIRE(d->cin % epc == 0, "cin must be a multiple of 16 bytes of elements");
  FSS_REQUIRE(d->kh > 0 && d->kw > 0 && d->stride > 0 && d->dilation > 0, "bad filter");
  FSS_REQUIRE(d->ho > 0 && d->wo > 0, "bad output size");
  FSS_REQUIRE(d->pad_top >= 0 && d->pad_left >= 0, "negative padding");
  // every output pixel's first tap must start inside the padded image
  FSS_REQUIRE((long long)(d->ho - 1) * d->stride - d->pad_top < d->h &&
                  (long long)(d->wo - 1) * d->stride - d->pad_left < d->w,
              "output size exceeds the input");
  int k_pad = 0, cout_pad = 0;
  fss_conv_pack_dims(d->cin, d->kh, d->kw, d->cout, d->in_dtype, &k_pad, &cout_pad);
  FSS_REQUIRE(d->k_pad == k_pad && d->cout_pad == cout_pad, "packed weight dims mismatch");
  FSS_REQUIRE(d->y_ld >= d->cout, "y_ld < cout");
  FSS_REQUIRE(residual == nullptr || d->res_ld >= d->cout, "res_ld < cout");
  FSS_REQUIRE((long long)d->n * d->ho * d->wo < 0x7fffffffLL, "too many output pixels");
  FSS_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w_packed & 15) == 0, "unaligned pointer");

  ConvArgs a;
  a.x = x; a.w = w_packed; a.bias = bias; a.res = residual; a.y = y;
  a.N = d->n; a.H = d->h; a.W = d->w; a.Cin = d->cin; a.Cout = d->cout;
  a.KH = d->kh; a.KW = d->kw; a.stride = d->stride; a.dil = d->dilation;
  a.pad_t = d->pad_top; a.pad_l = d->pad_left; a.Ho = d->ho; a.Wo = d->wo; a.act = d->act;
  a.M = d->n * d->ho * d->wo;
  a.K = d->cin * d->kh * d->kw;
  a.Kpad = k_pad; a.y_ld = d->y_ld; a.res_ld = d->res_ld;
  a.tiles_m = a.tiles_n = 0;
  hipStream_t s = (hipStream_t)stream;
  if (d->in_dtype == FSS_F32) return dispatch_conv<float, float>(a, s);
  if (d->out_dtype == FSS_BF16) return dispatch_conv<bf16_t, bf16_t>(a, s);
  return dispatch_conv<bf16_t, float>(a, s);
}
