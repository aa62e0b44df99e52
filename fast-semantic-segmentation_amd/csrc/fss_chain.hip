// Two 1x1 convolutions in a row, the first with a residual, in one streaming kernel
// (gfx950, bf16):
//
//   y  = act1(conv1x1([x | x2], W) + bias + residual)        cout1 channels, stored
//   y2 = act2(conv1x1(bf16(y), W2) + bias2)                  cout2 channels, stored
//
// This is the end of one ResNet bottleneck unit and the beginning of the next
// (third_party/dilated_resnet_v1.py:50-113 in the reference: conv3 (+ projection shortcut)
// + add + relu, then the next unit's conv1), in the first block, where the filters are tiny
// (K <= 128, cout1 <= 256, cout2 <= 64) and both layers are bound by the bytes of the
// cout1-channel tensor: run one after the other it is written once and read twice; here
// the second layer takes it from the accumulator registers of the first.
//
// No activation ever touches LDS.  Both products run on the matrix cores with the weights
// as the MFMA "A" operand (from LDS, loaded once per block) and 16 pixels as the "B"
// operand: D[channel][pixel].  A wave owns 16 pixels per pass: lane (r, g) loads the 16
// bytes of channels 32 s + 8 g .. + 7 of pixel r straight from global memory into the B
// operand of k step s.  The weight rows of a 16-row tile are ordered so that the 8
// accumulator values a lane owns in the tile pair (2 u, 2 u + 1) are the 8 CONSECUTIVE
// channels 32 u + 8 g .. + 7 of its pixel: the residual is one 16-byte load, the output one
// 16-byte store, and, rounded to bf16, those 8 values are exactly the lane's B operand of k
// step u of the second product.  Blocks are persistent; each wave has the pixels and the
// residual of its next two passes in flight while it computes the current one.
#include <algorithm>

#include "fss_common.h"

namespace fss {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

struct ChainArgs {
  const bf16_t* x;
  const bf16_t* x2;
  const bf16_t* w;
  const float* bias;
  const bf16_t* res;
  bf16_t* y;
  const bf16_t* w2;
  const float* bias2;
  bf16_t* y2;
  long long M;
  int ks1;                      // k steps (of 32 channels) that come from x; the rest from x2
  int x_ld, x2_ld, res_ld, y_ld, y2_ld;
  int act1, act2;
};

constexpr int kChainWaves = 8;                  // waves per block
constexpr int kChainPixels = 16 * kChainWaves;  // pixels per block and pass

// LDS map: first filter, second filter, the two bias vectors.  A filter is stored fragment by
// fragment: the 16 rows x 64 bytes that one MFMA step of one 16-row tile reads are 1 KiB of
// consecutive bytes (row pitch 64), and the 16-byte chunk c of row rho sits at chunk
// c ^ ((0 - (rho >> 2)) & 3), which spreads the 16 lanes of every ds_read_b128 service group
// over all banks.
template <int KS, int C1P, int C2P> struct ChainLds {
  static constexpr int OFF_W2 = 2 * C1P * KS * 1024;
  static constexpr int OFF_B = OFF_W2 + 2 * C2P * C1P * 1024;
  // per wave: 16 pixels x 128 channels of the first output on their way to memory (row pitch
  // 256 + 16 bytes)
  static constexpr int STG_ROW = 272, STG_WAVE = 16 * STG_ROW;
  static constexpr int OFF_STG = OFF_B + 32 * (C1P + C2P) * 4;
  static constexpr int BYTES = OFF_STG + kChainWaves * STG_WAVE;
};

// byte offset, inside a filter image of KSTEPS k steps per tile, of chunk c (8 k values) of row
// rho of tile j
__device__ __forceinline__ int chain_lds_offset(int j, int rho, int c, int ksteps) {
  return ((j * ksteps + (c >> 2)) * 16 + rho) * 64 + (((c & 3) ^ ((0 - (rho >> 2)) & 3)) << 4);
}

// channel held by row rho of 16-row tile j (see the header)
__host__ __device__ inline int chain_channel(int j, int rho) {
  return 32 * (j >> 1) + 8 * (rho >> 2) + 4 * (j & 1) + (rho & 3);
}

// KS = (cin + cin2) / 32, C1P = cout1 / 32, C2P = cout2 / 32; KS1T = cin / 32 and REST = "there is
// a residual" when known at compile time, -1 = taken from the arguments.  The loop body must be
// free of branches around its loads and stores: the compiler counts the memory operations in
// flight per path, and where paths with different counts join it waits for all of them, which
// would serialise every pass behind the loads of the passes after it.  Hence: the shapes the
// models use are compile-time variants; loads past the last pass re-read the last pass;
// pixels past the end stand in for the last pixel and store its (identical) values again.
template <int KS, int C1P, int C2P, int KS1T, int REST>
__global__ void __launch_bounds__(kChainWaves * 64) conv1x1_chain_kernel(const ChainArgs a) {
  typedef ChainLds<KS, C1P, C2P> L;
  constexpr int K = 32 * KS, C1 = 32 * C1P, C2 = 32 * C2P;
  constexpr int NT = kChainWaves * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wl = smem;
  char* w2l = smem + L::OFF_W2;
  float* bl = reinterpret_cast<float*>(smem + L::OFF_B);

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int r = lane & 15, g = lane >> 4;

  for (int idx = tid; idx < C1 * (K / 8); idx += NT) {
    const int row = idx / (K / 8), c = idx - row * (K / 8);
    *reinterpret_cast<uint4*>(wl + chain_lds_offset(row >> 4, row & 15, c, KS)) =
        *reinterpret_cast<const uint4*>(a.w + (size_t)chain_channel(row >> 4, row & 15) * K + c * 8);
  }
  for (int idx = tid; idx < C2 * (C1 / 8); idx += NT) {
    const int row = idx / (C1 / 8), c = idx - row * (C1 / 8);
    *reinterpret_cast<uint4*>(w2l + chain_lds_offset(row >> 4, row & 15, c, C1P)) =
        *reinterpret_cast<const uint4*>(a.w2 + (size_t)chain_channel(row >> 4, row & 15) * C1 + c * 8);
  }
  for (int idx = tid; idx < C1 + C2; idx += NT)
    bl[idx] = idx < C1 ? (a.bias != nullptr ? a.bias[idx] : 0.f) : (a.bias2 != nullptr ? a.bias2[idx - C1] : 0.f);
  __syncthreads();

  const float lo1 = a.act1 == FSS_ACT_NONE ? -INFINITY : 0.f, hi1 = a.act1 == FSS_ACT_RELU6 ? 6.f : INFINITY;
  const float lo2 = a.act2 == FSS_ACT_NONE ? -INFINITY : 0.f, hi2 = a.act2 == FSS_ACT_RELU6 ? 6.f : INFINITY;
  // this lane's fragment of weight row r, k group g (tile and k step are immediate offsets)
  const char* wfrag = wl + chain_lds_offset(0, r, g, KS);
  const char* w2frag = w2l + chain_lds_offset(0, r, g, C1P);

  // The residual enters through the matrix cores as one more k step against a 0/1 matrix
  // (exact: bf16 values times 1.0 into float accumulators), which saves the conversion and
  // the additions on the vector ALU.  idw[j]: this lane's fragment of the 16 x 32 selection
  // matrix of tile j of a pair, whose row rho picks k = 8 (rho >> 2) + 4 j + (rho & 3): the
  // position, among the 32 channels of the pair, of the channel that row accumulates.
  uint4 idw[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int e = 4 * j + (r & 3);
    unsigned wsel[4] = {0u, 0u, 0u, 0u};
    if (g == (r >> 2)) wsel[e >> 1] = 0x3f80u << (16 * (e & 1));
    idw[j] = make_uint4(wsel[0], wsel[1], wsel[2], wsel[3]);
  }
  const bool has_res = REST > 0 || (REST < 0 && a.res != nullptr);
  // The first output leaves through a wave-private LDS slab, 128 channels at a time: a lane
  // holds 16 bytes of each of 8 pixel rows' 64-byte pieces, and stored as they are the pieces
  // of one instruction would be 64 bytes each; read back row-major, one instruction writes 4
  // pixels x 256 consecutive bytes (measured: 0.91 -> 0.80 ms and 1.16 -> 1.09 ms per 128
  // frames with contiguous stores).
  char* stg = smem + L::OFF_STG + wave * L::STG_WAVE;
  char* stg_w = stg + r * L::STG_ROW + g * 16;                    // + 64 * (pair & 3)
  const char* stg_r = stg + (lane >> 4) * L::STG_ROW + (lane & 15) * 16;   // + 4 * j rows

  const long long passes = (a.M + kChainPixels - 1) / kChainPixels;
  // pixels and residual of pass `pass`, for this lane: pixel r of the wave's 16, channel group g
  auto fetch = [&](long long pass, uint4 (&xb)[KS], uint4 (&rb)[C1P]) {
    pass = pass < passes ? pass : passes - 1;
    long long p = pass * kChainPixels + wave * 16 + r;
    p = p < a.M ? p : a.M - 1;
    const bf16_t* px = a.x + (size_t)p * a.x_ld + 8 * g;
    if constexpr (KS1T >= 0) {
      const bf16_t* px2 = a.x2 + (size_t)p * a.x2_ld + 8 * g;       // not used when KS1T == KS
#pragma unroll
      for (int s = 0; s < KS; ++s)
        xb[s] = *reinterpret_cast<const uint4*>(s < KS1T ? px + 32 * s : px2 + 32 * (s - KS1T));
    } else {
      const bf16_t* px2 = a.x2 != nullptr ? a.x2 + (size_t)p * a.x2_ld + 8 * g : px;
#pragma unroll
      for (int s = 0; s < KS; ++s)
        xb[s] = *reinterpret_cast<const uint4*>(s < a.ks1 ? px + 32 * s : px2 + 32 * (s - a.ks1));
    }
    if (has_res) {
      const bf16_t* pr = a.res + (size_t)p * a.res_ld + 8 * g;
#pragma unroll
      for (int u = 0; u < C1P; ++u) rb[u] = *reinterpret_cast<const uint4*>(pr + 32 * u);
    } else {
#pragma unroll
      for (int u = 0; u < C1P; ++u) rb[u] = make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto compute = [&](long long pass, const uint4 (&xb)[KS], const uint4 (&rb)[C1P]) {
    long long p = pass * kChainPixels + wave * 16 + r;
    p = p < a.M ? p : a.M - 1;       // past the end: the last pixel again, same values
    f32x4 acc2[2 * C2P];
#pragma unroll
    for (int j = 0; j < 2 * C2P; ++j) {
      const float4 b = *reinterpret_cast<const float4*>(bl + C1 + 32 * (j >> 1) + 8 * g + 4 * (j & 1));
      acc2[j] = f32x4{b.x, b.y, b.z, b.w};
    }
    // rows of the slab are the wave's 16 consecutive pixels; row 4 j + (lane >> 4) is stored
    // by this lane (past the end: the last pixel again, whose values those rows repeat)
    const long long pw = pass * kChainPixels + wave * 16 + (lane >> 4);
    bf16_t* py[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long pj = pw + 4 * j < a.M ? pw + 4 * j : a.M - 1;
      py[j] = a.y + (size_t)pj * a.y_ld + 8 * (lane & 15);
    }
    // One tile pair (32 output channels) at a time: its two accumulators, its epilogue, its k
    // step of the second product.  Only the second product's accumulators live across pairs.
#pragma unroll
    for (int u = 0; u < C1P; ++u) {
      // the accumulators start from the biases of the lane's 8 channels
      const float4 b0 = *reinterpret_cast<const float4*>(bl + 32 * u + 8 * g);
      const float4 b1 = *reinterpret_cast<const float4*>(bl + 32 * u + 8 * g + 4);
      f32x4 acc[2] = {f32x4{b0.x, b0.y, b0.z, b0.w}, f32x4{b1.x, b1.y, b1.z, b1.w}};
      if (has_res) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, idw[j]),
                                                           __builtin_bit_cast(bf16x8_t, rb[u]), acc[j], 0, 0, 0);
      }
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const uint4 wa = *reinterpret_cast<const uint4*>(wfrag + ((2 * u + j) * KS + s) * 1024);
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wa),
                                                           __builtin_bit_cast(bf16x8_t, xb[s]), acc[j], 0, 0, 0);
        }
      float v[8] = {acc[0][0], acc[0][1], acc[0][2], acc[0][3], acc[1][0], acc[1][1], acc[1][2], acc[1][3]};
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], lo1, hi1);
      const uint4 yb = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                  pack_bf16x2(v[6], v[7]));
      *reinterpret_cast<uint4*>(stg_w + 64 * (u & 3)) = yb;
      if ((u & 3) == 3) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint4 t = *reinterpret_cast<const uint4*>(stg_r + 4 * j * L::STG_ROW);
          store16_nt(py[j] + 128 * (u >> 2), t.x, t.y, t.z, t.w);
        }
      }
#pragma unroll
      for (int j = 0; j < 2 * C2P; ++j) {
        const uint4 wa = *reinterpret_cast<const uint4*>(w2frag + (j * C1P + u) * 1024);
        acc2[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wa),
                                                          __builtin_bit_cast(bf16x8_t, yb), acc2[j], 0, 0, 0);
      }
      // keep the weight reads of later pairs from being hoisted above this point (they would
      // all be live at once)
      if (u % 4 == 3) __builtin_amdgcn_sched_barrier(0);
    }
    bf16_t* py2 = a.y2 + (size_t)p * a.y2_ld + 8 * g;
#pragma unroll
    for (int u = 0; u < C2P; ++u) {
      float v[8] = {acc2[2 * u][0], acc2[2 * u][1], acc2[2 * u][2], acc2[2 * u][3],
                    acc2[2 * u + 1][0], acc2[2 * u + 1][1], acc2[2 * u + 1][2], acc2[2 * u + 1][3]};
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], lo2, hi2);
      store16_nt(py2 + 32 * u, pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                 pack_bf16x2(v[6], v[7]));
    }
  };

  // Three register sets in rotation: while one pass is computed, the loads of the next two are
  // in flight.  Three passes per trip, so that the sets swap roles without copies.
  const long long step = gridDim.x;
  long long pass = blockIdx.x;
  if (pass >= passes) return;
  uint4 xa[KS], ra[C1P], xb[KS], rb[C1P], xc[KS], rc[C1P];
  fetch(pass, xa, ra);
  fetch(pass + step, xb, rb);
  while (true) {
    fetch(pass + 2 * step, xc, rc);
    compute(pass, xa, ra);
    pass += step;
    if (pass >= passes) break;
    fetch(pass + 2 * step, xa, ra);
    compute(pass, xb, rb);
    pass += step;
    if (pass >= passes) break;
    fetch(pass + 2 * step, xb, rb);
    compute(pass, xc, rc);
    pass += step;
    if (pass >= passes) break;
  }
}

template <int KS, int C1P, int C2P, int KS1T = -1, int REST = -1>
static int launch_chain(const ChainArgs& a, hipStream_t stream) {
  constexpr size_t smem = ChainLds<KS, C1P, C2P>::BYTES;
  static_assert(smem <= 160 * 1024, "filters too large for LDS");
  auto kern = conv1x1_chain_kernel<KS, C1P, C2P, KS1T, REST>;
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
      set_error("fss_conv1x1_chain_nhwc: cannot query the device");
      return 1;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern),
                                                     kChainWaves * 64, smem) != hipSuccess || per_cu < 1)
      per_cu = 1;
    num_cu = prop.multiProcessorCount;
  }
  const long long passes = (a.M + kChainPixels - 1) / kChainPixels;
  const int blocks = (int)std::min<long long>(passes, (long long)num_cu * per_cu);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kChainWaves * 64), smem, stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error(std::string("fss_conv1x1_chain_nhwc launch: ") + hipGetErrorString(e));
    return 3;
  }
  return 0;
}

}  // namespace fss

using namespace fss;

extern "C" int fss_conv1x1_chain_nhwc(const void* x, int cin, int x_ld, const void* x2, int cin2, int x2_ld,
                                      const void* w, const float* bias, const void* residual, int res_ld,
                                      int act1, void* y, int y_ld, int cout1, const void* w2,
                                      const float* bias2, int act2, void* y2, int y2_ld, int cout2,
                                      long long pixels, int dtype, void* stream) {
  FSS_REQUIRE(x && w && y && w2 && y2, "null argument");
  FSS_REQUIRE(dtype == FSS_BF16, "only bfloat16 is supported");
  FSS_REQUIRE(pixels > 0, "bad shape");
  FSS_REQUIRE(cin > 0 && cin % 32 == 0 && cin2 >= 0 && cin2 % 32 == 0 && (cin2 == 0) == (x2 == nullptr),
              "cin and cin2 must be multiples of 32 (cin2 = 0 without a second input)");
  const int k = cin + cin2;
  FSS_REQUIRE(k == 32 || k == 64 || k == 128, "cin + cin2 must be 32, 64 or 128");
  FSS_REQUIRE(cout1 == 128 || cout1 == 256, "cout1 must be 128 or 256");
  FSS_REQUIRE(cout2 == 32 || cout2 == 64, "cout2 must be 32 or 64");
  FSS_REQUIRE(x_ld >= cin && x_ld % 8 == 0 && (x2 == nullptr || (x2_ld >= cin2 && x2_ld % 8 == 0)) &&
                  y_ld >= cout1 && y_ld % 8 == 0 && y2_ld >= cout2 && y2_ld % 8 == 0 &&
                  (residual == nullptr || (res_ld >= cout1 && res_ld % 8 == 0)),
              "pixel strides must be multiples of 8 and at least the channel count");
  for (const void* p : {x, x2, w, residual, (const void*)y, w2, (const void*)y2})
    FSS_REQUIRE(((uintptr_t)p & 15) == 0, "unaligned pointer");
  for (int act : {act1, act2})
    FSS_REQUIRE(act == FSS_ACT_NONE || act == FSS_ACT_RELU || act == FSS_ACT_RELU6, "bad act");
  ChainArgs a;
  a.x = (const bf16_t*)x; a.x2 = (const bf16_t*)x2; a.w = (const bf16_t*)w; a.bias = bias;
  a.res = (const bf16_t*)residual; a.y = (bf16_t*)y; a.w2 = (const bf16_t*)w2; a.bias2 = bias2;
  a.y2 = (bf16_t*)y2; a.M = pixels; a.ks1 = cin / 32;
  a.x_ld = x_ld; a.x2_ld = x2_ld; a.res_ld = res_ld; a.y_ld = y_ld; a.y2_ld = y2_ld;
  a.act1 = act1; a.act2 = act2;
  hipStream_t s = (hipStream_t)stream;
  // the shapes of the first ResNet block (full- and half-width models) as compile-time variants
  const int res = residual != nullptr;
  if (cout1 == 256 && cout2 == 64 && cin == 64 && cin2 == 0 && res) return launch_chain<2, 8, 2, 2, 1>(a, s);
  if (cout1 == 256 && cout2 == 64 && cin == 64 && cin2 == 64 && !res) return launch_chain<4, 8, 2, 2, 0>(a, s);
  if (cout1 == 128 && cout2 == 32 && cin == 32 && cin2 == 0 && res) return launch_chain<1, 4, 1, 1, 1>(a, s);
  if (cout1 == 128 && cout2 == 32 && cin == 32 && cin2 == 32 && !res) return launch_chain<2, 4, 1, 1, 0>(a, s);
  switch ((k / 32) * 100 + (cout1 / 32) * 10 + cout2 / 32) {
    case 141: return launch_chain<1, 4, 1>(a, s);
    case 142: return launch_chain<1, 4, 2>(a, s);
    case 241: return launch_chain<2, 4, 1>(a, s);
    case 242: return launch_chain<2, 4, 2>(a, s);
    case 281: return launch_chain<2, 8, 1>(a, s);
    case 282: return launch_chain<2, 8, 2>(a, s);
    case 441: return launch_chain<4, 4, 1>(a, s);
    case 442: return launch_chain<4, 4, 2>(a, s);
    case 481: return launch_chain<4, 8, 1>(a, s);
    case 482: return launch_chain<4, 8, 2>(a, s);
    case 181: return launch_chain<1, 8, 1>(a, s);
    default: return launch_chain<1, 8, 2>(a, s);
  }
}
