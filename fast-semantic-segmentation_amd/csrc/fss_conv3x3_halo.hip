// 3x3 convolution, stride 1, dilation 1, 128 -> 256 channels, with the input tile and its halo
// resident in LDS (gfx950; bfloat16 or half precision, template parameter H):
//
//   y = act(conv3x3(x, W) + bias),   cin = 128, cout = 256, one pixel of zero padding all round
//
// This is `Conv_3` of ICNet's full-resolution branch (architectures/icnet_architecture.py:223-225
// in the reference).  With K = 1152 the implicit-GEMM kernel spends as long outside its main loop
// as inside it (one block per CU at 256 x 256: ring fill, epilogue and block turnover are exposed;
// two blocks per CU at 128 x 256: twice the weight bytes through the global -> LDS path, which
// bounds the loop) and stages every input byte nine times.  Here
//   * a block is persistent and owns the CU; per tile of 4 x 64 output pixels (or a ragged band
//     at the right edge, see halo_tile_of) the 6 x 66 input pixels it needs are copied to LDS ONCE and
//     all nine taps read them there at shifted addresses;
//   * the filter streams through a ring of three 16 KiB stages (one tap, 32 input channels, all
//     256 filters) that never stops at a tile boundary: the stages of a tile's first two steps are
//     in flight while the previous tile finishes.  The host packs the filter as the exact LDS image
//     of the 36 stages, so every DMA piece is 1 KiB of consecutive global bytes;
//   * products run as D[channel][pixel] = W[channel][k] * X[k][pixel] with the rows of a weight
//     tile ordered so that a lane ends up with 8 consecutive channels of its pixel per tile pair:
//     the results go from the accumulators to memory as 16-byte stores, no staging through LDS,
//     and the next tile's input is already on its way while they are written.
// The k axis is walked in the order of the implicit GEMM (tap-major, then channel), with the same
// MFMA instruction, so the results agree with it bit for bit.
#include <algorithm>
#include <type_traits>

#include "fss_conv_args.h"

namespace fss {

constexpr int kHWaves = 8;                           // waves per block: 4 (pixels) x 2 (channels)
constexpr int kHThreads = kHWaves * 64;
constexpr int kHCin = 128, kHCout = 256;
constexpr int kHTW = 64, kHTH = 4;                   // regular tile: 4 rows of 64 pixels
constexpr int kHPatchPx = (kHTH + 2) * (kHTW + 2);   // 396 input pixels
constexpr int kHPixBytes = kHCin * 2;                // one pixel in LDS: 256 bytes = all 64 banks
constexpr int kHPatchBytes = kHPatchPx * kHPixBytes; // 99 KiB
constexpr int kHStageBytes = kHCout * 64;            // 16 KiB: 256 filters x 32 k values
constexpr int kHSlots = 3;
constexpr int kHStages = 9 * (kHCin / 32);           // 36 per tile
constexpr int kHOffRing = kHPatchBytes;
constexpr int kHOffBias = kHOffRing + kHSlots * kHStageBytes;
constexpr int kHSmem = kHOffBias + kHCout * 4;
static_assert(kHPatchBytes % 1024 == 0, "the patch is filled in 1 KiB pieces");
static_assert(kHSmem <= 160 * 1024, "LDS budget");

struct HaloArgs {
  const char* x;        // [N][H][W][x_ld] 16-bit
  const char* w;        // 36 stage images of 16 KiB (see fss_conv3x3_c128_pack_index)
  const float* bias;    // [256] or null
  char* y;              // [N][H][W][y_ld] 16-bit
  int N, H, W, x_ld, y_ld, act;
  int bands, row_tiles;             // full-width bands (W / 64) and tiles per band (ceil(H / 4))
  int rag_w, rag_h;                 // ragged band at the right edge: its width (0: none), rows per tile
  int tiles_per_frame;
  int tiles;
};

// Tile t of the launch: frame, first output row / column, rows and columns (wave-uniform).
// A frame is cut into bands of 64 columns, each into tiles of 4 rows; the W % 64 columns left at
// the right edge form one more band whose tiles are as tall as 256 pixels and the patch allow
// (a single column of 129 rows is ONE tile).
struct HaloTile { int n, y0, x0, th, tw; };
__device__ __forceinline__ HaloTile halo_tile_of(const HaloArgs& a, int t) {
  HaloTile T;
  T.n = t / a.tiles_per_frame;
  int k = t - T.n * a.tiles_per_frame;
  const int regular = a.bands * a.row_tiles;
  if (k < regular) {
    const int band = k / a.row_tiles, rt = k - band * a.row_tiles;
    T.x0 = band * kHTW;
    T.y0 = rt * kHTH;
    T.tw = kHTW;
    T.th = min(kHTH, a.H - T.y0);
  } else {
    k -= regular;
    T.x0 = a.bands * kHTW;
    T.y0 = k * a.rag_h;
    T.tw = a.rag_w;
    T.th = min(a.rag_h, a.H - T.y0);
  }
  T.n = __builtin_amdgcn_readfirstlane(T.n);
  T.y0 = __builtin_amdgcn_readfirstlane(T.y0);
  T.x0 = __builtin_amdgcn_readfirstlane(T.x0);
  T.th = __builtin_amdgcn_readfirstlane(T.th);
  T.tw = __builtin_amdgcn_readfirstlane(T.tw);
  return T;
}

// 16-byte window of patch pixel p that holds its logical chunk c (8 channels): ds_read_b128 serves
// 8 lanes of one k-group (8 consecutive pixels, the same chunk) together with 8 lanes of the next
// k-group (8 other pixels, the chunk with bit 0 flipped).  Bits 1-3 of the window come from the
// pixel, bit 0 from the chunk alone, so the 16 windows of a service group are all different
// whatever the tap's shift.
__device__ __forceinline__ int halo_window(int p, int c) { return c ^ ((p & 7) << 1); }

template <typename H>
__global__ void __launch_bounds__(kHThreads) conv3x3_c128_halo_kernel(const HaloArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int r = lane & 15, g = lane >> 4;
  const int wp = wave >> 1, wc = wave & 1;           // pixel quarter, channel half of the tile

  float* bl = reinterpret_cast<float*>(smem + kHOffBias);
  for (int i = tid; i < kHCout; i += kHThreads) bl[i] = a.bias != nullptr ? a.bias[i] : 0.f;
  __syncthreads();

  const float lo = a.act == FSS_ACT_NONE ? -INFINITY : 0.f;
  const float hi = a.act == FSS_ACT_RELU6 ? 6.f : INFINITY;
  // half-precision outputs saturate instead of overflowing
  constexpr bool kHalf = std::is_same<H, f16_t>::value;
  const float out_lo = kHalf ? fmaxf(lo, -65504.f) : lo, out_hi = kHalf ? fminf(hi, 65504.f) : hi;

  // ---- filter ring: stage n of the block's stream sits in slot n % 3; every wave copies 2 KiB
  const fss_i32x4 rs_w = dma_resource(a.w, (unsigned long long)kHStages * kHStageBytes);
  const unsigned w_lane = (unsigned)(wave * 2048 + lane * 16);
  auto issue_w = [&](int stage, int slot) {          // stage in [0, 36), slot in [0, 3)
    const unsigned dst = smem_base + kHOffRing + slot * kHStageBytes + wave * 2048;
    blds16(w_lane + (unsigned)(stage * kHStageBytes), rs_w, dst);
    blds16(w_lane + (unsigned)(stage * kHStageBytes + 1024), rs_w, dst + 1024);
  };
  constexpr int kWPieces = 2;                        // DMA pieces per wave and stage
  // this lane's weight fragment of tile jj of its channel half, k step of a stage:
  // row r, chunk g of the 1 KiB image of that tile (chunks XOR-swizzled with the row)
  const char* wfrag = smem + kHOffRing + ((8 * wc) * 16 + r) * 64 + ((g ^ ((0 - (r >> 2)) & 3)) << 4);

  // ---- patch fill, row by row: piece j of patch row py = its pixels 4 j .. 4 j + 3 (1 KiB); lane ->
  // pixel 4 j + lane / 16, window lane % 16.  Lanes past the end of the row stay out of the copy
  // (their bytes belong to the next row's first piece).
  const int frame_bytes_i = a.H * a.W * a.x_ld * 2;   // < 2^31 (host)
  auto issue_patch = [&](const HaloTile& T) {
    const int PW = T.tw + 2, PH = T.th + 2;
    const int ppr = (PW + 3) >> 2;                    // pieces per row
    const fss_i32x4 rs_x = dma_resource(a.x + (size_t)T.n * (unsigned)frame_bytes_i, (unsigned)frame_bytes_i);
    int py = 0, j = wave;                             // this wave's pieces: every 8th
    while (j >= ppr) { j -= ppr; ++py; }
    const int sub = lane >> 4;
    while (py < PH) {
      const int px = 4 * j + sub;
      const int iy = T.y0 - 1 + py, ix = T.x0 - 1 + px;
      const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      const int p = py * PW + px;
      const int c = halo_window(p, lane & 15);       // (the map is its own inverse)
      const unsigned off = (unsigned)((iy * a.W + ix) * a.x_ld * 2 + c * 16);
      if (px < PW) blds16(ok ? off : kDmaOutOfRange, rs_x, smem_base + (py * PW + 4 * j) * kHPixBytes);
      j += kHWaves;
      while (j >= ppr) { j -= ppr; ++py; }
    }
  };

  const int step = gridDim.x;
  int tile = blockIdx.x;
  if (tile >= a.tiles) return;
  HaloTile T = halo_tile_of(a, tile);
  issue_w(0, 0);
  issue_w(1, 1);
  issue_w(2, 2);
  issue_patch(T);
  int stores_in_flight = 0;                          // of the previous tile, per wave (wave-uniform)

  typedef fss_f32x4 acc_t;
  while (true) {
    // The patch of T and stages 0-2 have landed once nothing older than the previous tile's stores
    // is outstanding (the memory pipe completes a wave's operations in issue order): the stores
    // themselves drain under the first stages.
    switch (stores_in_flight) {
      case 16: wait_vmcnt<16>(); break;
      case 12: wait_vmcnt<12>(); break;
      case 8: wait_vmcnt<8>(); break;
      case 4: wait_vmcnt<4>(); break;
      default: wait_vmcnt<0>(); break;
    }
    __builtin_amdgcn_s_barrier();

    // this lane's pixel of group i (16 pixels each, four groups per wave): byte address, window
    // bits apart, of its patch pixel under tap (0, 0); and its place in the output frame
    // (idx / tw through the reciprocal: exact for idx < 256, tw <= 64)
    const int PW = T.tw + 2, npix = T.th * T.tw;
    const float inv_tw = __builtin_amdgcn_rcpf((float)T.tw);
    int p0x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = wp * 64 + i * 16 + r;
      const int ty = (int)(((float)idx + 0.5f) * inv_tw), tx = idx - ty * T.tw;
      p0x[i] = idx < npix ? (ty * PW + tx) * kHPixBytes : 0;
      asm volatile("" : "+v"(p0x[i]));
    }
    acc_t acc[8][4];                                 // [weight tile jj][pixel group i]

    // Stage t = one tap (t / 4), 32 input channels (t % 4).  The fragments of stage t + 1 are read
    // from LDS while the MFMAs of stage t run (two register sets): between two barriers the LDS
    // and the matrix cores work at the same time instead of one after the other.
    uint4 wf[2][8], pf[2][4];
    int base[4];
    auto prepare = [&](auto t_c) {
      constexpr int t = decltype(t_c)::value;
      constexpr int kr = t / 12, ks = (t / 4) % 3, q = t & 3;
      if constexpr (t > 0) {
        // this wave's pieces of stage t have landed (stages 0-2: see above) ...
        if constexpr (t > 2) wait_vmcnt<kWPieces>();
        __builtin_amdgcn_s_barrier();                // ... and everybody's; every wave has its stage t - 1
        issue_w((t + 2) % kHStages, (t + 2) % kHSlots);   // fragments in registers: that slot is free
      }
      if constexpr (q == 0) {
        // byte address of channel quarter 0 of the lane's pixels under this tap; the other quarters
        // are this address with bits 6-7 flipped (the window map is XOR-linear)
        const int d256 = (kr * PW + ks) * kHPixBytes;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int tt = p0x[i] + d256;              // pixel p = tt / 256; window = g ^ ((p & 7) << 1)
          base[i] = tt + ((g << 4) ^ ((tt >> 3) & 0xe0));
          asm volatile("" : "+v"(base[i]));
        }
      }
#pragma unroll
      for (int jj = 0; jj < 8; ++jj)
        wf[t & 1][jj] = *reinterpret_cast<const uint4*>(wfrag + (t % kHSlots) * kHStageBytes + jj * 1024);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        pf[t & 1][i] = *reinterpret_cast<const uint4*>(smem + (base[i] ^ (q << 6)));
    };
    prepare(std::integral_constant<int, 0>{});
    wave_lds_fence();
    auto stage = [&](auto t_c) {
      constexpr int t = decltype(t_c)::value;
      if constexpr (t + 1 < kHStages) prepare(std::integral_constant<int, t + 1>{});
#pragma unroll
      for (int jj = 0; jj < 8; ++jj)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[jj][i] = mfma_16x16x32<H>(wf[t & 1][jj], pf[t & 1][i], t == 0 ? acc_t{0.f, 0.f, 0.f, 0.f} : acc[jj][i]);
      // one fragment read, then two MFMAs: a wave that pushed its twelve reads at once would sit
      // behind the other waves' reads in the LDS queue before its first MFMA could issue
      if constexpr (t + 1 < kHStages) {
#pragma unroll
        for (int k = 0; k < 12; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      wave_lds_fence();                              // the fragments of stage t + 1 are in registers
    };
    static_for<0, kHStages>(stage);

    const int next = tile + step;
    const bool more = next < a.tiles;
    __builtin_amdgcn_s_barrier();                    // everybody is done with the patch and slot 2
    HaloTile Tn = T;
    if (more) {
      issue_w(2, 2);                                 // (stages 0 and 1 went out during stages 34 and 35)
      Tn = halo_tile_of(a, next);
      issue_patch(Tn);                               // lands while the results below are written
    }
    // results: lane (r, g) holds, for pixel r of group i, channels 128 wc + 32 u + 8 g .. + 7 in the
    // tile pair (2 u, 2 u + 1)
    // (the bias is added to the finished sums, as the implicit GEMM does: same bits)
    char* yframe = a.y + (size_t)T.n * a.H * a.W * a.y_ld * 2 + (128 * wc + 8 * g) * 2;
    float4 bv[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj)
      bv[jj] = *reinterpret_cast<const float4*>(bl + 128 * wc + 32 * (jj >> 1) + 8 * g + 4 * (jj & 1));
    // groups that hold pixels of the tile (wave-uniform): four stores each, the count the next
    // pass of the loop leaves in flight
    const int groups = min(4, max(0, (npix - wp * 64 + 15) >> 4));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < groups) {
        const int idx = wp * 64 + i * 16 + r;
        const int ty = (int)(((float)idx + 0.5f) * inv_tw), tx = idx - ty * T.tw;
        char* py = yframe + ((size_t)(T.y0 + ty) * a.W + (T.x0 + tx)) * (size_t)(a.y_ld * 2);
        const bool mine = idx < npix;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float4 b0 = bv[2 * u], b1 = bv[2 * u + 1];
          float v[8] = {acc[2 * u][i][0] + b0.x, acc[2 * u][i][1] + b0.y, acc[2 * u][i][2] + b0.z,
                        acc[2 * u][i][3] + b0.w, acc[2 * u + 1][i][0] + b1.x, acc[2 * u + 1][i][1] + b1.y,
                        acc[2 * u + 1][i][2] + b1.z, acc[2 * u + 1][i][3] + b1.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], out_lo, out_hi);
          const uint4 o = make_uint4(Half16<H>::pack2(v[0], v[1]), Half16<H>::pack2(v[2], v[3]),
                                     Half16<H>::pack2(v[4], v[5]), Half16<H>::pack2(v[6], v[7]));
          // (lane 0 of the group always holds a pixel of the tile: the store is issued, and counted)
          if (mine) *reinterpret_cast<uint4*>(py + 64 * u) = o;
        }
      }
    }
    if (!more) break;
    stores_in_flight = 4 * groups;
    T = Tn;
    tile = next;
  }
  // the ring ran two stages ahead: nothing may still be writing into LDS when the block ends
  wait_vmcnt<0>();
}

template <typename H>
static int launch_halo(const HaloArgs& a, hipStream_t stream) {
  auto kern = conv3x3_c128_halo_kernel<H>;
  static bool attr_set = false;   // per instantiation
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kHSmem);
    if (e != hipSuccess) {
      set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
      return 1;
    }
    attr_set = true;
  }
  static int num_cu = 0;
  if (num_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      set_error("fss_conv3x3_c128_nhwc: cannot query the device");
      return 1;
    }
    num_cu = prop.multiProcessorCount;
  }
  // persistent blocks, one per CU (the block's LDS and registers fill it)
  const int blocks = std::min(a.tiles, num_cu);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kHThreads), kHSmem, stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error(std::string("fss_conv3x3_c128_nhwc launch: ") + hipGetErrorString(e));
    return 3;
  }
  return 0;
}

}  // namespace fss

using namespace fss;

extern "C" int fss_conv3x3_c128_pack_index(int32_t* index) {
  FSS_REQUIRE(index != nullptr, "null argument");
  // element e of the packed filter = W[cout = index[2 e]][k = index[2 e + 1]], k = tap * 128 + channel
  for (int s = 0; s < kHStages; ++s)
    for (int j = 0; j < 16; ++j)
      for (int rho = 0; rho < 16; ++rho)
        for (int cp = 0; cp < 4; ++cp)
          for (int e = 0; e < 8; ++e) {
            const int c = cp ^ ((0 - (rho >> 2)) & 3);                 // logical chunk in window cp
            const int jj = j & 3;
            const int channel = 64 * (j >> 2) + 32 * (jj >> 1) + 8 * (rho >> 2) + 4 * (jj & 1) + (rho & 3);
            const int k = (s >> 2) * kHCin + 32 * (s & 3) + 8 * c + e;
            const size_t pos = ((((size_t)s * 16 + j) * 16 + rho) * 4 + cp) * 8 + e;
            index[2 * pos] = channel;
            index[2 * pos + 1] = k;
          }
  return 0;
}

extern "C" int fss_conv3x3_c128_nhwc(const void* x, int x_ld, const void* w_image, const float* bias, void* y,
                                     int y_ld, int n, int h, int wd, int act, int dtype, void* stream) {
  FSS_REQUIRE(x && w_image && y, "null argument");
  FSS_REQUIRE(dtype == FSS_BF16 || dtype == FSS_F16, "only bfloat16 and half precision are supported");
  FSS_REQUIRE(n > 0 && h > 0 && wd > 0, "bad shape");
  FSS_REQUIRE(x_ld >= kHCin && x_ld % 8 == 0 && y_ld >= kHCout && y_ld % 8 == 0,
              "pixel strides must be multiples of 8 and at least the channel count");
  FSS_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w_image & 15) == 0 && ((uintptr_t)y & 15) == 0,
              "unaligned pointer");
  FSS_REQUIRE(act == FSS_ACT_NONE || act == FSS_ACT_RELU || act == FSS_ACT_RELU6, "bad act");
  FSS_REQUIRE((long long)h * wd * x_ld * 2 < 0x7fffffffLL, "frame too large (byte offsets inside one frame are 32-bit)");
  HaloArgs a;
  a.x = (const char*)x; a.w = (const char*)w_image; a.bias = bias; a.y = (char*)y;
  a.N = n; a.H = h; a.W = wd; a.x_ld = x_ld; a.y_ld = y_ld; a.act = act;
  a.bands = wd / kHTW;
  a.row_tiles = (h + kHTH - 1) / kHTH;
  a.rag_w = wd % kHTW;
  a.rag_h = 0;
  int rag_tiles = 0;
  if (a.rag_w > 0) {
    a.rag_h = std::min(256 / a.rag_w, kHPatchPx / (a.rag_w + 2) - 2);
    a.rag_h = std::min(a.rag_h, h);
    rag_tiles = (h + a.rag_h - 1) / a.rag_h;
  }
  a.tiles_per_frame = a.bands * a.row_tiles + rag_tiles;
  FSS_REQUIRE((long long)n * a.tiles_per_frame < 0x7fffffffLL, "too many tiles");
  a.tiles = n * a.tiles_per_frame;
  hipStream_t s = (hipStream_t)stream;
  return dtype == FSS_F16 ? launch_halo<f16_t>(a, s) : launch_halo<bf16_t>(a, s);
}
