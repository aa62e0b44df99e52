// Implicit-GEMM convolution for gfx950 (MI355X) on the matrix cores.
//
//   C[M][N] = A[M][K] * B[K][N],  M = n*ho*wo (output pixels), N = cout,
//   K = kh*kw*cin with k = (r*kw + s)*cin + c.
//
// A is never materialised: every 16-byte chunk of an A row is cin-contiguous
// in the NHWC input, so one lane gathers it with one 16-byte load whose source
// address is the tap's pixel (or a zero page for padding taps, rows past M and
// k past K).  B is the packed weight matrix Wp[cout_pad][k_pad] (K contiguous).
//
// Staging is LDS-DMA (global_load_lds_dwordx4): each wave instruction drops 64
// chunks = 1 KiB of consecutive LDS bytes, so a stage is an unpadded image of
// BM (resp. BN) rows x BKB bytes.  Bank conflicts of the ds_read_b128 fragment
// reads are removed by an XOR swizzle of the 16-byte chunk index with bits of
// the row index, applied on the SOURCE side of the DMA (which lane fetches which
// logical chunk) and again on the read address.  NSTAGE stages form a ring: the
// loads of tile t+NSTAGE-1 are issued before the MFMAs of tile t, and a counted
// s_waitcnt vmcnt + one s_barrier per K tile is all the synchronisation.  The
// DMA is issued from inline asm, so the compiler neither counts nor drains it.
//
// One LDS byte layout serves both dtypes: lane (r = lane&31, h = lane>>5) reads
// the 16 bytes of logical chunk 2*s + h of row r, which is 8 consecutive k for
// v_mfma_f32_32x32x16_bf16 and 4 k values (consumed by 4 x
// v_mfma_f32_32x32x2_f32, with the same k permutation on both operands) for
// float.
//
// Two-input form: y = act(conv(x, W) + conv1x1(x2, W2) + bias).  The K axis is the taps of
// the first filter followed by the cin2 channels of the second input, whose A rows are the
// pixels of x2 themselves ("one more tap" in the scalar tap walk); used for the cascade
// feature fusion, where the sum of two convolutions is needed and neither alone.
//
// Classifier-head form (HEAD kernels): the activated output tile is not stored; it is
// multiplied, from the accumulator registers, by a small [classes][cout] matrix and only the
// class scores leave the block.  The main loop then feeds the MFMA with the operands
// exchanged (D[cout][pixel]), which puts 4 consecutive output channels of one pixel into each
// lane: two such D tiles are the 8 k values of that pixel's B operand of the second product.
//
// Epilogue: each wave transposes its accumulators through LDS (32 rows at a
// time) so that bias / residual / activation run on, and the stores write,
// 16-byte row-contiguous pieces (whole 128-byte lines per row segment), with an
// output pixel stride so a branch can write into a channel slice of a concat
// buffer.
#include "fss_conv_igemm.h"

using namespace fss;

extern "C" int fss_conv_pack_dims(int cin, int kh, int kw, int cout, int dtype, int* k_pad,
                                  int* cout_pad) {
  FSS_REQUIRE(k_pad && cout_pad, "null argument");
  // (FSS_F16_W2: the dimensions of FSS_F16; the packed rows then hold 2 * k_pad elements)
  FSS_REQUIRE(dtype == FSS_F32 || dtype == FSS_BF16 || dtype == FSS_F16 || dtype == FSS_F32_BF16X3 ||
                  dtype == FSS_F16_W2, "bad dtype");
  FSS_REQUIRE(cin > 0 && kh > 0 && kw > 0 && cout > 0, "bad shape");
  const int k = cin * kh * kw;
  *k_pad = (k + 63) / 64 * 64;          // a multiple of every K tile in use
  *cout_pad = (cout + 255) / 256 * 256; // a multiple of every N tile in use
  return 0;
}

extern "C" int fss_conv_config_count(void) { return kNumConvConfigs; }

extern "C" int fss_conv_config_info(int config, int info[7]) {
  FSS_REQUIRE(info != nullptr, "null argument");
  switch (config) {
#define FSS_INFO(ID, BM, BN, WM, WN, BKB, NS, FM)                                  \
  case ID: {                                                                       \
    const int v[7] = {BM, BN, WM, WN, BKB, NS, FM};                                \
    for (int i = 0; i < 7; ++i) info[i] = v[i];                                    \
    return 0;                                                                      \
  }
    FSS_CONV_CONFIGS(FSS_INFO)
#undef FSS_INFO
    case kConv8Config: {
      const int v[7] = {256, 256, 2, 4, 128, 2, 16};
      for (int i = 0; i < 7; ++i) info[i] = v[i];
      return 0;
    }
    default: break;
  }
  set_error("fss_conv_config_info: config out of range");
  return 2;
}

// Argument checks and ConvArgs of both two-input entry points.
static int conv_setup(const fss_conv_desc* d, const void* x, const void* x2, int cin2,
                      const void* w_packed, const float* bias, const void* residual, void* y,
                      ConvArgs& a) {
  FSS_REQUIRE(d && x && w_packed && y, "null argument");
  FSS_REQUIRE(d->in_dtype == FSS_F32 || d->in_dtype == FSS_BF16 || d->in_dtype == FSS_F32_BF16X3 ||
                  d->in_dtype == FSS_F16 || d->in_dtype == FSS_F16_W2, "bad in_dtype");
  FSS_REQUIRE(d->out_dtype == FSS_F32 || d->out_dtype == FSS_BF16 || d->out_dtype == FSS_F16 ||
                  d->out_dtype == FSS_F32_BF16X3, "bad out_dtype");
  FSS_REQUIRE(!(d->in_dtype != FSS_BF16 && d->out_dtype == FSS_BF16),
              "bf16 output needs bf16 input");
  const bool f16_in = d->in_dtype == FSS_F16 || d->in_dtype == FSS_F16_W2;
  FSS_REQUIRE(!((d->out_dtype == FSS_F16 || d->out_dtype == FSS_F32_BF16X3) && !f16_in &&
                d->in_dtype != FSS_F32_BF16X3),
              "f16 and split-bf16 outputs need f16 or FSS_F32_BF16X3 input");
  FSS_REQUIRE(d->in_dtype != FSS_F16_W2 || d->out_dtype == FSS_F16 || d->out_dtype == FSS_F32_BF16X3,
              "FSS_F16_W2 stores FSS_F16 or FSS_F32_BF16X3");
  const bool half_in = d->in_dtype == FSS_BF16 || f16_in;
  // channels per addressable group of the input (16 bytes; 32 for split-bf16 tensors)
  const int epc = half_in || d->in_dtype == FSS_F32_BF16X3 ? 8 : 4;
  const int esz = half_in ? 2 : 4;
  if (d->out_dtype == FSS_F32_BF16X3)
    FSS_REQUIRE(d->cout % 8 == 0 && d->y_ld % 8 == 0 && ((uintptr_t)y & 31) == 0,
                "split-bf16 output: cout and y_ld in multiples of 8, y 32-byte aligned");
  if (residual != nullptr && (d->in_dtype == FSS_F32_BF16X3 || (f16_in && d->out_dtype == FSS_F32_BF16X3)))
    FSS_REQUIRE(d->res_ld % 8 == 0 && ((uintptr_t)residual & 31) == 0,
                "split-bf16 residual: res_ld a multiple of 8, 32-byte aligned");
  FSS_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->cout > 0, "bad shape");
  FSS_REQUIRE(d->cin % epc == 0, "cin must be a multiple of 16 bytes of elements (split-bf16: of 8)");
  if (d->in_dtype == FSS_F32_BF16X3) FSS_REQUIRE(((uintptr_t)x & 31) == 0, "split-bf16 input must be 32-byte aligned");
  FSS_REQUIRE(d->kh > 0 && d->kw > 0 && d->stride > 0 && d->dilation > 0, "bad filter");
  FSS_REQUIRE(d->ho > 0 && d->wo > 0, "bad output size");
  FSS_REQUIRE(d->pad_top >= 0 && d->pad_left >= 0, "negative padding");
  // every output pixel's first tap must start inside the padded image
  FSS_REQUIRE((long long)(d->ho - 1) * d->stride - d->pad_top < d->h &&
                  (long long)(d->wo - 1) * d->stride - d->pad_left < d->w,
              "output size exceeds the input");
  FSS_REQUIRE((x2 == nullptr) == (cin2 == 0) && cin2 >= 0, "x2 and cin2 go together");
  if (cin2 > 0) {
    FSS_REQUIRE(d->cin % 64 == 0 && cin2 % 64 == 0, "two-input form: cin and cin2 must be multiples of 64");
    FSS_REQUIRE(((uintptr_t)x2 & 15) == 0, "unaligned pointer");
    FSS_REQUIRE((long long)d->n * d->ho * d->wo * cin2 * esz < 0xffffffffLL,
                "second input must be smaller than 4 GiB");
  }
  const int k = d->cin * d->kh * d->kw + cin2;
  const int k_pad = (k + 63) / 64 * 64, cout_pad = (d->cout + 255) / 256 * 256;
  FSS_REQUIRE(d->k_pad == k_pad && d->cout_pad == cout_pad, "packed weight dims mismatch");
  FSS_REQUIRE(d->y_ld >= d->cout, "y_ld < cout");
  FSS_REQUIRE(residual == nullptr || d->res_ld >= d->cout, "res_ld < cout");
  FSS_REQUIRE((long long)d->n * d->ho * d->wo < 0x7fffffffLL, "too many output pixels");
  FSS_REQUIRE((long long)d->h * d->w * d->cin * esz < 0x7fffff00LL,
              "one input image must be smaller than 2 GiB");
  FSS_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w_packed & 15) == 0, "unaligned pointer");

  a.x = x; a.x2 = x2; a.w = w_packed; a.bias = bias; a.res = residual; a.y = y;
  a.N = d->n; a.H = d->h; a.W = d->w; a.Cin = d->cin; a.Cout = d->cout;
  a.KH = d->kh; a.KW = d->kw; a.stride = d->stride; a.dil = d->dilation;
  a.pad_t = d->pad_top; a.pad_l = d->pad_left; a.Ho = d->ho; a.Wo = d->wo; a.act = d->act;
  a.Cin2 = cin2;
  a.M = d->n * d->ho * d->wo;
  a.K = k;
  a.Kpad = k_pad; a.y_ld = d->y_ld; a.res_ld = d->res_ld;
  a.tiles_m = a.tiles_n = 0;
  a.splits = 1;
  a.w_head = nullptr; a.bias_head = nullptr; a.cout_head = 0;
  a.zy = a.zytb = a.zylr = a.zycn = nullptr; a.zH = a.zW = 0;
  FSS_REQUIRE(d->config >= 0 && d->config <= kNumConvConfigs, "config out of range");
  return 0;
}

extern "C" int fss_conv2d_dual_nhwc(const fss_conv_desc* d, const void* x, const void* x2, int cin2,
                                    const void* w_packed, const float* bias, const void* residual,
                                    void* y, void* stream) {
  ConvArgs a;
  if (int rc = conv_setup(d, x, x2, cin2, w_packed, bias, residual, y, a)) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (d->in_dtype == FSS_F16 || d->in_dtype == FSS_F16_W2 || d->out_dtype == FSS_F16)
    return dispatch_conv_half(a, d->in_dtype, d->out_dtype, d->config, s);
  if (d->in_dtype == FSS_F32) return dispatch_conv<float, float>(a, d->config, s);
  if (d->in_dtype == FSS_F32_BF16X3 && d->out_dtype == FSS_F32_BF16X3) return dispatch_conv<f32s_t, f32s_t>(a, d->config, s);
  if (d->in_dtype == FSS_F32_BF16X3) return dispatch_conv<f32s_t, float>(a, d->config, s);
  if (d->out_dtype == FSS_BF16) return dispatch_conv<bf16_t, bf16_t>(a, d->config, s);
  return dispatch_conv<bf16_t, float>(a, d->config, s);
}

extern "C" int fss_conv2d_dual_head_nhwc(const fss_conv_desc* d, const void* x, const void* x2, int cin2,
                                         const void* w_packed, const float* bias, const void* residual,
                                         const void* const* zoom_residual, int zh, int zw,
                                         const void* w_head, const float* bias_head, int cout_head,
                                         float* y_head, int y_head_ld, void* stream) {
  FSS_REQUIRE(d && w_head && y_head, "null argument");
  FSS_REQUIRE(d->in_dtype == FSS_BF16 || d->in_dtype == FSS_F32_BF16X3 || d->in_dtype == FSS_F16_W2,
              "the classifier-head form takes FSS_BF16, FSS_F32_BF16X3 or FSS_F16_W2 input");
  const unsigned long long esz = d->in_dtype == FSS_BF16 ? 2ull : 4ull;
  if (d->in_dtype != FSS_BF16 && residual != nullptr)
    FSS_REQUIRE(((uintptr_t)residual & 31) == 0, "split-bf16 residual must be 32-byte aligned");
  FSS_REQUIRE(cout_head > 0 && cout_head <= kHeadMax, "cout_head out of range");
  FSS_REQUIRE(y_head_ld >= cout_head, "y_head_ld < cout_head");
  FSS_REQUIRE(d->cout % 32 == 0 && d->cout <= 256, "cout must be a multiple of 32, at most 256");
  FSS_REQUIRE(((uintptr_t)w_head & 15) == 0 && (bias == nullptr || ((uintptr_t)bias & 15) == 0),
              "unaligned pointer");
  ConvArgs a;
  fss_conv_desc dd = *d;
  dd.y_ld = d->cout;        // no activation tensor is written; keep the shared checks satisfied
  if (d->in_dtype == FSS_F16_W2) dd.out_dtype = FSS_F32_BF16X3;   // (residuals are split-bfloat16)
  FSS_REQUIRE(residual == nullptr || (((uintptr_t)residual & 15) == 0 && d->res_ld % 8 == 0),
              "unaligned residual");
  if (int rc = conv_setup(&dd, x, x2, cin2, w_packed, bias, residual, y_head, a)) return rc;
  const int bk = 64;        // widest K tile in elements (128-byte tile rows of bf16)
  FSS_REQUIRE(a.Cin % bk == 0 && a.Cin2 % bk == 0 && a.KH * a.KW <= 62,
              "the classifier-head form needs cin (and cin2) in multiples of 64");
  if (zoom_residual != nullptr) {
    FSS_REQUIRE(residual == nullptr, "residual and zoom_residual exclude each other");
    FSS_REQUIRE(zh >= 4 && zw >= 4 && d->ho == 2 * zh - 1 && d->wo == 2 * zw - 1,
                "zoom_residual: the output must be (2 zh - 1) x (2 zw - 1), zh, zw >= 4");
    for (int i = 0; i < 4; ++i)
      FSS_REQUIRE(zoom_residual[i] != nullptr && ((uintptr_t)zoom_residual[i] & 15) == 0,
                  "zoom_residual: null or unaligned tensor");
    FSS_REQUIRE((unsigned long long)d->n * zh * zw * d->cout * esz < 0xffffff00ull,
                "zoom_residual: the half-resolution tensor must be below 4 GB");
    a.zy = zoom_residual[0]; a.zytb = zoom_residual[1]; a.zylr = zoom_residual[2]; a.zycn = zoom_residual[3];
    a.zH = zh; a.zW = zw;
  }
  a.y = y_head; a.y_ld = y_head_ld;
  a.w_head = w_head; a.bias_head = bias_head; a.cout_head = cout_head;
  if (d->in_dtype == FSS_F32_BF16X3) return dispatch_conv_head<f32s_t>(a, d->config, (hipStream_t)stream);
  if (d->in_dtype == FSS_F16_W2) return dispatch_conv_head_wide(a, d->config, (hipStream_t)stream);
  return dispatch_conv_head<bf16_t>(a, d->config, (hipStream_t)stream);
}

// Second launch of the split-K form: y = act(sum over slabs + bias + residual), 4 channels a thread,
// slabs added in index order (the result does not depend on which block finished first).
template <typename TR, typename TO>
__global__ void __launch_bounds__(256)
splitk_reduce_kernel(const float* __restrict__ ws, int splits, int M, int ld, int cout,
                     const float* __restrict__ bias, const TR* __restrict__ res, int res_ld,
                     TO* __restrict__ y, int y_ld, int act) {
  const int quads = ld >> 2;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)M * quads) return;
  const int m = (int)(t / quads), n = (int)(t - (long long)m * quads) * 4;
  if (n >= cout) return;
  const size_t slab = (size_t)M * ld;
  const float* p = ws + (size_t)m * ld + n;
  float4 v = *reinterpret_cast<const float4*>(p);
  for (int z = 1; z < splits; ++z) {
    const float4 u = *reinterpret_cast<const float4*>(p + z * slab);
    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
  }
  float o[4] = {v.x, v.y, v.z, v.w};
  const float lo = act == FSS_ACT_NONE ? -INFINITY : 0.f, hi = act == FSS_ACT_RELU6 ? 6.f : INFINITY;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (n + e < cout) {
      float r = o[e] + (bias != nullptr ? bias[n + e] : 0.f);
      if (res != nullptr) r += ElemTraits<TR>::load(res + (size_t)m * res_ld + n + e);
      ElemTraits<TO>::store(y + (size_t)m * y_ld + n + e, fminf(fmaxf(r, lo), hi));
    }
  }
}

extern "C" int fss_conv2d_splitk_nhwc(const fss_conv_desc* d, const void* x, const void* w_packed,
                                      const float* bias, const void* residual, void* y, int splits,
                                      float* workspace, size_t workspace_bytes, void* stream) {
  FSS_REQUIRE(splits >= 2 && splits <= 16, "splits must be 2..16");
  FSS_REQUIRE(workspace != nullptr && ((uintptr_t)workspace & 15) == 0, "null or unaligned workspace");
  FSS_REQUIRE(d == nullptr || d->in_dtype != FSS_F16_W2, "no split-K form of FSS_F16_W2");
  ConvArgs a;
  if (int rc = conv_setup(d, x, nullptr, 0, w_packed, bias, residual, y, a)) return rc;
  const int ld = (d->cout + 3) / 4 * 4;
  FSS_REQUIRE((size_t)splits * a.M * ld * sizeof(float) <= workspace_bytes,
              "workspace too small: splits * n * ho * wo * round_up(cout, 4) floats");
  a.bias = nullptr; a.res = nullptr; a.act = FSS_ACT_NONE;
  a.y = workspace; a.y_ld = ld; a.splits = splits;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (d->in_dtype == FSS_F32) rc = dispatch_conv<float, float>(a, d->config, s);
  else if (d->in_dtype == FSS_F16) rc = dispatch_conv_half(a, FSS_F16, FSS_F32, d->config, s);
  else if (d->in_dtype == FSS_F32_BF16X3) rc = dispatch_conv<f32s_t, float>(a, d->config, s);
  else rc = dispatch_conv<bf16_t, float>(a, d->config, s);
  if (rc) return rc;
  const long long threads = (long long)a.M * (ld / 4);
  const unsigned blocks = (unsigned)((threads + 255) / 256);
  // residual type: the input's; for half-precision input, the output's
#define FSS_REDUCE(TR, TO)                                                                              \
  hipLaunchKernelGGL((splitk_reduce_kernel<TR, TO>), dim3(blocks), dim3(256), 0, s, workspace, splits, a.M, \
                     ld, d->cout, bias, (const TR*)residual, d->res_ld, (TO*)y, d->y_ld, d->act)
  const int in = d->in_dtype, out = d->out_dtype;
  if (in == FSS_BF16 && out == FSS_BF16) FSS_REDUCE(bf16_t, bf16_t);
  else if (in == FSS_BF16) FSS_REDUCE(bf16_t, float);
  else if (in == FSS_F16 && out == FSS_F16) FSS_REDUCE(f16_t, f16_t);
  else if (in == FSS_F16 && out == FSS_F32_BF16X3) FSS_REDUCE(f32s_t, f32s_t);
  else if (in == FSS_F16) FSS_REDUCE(float, float);
  else if (in == FSS_F32_BF16X3 && out == FSS_F16) FSS_REDUCE(f32s_t, f16_t);
  else if (in == FSS_F32_BF16X3 && out == FSS_F32_BF16X3) FSS_REDUCE(f32s_t, f32s_t);
  else if (in == FSS_F32_BF16X3) FSS_REDUCE(f32s_t, float);
  else FSS_REDUCE(float, float);
#undef FSS_REDUCE
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error(std::string("splitk_reduce launch: ") + hipGetErrorString(e));
    return 3;
  }
  return 0;
}

extern "C" int fss_conv2d_nhwc(const fss_conv_desc* d, const void* x, const void* w_packed,
                               const float* bias, const void* residual, void* y, void* stream) {
  return fss_conv2d_dual_nhwc(d, x, nullptr, 0, w_packed, bias, residual, y, stream);
}
