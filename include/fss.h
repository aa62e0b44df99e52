/*
 * libfss_gfx950 -- C ABI of the MI355X (gfx950) segmentation forward-pass kernels.
 *
 * The reference project (oandrienko/fast-semantic-segmentation) has no FFI of
 * its own: every arithmetic op on its inference / evaluation path is a call
 * into TensorFlow 1.8.  Each entry point below replaces one such op (or a
 * fixed chain of them); the comment names the reference call site(s), as
 * file:line relative to the reference checkout.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure;
 *     fss_last_error() returns a thread-local message for the last failure.
 *   - all tensor pointers are DEVICE pointers owned by the caller; activations
 *     are NHWC, dense unless a pixel stride ("ld", in elements) is given.
 *   - `stream` is a hipStream_t (NULL = the null stream); launches are
 *     asynchronous on it.  No entry point in the "kernels" section allocates,
 *     frees or synchronises, so a sequence of them can be stream-captured.
 *   - dtype codes: FSS_F32 = float, FSS_BF16 = bfloat16 (uint16 storage), FSS_F16 = IEEE
 *     half precision (uint16 storage; accepted where an entry point says so).
 */
#ifndef FSS_H_
#define FSS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FSS_F32 0
#define FSS_BF16 1
/* Split-bfloat16 tensors: float values stored as two bfloat16 numbers each, hi = bf16(v) and
 * lo = bf16(v - hi) (about 16 mantissa bits).  A tensor keeps the float layout's size and
 * strides (4 bytes per element, strides and channel counts in elements), but every aligned
 * group of 8 channels - 32 bytes - holds its 8 high parts followed by its 8 remainders, so
 * channel counts, pixel strides and channel offsets are multiples of 8 and base addresses
 * multiples of 32 bytes.  Convolutions form every product on the bf16 matrix pipe as
 * a_hi*b_hi + a_hi*b_lo + a_lo*b_hi with float accumulation: results of float quality (well
 * inside 1e-3 of the logit scale) at several times the speed of the float MFMAs, and because
 * producers split once when they store, the matrix pipe is fed from memory without any
 * conversion.  Packed weights have the FSS_F32 dimensions and byte size with every 8
 * consecutive k values of a row stored the same way.  Every entry point with a dtype
 * argument takes it; a convolution may read it and store FSS_F32 (plain floats, e.g. class
 * scores), FSS_F16 or FSS_F32_BF16X3.  Needs a tile configuration with an even number of k
 * steps per tile. */
#define FSS_F32_BF16X3 2
/* IEEE half precision tensors (11 significant bits against bfloat16's 8, same bytes and the
 * same matrix-core rate): every product is one f16 MFMA with float accumulation, outputs are
 * rounded to nearest even and saturate at +-65504.  Taken by the convolution entry points
 * that name it (fss_conv2d_nhwc and its dual / split-K forms, fss_conv3x3_c64_nhwc,
 * fss_stem2_conv_u8); a convolution may also read FSS_F32_BF16X3 input and store FSS_F16
 * output, or read FSS_F16 and store float, which is how half-precision stretches of a graph
 * join the float ones. */
#define FSS_F16 3
/* fss_conv_desc.in_dtype only: FSS_F16 activations against weights of float quality.  The packed
 * weights hold, per output channel, the k_pad IEEE-half high parts f16(W) followed by the k_pad
 * remainders f16(W - high) ([cout_pad][2 * k_pad], dims from fss_conv_pack_dims); every product
 * is two f16 MFMAs.  Rounding an activation to half precision is an independent error per
 * element that averages out over the K axis, where a rounded weight repeats its error at every
 * pixel: this mode takes the first and avoids the second.  out_dtype FSS_F16 or
 * FSS_F32_BF16X3; the residual has the output's type; configurations whose ring does not fit
 * the LDS with the doubled weight tile are rejected. */
#define FSS_F16_W2 4

#define FSS_ACT_NONE 0
#define FSS_ACT_RELU 1
#define FSS_ACT_RELU6 2

/* ------------------------------------------------------------------ runtime */
const char* fss_last_error(void);
int fss_version(void);
int fss_device_count(int* count);
int fss_set_device(int device);
int fss_device_name(char* buf, int buflen);

int fss_malloc(void** ptr, size_t bytes);
int fss_free(void* ptr);
int fss_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream);
int fss_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream);
int fss_memset(void* dst, int value, size_t bytes, void* stream);

int fss_stream_create(void** stream);
int fss_stream_destroy(void* stream);
int fss_stream_sync(void* stream);
int fss_device_sync(void);

int fss_event_create(void** event);
int fss_event_destroy(void* event);
int fss_event_record(void* event, void* stream);
/* Work enqueued on `stream` after this call waits for `event` (also how a second
 * stream joins, and leaves, a capture started on another one). */
int fss_stream_wait_event(void* stream, void* event);
int fss_event_sync(void* event);
int fss_event_elapsed_ms(void* start, void* stop, float* ms);

/* Stream capture of a whole forward pass into a hipGraph.  Replaces the role
 * of the TF session plan that inference.py:97-98 / libs/evaluator.py:188-199
 * re-run once per image. */
int fss_capture_begin(void* stream);
int fss_capture_end(void* stream, void** graph_exec);
int fss_graph_launch(void* graph_exec, void* stream);
int fss_graph_destroy(void* graph_exec);

/* ------------------------------------------------------------------ kernels */

/* Convolution descriptor.  Weights are pre-packed by the caller as
 * Wp[cout_pad][k_pad] in the INPUT dtype, k = (r*kw + s)*cin + c, with the
 * batch-norm scale folded in and zero padding; fss_conv_pack_dims gives
 * k_pad / cout_pad.  cin must be a multiple of 16/sizeof(in dtype). */
typedef struct fss_conv_desc {
  int32_t n, h, w, cin;      /* input  [n,h,w,cin]                         */
  int32_t cout, kh, kw;      /* filter [kh,kw,cin,cout]                    */
  int32_t stride, dilation;
  int32_t pad_top, pad_left; /* zero padding before the first row / column */
  int32_t ho, wo;            /* output spatial size                        */
  int32_t act;               /* FSS_ACT_*                                  */
  int32_t in_dtype;          /* FSS_F32 | FSS_BF16 | FSS_F16 | FSS_F32_BF16X3
                                (x, weights, residual), or FSS_F16_W2
                                (FSS_F16 x, two-f16 weights, residual of
                                the output's type)                          */
  int32_t out_dtype;         /* FSS_F32 (any input but FSS_F16_W2) |
                                FSS_BF16 (bf16 input) |
                                FSS_F16 or FSS_F32_BF16X3 (FSS_F16,
                                FSS_F16_W2 or FSS_F32_BF16X3 input)         */
  int32_t y_ld;              /* output pixel stride in elements (>= cout)  */
  int32_t res_ld;            /* residual pixel stride in elements          */
  int32_t k_pad, cout_pad;   /* packed weight dims                         */
  int32_t config;            /* tile configuration: 0 = automatic, else
                                1..fss_conv_config_count(); configurations
                                with the same mfma_edge (see
                                fss_conv_config_info) give bit-identical
                                results, only speed differs                */
} fss_conv_desc;

int fss_conv_pack_dims(int cin, int kh, int kw, int cout, int dtype,
                       int* k_pad, int* cout_pad);
int fss_conv_config_count(void);
/* Geometry of configuration `config` (1..count), for tuners and tests:
 * info[0..6] = block rows (output pixels), block columns (output channels),
 * waves along rows, waves along columns, bytes of K per tile row, LDS ring
 * stages, edge of the MFMA output tile (32: v_mfma_*_32x32, 16: v_mfma_*_16x16). */
int fss_conv_config_info(int config, int info[7]);

/* y = act(conv(x, W) + bias [+ residual]).
 * Replaces slim.conv2d + slim.batch_norm(inference) + activation
 * (libs/compressible_ops.py:9-24, third_party/dilated_resnet_v1.py:90-102,
 * third_party/resnet_utils.py:111-122, architectures/icnet_architecture.py:208-247,
 * architectures/pspnet_architecture.py:135,144) and, with `residual`, the
 * shortcut add + relu of third_party/dilated_resnet_v1.py:109, the add + relu of
 * architectures/icnet_architecture.py:251-252 and third_party/conv_blocks.py:313.
 * `residual` has the output's shape and the input dtype (FSS_F16 input: the OUTPUT dtype, so
 * that a float residual stream can absorb a half-precision convolution); NULL = none. */
int fss_conv2d_nhwc(const fss_conv_desc* desc, const void* x, const void* w_packed,
                    const float* bias, const void* residual, void* y, void* stream);

/* The same convolution with the K axis split over `splits` (2..16) blocks per output tile, for
 * layers whose output has too few tiles to fill the chip (a few thousand pixels, K in the
 * thousands: the backbone's late blocks at one frame per launch).  Two launches: the blocks store
 * raw float sums of their K share into slab z of `workspace` ([splits][n*ho*wo][round_up(cout, 4)]
 * floats, workspace_bytes >= that), then one pass adds the slabs in index order and applies bias,
 * residual and activation.  Same call sites as fss_conv2d_nhwc; the result differs from it only
 * by the order of the float additions. */
int fss_conv2d_splitk_nhwc(const fss_conv_desc* desc, const void* x, const void* w_packed,
                           const float* bias, const void* residual, void* y, int splits,
                           float* workspace, size_t workspace_bytes, void* stream);

/* Sum of two convolutions in one launch:
 *   y = act(conv(x, W) + conv1x1(x2, W2) + bias [+ residual])
 * x2: [n,ho,wo,cin2] (the output's spatial size, stride 1).  Packed weights: the K axis
 * of W (k = (r*kw + s)*cin + c) followed by the cin2 channels of W2, i.e.
 * k_pad = roundup64(kh*kw*cin + cin2); bias = the sum of both layers' shifts.
 * Replaces the cascade feature fusion of architectures/icnet_architecture.py:238-252
 * (conv 1x1 of the high-resolution input + dilated conv 3x3 of the upsampled
 * low-resolution input, each with batch norm, added, relu) without storing either
 * addend.  cin and cin2 must be multiples of 64.  x2 = NULL, cin2 = 0 is
 * fss_conv2d_nhwc. */
int fss_conv2d_dual_nhwc(const fss_conv_desc* desc, const void* x, const void* x2, int cin2,
                         const void* w_packed, const float* bias, const void* residual,
                         void* y, void* stream);

/* Convolution (one or two inputs, as above) followed by a 1x1 classifier, in one launch:
 *   y_head[m][c] = bias_head[c] + sum_n bf16(act(conv(x, W) + conv1x1(x2, W2) + bias + residual))[m][n] * w_head[c][n]
 * The activated cout-channel tensor is never stored; only the cout_head class scores per
 * pixel leave the kernel (float32, pixel stride y_head_ld).  Replaces the last cascade
 * feature fusion + the `Predictions` slim.conv2d of architectures/icnet_architecture.py:
 * 131-142,238-252 (the classifier applied before the x2 upsampling, which commutes with a
 * 1x1 convolution).  desc->in_dtype FSS_F16_W2 behaves as FSS_F32_BF16X3 below with FSS_F16
 * activations x, x2 and the two-f16 packed weights of that mode (residuals, w_head and the
 * zoom sources as for FSS_F32_BF16X3).  desc->in_dtype FSS_BF16 (the rounding to bf16 in the formula above is
 * that mode's) or FSS_F32_BF16X3 (split-bfloat16 tensors, nothing rounded to 8 bits: the
 * activated values are split like every other activation); cin (and cin2) multiples of 64; cout a multiple
 * of 32 and <= 256; cout_head <= 20; w_head: bf16 [32][cout], rows >= cout_head zero
 * (FSS_F32_BF16X3: [2][32][cout], the bf16 high parts of the float weights, then their bf16
 * remainders); desc->config: 0 or a 16x16-MFMA configuration whose block columns >= cout
 * (FSS_F32_BF16X3: with 128 bytes of K per tile row);
 * `residual`: [pixels][cout] of the activation type with pixel stride desc->res_ld, or NULL.
 * `zoom_residual` (or NULL): the residual given at half resolution, as the four tensors
 * {y, ytb, ylr, ycn} of fss_zoom2_conv_commute (bfloat16 with FSS_BF16 input, plain FSS_F32
 * with FSS_F32_BF16X3 input: nothing multiplies them, so their producer stores them unsplit;
 * cout channels, y: [n,zh,zw,cout], ytb: [2n,1,zw,cout], ylr: [2n,zh,1,cout], ycn: [4n,1,1,cout]) and
 * interpolated in the epilogue instead of being read from a zoomed copy; ho = 2 zh - 1,
 * wo = 2 zw - 1.
 * desc->out_dtype and y_ld are ignored.  Equals fss_conv2d_dual_nhwc followed by
 * fss_conv2d_nhwc up to float summation order. */
int fss_conv2d_dual_head_nhwc(const fss_conv_desc* desc, const void* x, const void* x2, int cin2,
                              const void* w_packed, const float* bias, const void* residual,
                              const void* const* zoom_residual, int zh, int zw,
                              const void* w_head, const float* bias_head, int cout_head,
                              float* y_head, int y_head_ld, void* stream);

/* Two 1x1 convolutions in a row, the first with a residual, as one streaming launch (bfloat16):
 *   y  = act1(conv1x1([x | x2], w) + bias + residual)       [pixels][cout1], stored
 *   y2 = act2(conv1x1(y as stored, w2) + bias2)             [pixels][cout2], stored
 * Replaces the last convolution of a bottleneck unit (with its projection shortcut as the
 * second input x2, or its identity shortcut as `residual`), the add + relu, and the first
 * convolution of the next unit (third_party/dilated_resnet_v1.py:50-113) where the filters are
 * small: cin + cin2 in {32, 64, 128} (each a multiple of 32), cout1 in {128, 256}, cout2 in
 * {32, 64}.  The cout1-channel tensor is written once and not read back by the second layer.
 * w: bf16 [cout1][cin + cin2] (batch norm folded in, the channels of x first), w2: bf16
 * [cout2][cout1]; bias, bias2: float or NULL; residual: bf16 [pixels][cout1] or NULL; *_ld:
 * pixel strides in elements (multiples of 8).  Equals fss_conv2d_dual_nhwc followed by
 * fss_conv2d_nhwc up to float summation order. */
int fss_conv1x1_chain_nhwc(const void* x, int cin, int x_ld, const void* x2, int cin2, int x2_ld,
                           const void* w, const float* bias, const void* residual, int res_ld,
                           int act1, void* y, int y_ld, int cout1, const void* w2,
                           const float* bias2, int act2, void* y2, int y2_ld, int cout2,
                           long long pixels, int dtype, void* stream);

/* 3x3 convolution of a 64-channel bfloat16 tensor with the whole filter resident in LDS:
 *   y = act(conv3x3(x, w) + bias),  cout 64 or 128, any stride (<= 4), dilation, padding.
 * Same result as fss_conv2d_nhwc up to float summation order; for the layers whose K axis
 * (576) is too short for the implicit GEMM to amortise its staging: conv2 of the first ResNet
 * block's units (third_party/dilated_resnet_v1.py:95-97) and `Conv_2` of the full-resolution
 * branch (architectures/icnet_architecture.py:220-222).  w: bf16 [cout][576], k = (r*3 + s)*64
 * + c, batch norm folded in (the first cout rows of fss_conv2d_nhwc's packed matrix, whose row
 * length for this layer is 576 too); bias: float [cout] or NULL; x_ld / y_ld: pixel strides in
 * elements (multiples of 8). */
int fss_conv3x3_c64_nhwc(const void* x, int x_ld, const void* w, const float* bias, void* y,
                         int y_ld, int n, int h, int wd, int ho, int wo, int cout, int stride,
                         int dilation, int pad_top, int pad_left, int act, int dtype, void* stream);

/* 3x3 convolution, stride 1, dilation 1, one pixel of zero padding all round, 128 -> 256
 * channels (FSS_BF16 or FSS_F16), with the input tile and its halo resident in LDS:
 *   y[n,h,wd,256] = act(conv3x3(x[n,h,wd,128], w) + bias)
 * `Conv_3` of the full-resolution branch (architectures/icnet_architecture.py:223-225).  Persistent
 * blocks; every input pixel goes through the global -> LDS path once per tile instead of once per
 * tap, the filter streams through a ring that does not stop between tiles, the results leave
 * the accumulators as 16-byte stores.  Bit-identical to fss_conv2d_nhwc on the same operands
 * (same MFMA, same order along k).  w_image: the filter as the kernel's LDS image, 294912
 * 16-bit elements; element e is W[index[2e]][index[2e+1]] of fss_conv2d_nhwc's packed matrix
 * ([256][1152], k = (r*3 + s)*128 + c, batch norm folded in) with `index` from
 * fss_conv3x3_c128_pack_index (2 * 294912 entries).  bias: float [256] or NULL; x_ld / y_ld: pixel
 * strides in elements (multiples of 8, at least 128 / 256); a frame of x is less than 2 GiB. */
#define FSS_CONV3X3_C128_IMAGE_ELEMS 294912
int fss_conv3x3_c128_pack_index(int32_t* index);
int fss_conv3x3_c128_nhwc(const void* x, int x_ld, const void* w_image, const float* bias, void* y,
                          int y_ld, int n, int h, int wd, int act, int dtype, void* stream);

/* Pieces of  conv3x3_dilation2(zoom_x2(x)) = zoom_x2(conv3x3(x))  (align_corners zoom to
 * (2h-1, 2w-1), zero padding, both sides linear in x), which holds everywhere except on
 * output rows / columns 1 and 2h-3 / 2w-3: there one of the source results that the zoom
 * averages must be computed without the border row / column of x, because the tap that
 * reaches it lands on the zero padding of the zoomed map.  Used for the dilated convolution
 * of the cascade feature fusion (architectures/icnet_architecture.py:238-247), which then
 * runs on a quarter of the pixels.
 *   fss_zoom2_border_gather: crops of x [n,h,w,c] with that row / column zeroed:
 *     tb [2n,3,w,c] (top rows 0-2 with row 0 zero; bottom rows h-3..h-1 with the last zero),
 *     lr [2n,h,3,c] (left, right likewise), cn [4n,3,3,c] (corners TL, TR, BL, BR, both zero).
 *   The caller convolves x with the 3x3 filter (dilation 1, padding 1) -> y [n,h,w,c'] and
 *   the crops with the same filter, padded along their long axis only, so that just the
 *   middle row / column / pixel comes out: ytb [2n,1,w,c'] (pad_top 0, pad_left 1),
 *   ylr [2n,h,1,c'] (pad_top 1, pad_left 0), ycn [4n,1,1,c'] (no padding).
 *   fss_zoom2_conv_commute: r [n,2h-1,2w-1,c'] = zoom_x2(y) with ytb / ylr / ycn substituted
 *     where the identity needs them.  h, w >= 4. */
int fss_zoom2_border_gather(const void* x, void* tb, void* lr, void* cn, int n, int h, int w, int c,
                            int dtype, void* stream);
int fss_zoom2_conv_commute(const void* y, const void* ytb, const void* ylr, const void* ycn, void* r,
                           int n, int h, int w, int c, int dtype, void* stream);

/* Depthwise 3x3-style convolution + bias + activation.
 * Replaces slim.separable_conv2d(num_outputs=None) + batch_norm + relu6
 * (third_party/conv_blocks.py:240-249,280).  w: float [kh*kw][c] (BN folded).  dtype may
 * also be FSS_F16 (float arithmetic, half-precision tensors). */
int fss_dwconv2d_nhwc(const void* x, const float* w, const float* bias, void* y,
                      int n, int h, int wd, int c, int kh, int kw, int stride,
                      int dilation, int pad_top, int pad_left, int ho, int wo,
                      int act, int dtype, void* stream);

/* Max pooling; padded taps are ignored.  dtype may also be FSS_F16.
 * Replaces slim.max_pool2d(net, [3,3], stride=2) under SAME
 * (third_party/dilated_resnet_v1.py:175). */
int fss_maxpool_nhwc(const void* x, void* y, int n, int h, int wd, int c, int k,
                     int stride, int pad_top, int pad_left, int ho, int wo,
                     int dtype, void* stream);

/* y[n,i,j,:] = x[n,i*stride,j*stride,:].
 * Replaces resnet_utils.subsample = max_pool2d([1,1], stride)
 * (third_party/resnet_utils.py:59-74). */
int fss_subsample_nhwc(const void* x, void* y, int n, int h, int wd, int c,
                       int stride, int ho, int wo, int dtype, void* stream);

/* VALID average pooling with kernel == stride == (kh,kw).
 * Replaces slim.avg_pool2d(x, (kh,kw), stride=(kh,kw))
 * (architectures/icnet_architecture.py:197-200,
 * architectures/pspnet_architecture.py:130-133).  y: [n,bh,bw,c]. */
int fss_avgpool_bins_nhwc(const void* x, void* y, int n, int h, int wd, int c,
                          int kh, int kw, int bh, int bw, int dtype, void* stream);

/* The `levels` (<= 4) bin-average maps of one pyramid pooling module in a single
 * pass over x: level l is fss_avgpool_bins_nhwc with (kh[l],kw[l],bh[l],bw[l]) and
 * output y[l].  Replaces the loop over pooling factors
 * (architectures/icnet_architecture.py:194-205,
 * architectures/pspnet_architecture.py:127-141).  `workspace` is caller-owned
 * device memory of at least n*h*sum(bw)*c floats (row sums between the two
 * launches); the entry point itself allocates nothing. */
int fss_avgpool_pyramid_nhwc(const void* x, void* const* y, const int* kh, const int* kw,
                             const int* bh, const int* bw, int levels, float* workspace,
                             size_t workspace_bytes, int n, int h, int wd, int c, int dtype,
                             void* stream);

/* fss_avgpool_pyramid_nhwc without a workspace argument, for callers that cannot allocate
 * device memory (code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc).  It has no counterpart in the reference.  The row sums go to a static device
 * array of FSS_PYRAMID_STATIC_MAX_BYTES inside the library (every GPU the library runs on
 * holds one copy until the process ends); n*h*sum(bw)*c*4 bytes beyond that are rejected
 * with an error.  The call allocates and frees nothing; its first call on a device (or
 * fss_set_device) looks the array's device address up once, which may load the library's code
 * object: callers that capture the call into a graph make fss_set_device, or one plain
 * warm-up call on that device, come first.  Every call on one GPU uses
 * that GPU's array, so two calls that overlap on one GPU (different streams) race; calls
 * ordered on one stream are safe.  Results are bit-identical to fss_avgpool_pyramid_nhwc. */
#define FSS_PYRAMID_STATIC_MAX_BYTES (32u << 20)
int fss_avgpool_pyramid_nhwc_static(const void* x, void* const* y, const int* kh, const int* kw,
                                    const int* bh, const int* bw, int levels, int n, int h,
                                    int wd, int c, int dtype, void* stream);

/* tf.image.resize_bilinear (TF 1.x legacy sampling), optional align_corners.
 * Replaces architectures/icnet_architecture.py:249,264-266,
 * architectures/pspnet_architecture.py:136-139,158-160,
 * third_party/dilated_resnet_v1.py:45-47, third_party/mobilenet_v2.py:49-51,
 * libs/evaluator.py:68-71.  y_ld = output pixel stride in elements. */
int fss_resize_bilinear_nhwc(const void* x, void* y, int n, int h, int wd, int c,
                             int ho, int wo, int y_ld, int align_corners,
                             int dtype, void* stream);

/* y = x + sum_i resize_bilinear_align_corners(p_i) for up to 8 pooled maps,
 * added in index order.  Replaces the avg_pool -> resize -> tf.add loop of
 * architectures/icnet_architecture.py:194-205.  p[i]: [n,bh[i],bw[i],c].
 * x = NULL: the sum of the resized maps alone (used when the 1x1 convolution that follows the
 * loop, architectures/icnet_architecture.py:206-210, is applied to x and to the small maps
 * separately, both operations being linear, and this sum is that convolution's residual). */
int fss_pyramid_add_nhwc(const void* x, const void* const* p, const int* bh,
                         const int* bw, int levels, void* y, int n, int h, int wd,
                         int c, int dtype, void* stream);

/* y = act(a + b), elementwise. (third_party/dilated_resnet_v1.py:109,
 * architectures/icnet_architecture.py:251-252 when not fused into a conv.) */
int fss_add_act(const void* a, const void* b, void* y, size_t count, int act,
                int dtype, void* stream);

/* y = x rounded to FSS_F16 (to nearest even, saturating), elementwise over `count` elements of
 * a dense FSS_F32 or FSS_F32_BF16X3 tensor.  The storage conversion in front of a convolution
 * that reads half-precision activations against two-f16 weights (in_dtype FSS_F16_W2) when its
 * input exists only at float quality and its K axis is long enough to repay the pass: the 3x3
 * convolution over the 4096-channel concatenation of architectures/pspnet_architecture.py:143-147. */
int fss_to_half(const void* x, void* y, size_t count, int dtype, void* stream);

/* Copies a [pixels, c] tensor into / out of a wider one (channel concat).
 * Replaces tf.concat(axis=-1) (architectures/pspnet_architecture.py:143). */
int fss_copy_channels(const void* x, int x_ld, void* y, int y_ld, size_t pixels,
                      int c, int dtype, void* stream);

/* uint8 image -> zero-padded, mean-subtracted, scaled float tensor:
 * y[n,i,j,ch] = ((i<h && j<w ? x[n,i,j,ch] : 0) - mean[ch]) * scale[ch] for
 * ch < 3, and 0 for 3 <= ch < c_out.  Replaces tf.image.pad_to_bounding_box +
 * tf.to_float + extractor.preprocess (libs/exporter.py:37-38,52-53,
 * extractors/pspnet_icnet_resnet_v1.py:43-45,
 * extractors/pspnet_icnet_mobilenet_v2.py:44-46). */
int fss_preprocess_u8(const uint8_t* x, void* y, int n, int h, int wd, int ho,
                      int wo, int c_out, const float* mean3, const float* scale3,
                      int dtype, void* stream);

/* First convolution of a network computed straight from the uint8 frame:
 *   img[n,i,j,ch] = (frame pixel (i*sub, j*sub), or 0 beyond the hr x wr frame,
 *                    minus mean[ch]) * scale[ch]       for 0 <= i < h, 0 <= j < w
 *   y = act(conv(img, W; stride, dilation, zero padding pad_top/pad_left) + bias)
 * in `dtype` arithmetic (bf16: img and W rounded to bf16, float accumulation)
 * with `dtype` output, pixel stride y_ld.  `sub` = 2 is the reference's
 * align_corners half-size resize of the padded frame (its scale is exactly 2,
 * so it keeps every second pixel).  Fuses tf.image.pad_to_bounding_box +
 * tf.to_float + extractor.preprocess (libs/exporter.py:37-38,52-53,
 * extractors/pspnet_icnet_resnet_v1.py:43-45) [+ the 0.5 resize of
 * architectures/icnet_architecture.py:92-101] + the first slim.conv2d + batch
 * norm + activation (third_party/dilated_resnet_v1.py conv1,
 * architectures/icnet_architecture.py:153-160 'Conv',
 * third_party/mobilenet_v2.py first conv).  Weights: [cout_pad][k_pad] with
 * k = (r*kw + s)*4 + ch (ch 3 = 0), dims from fss_stem_pack_dims, batch norm
 * folded in.  dtype FSS_F32_BF16X3: split-bfloat16 output (cout and y_ld multiples of 8) of
 * float quality.  The kernel then multiplies the RAW frame bytes (exact in half precision) and,
 * as channel 3, a 1 for every pixel inside the h x w image; the caller folds mean and scale into
 * the weights - W'[.][4 tap + c] = W[tap][c] * scale[c], W'[.][4 tap + 3] = -sum_c W[tap][c] *
 * mean[c] * scale[c] - and packs them as two IEEE-half matrices [cout_pad][k_pad], the high parts
 * f16(W') followed by the remainders f16(W' - high) (dims from fss_stem_pack_dims with that
 * dtype); mean3 / scale3 are not used.  No activation is rounded and a product costs two MFMAs.
 * dtype FSS_F16_W2: the same arithmetic and weights with FSS_F16 output (saturating).
 * Limits: cout <= 64, kh*kw <= 64, stride, dilation <= 4. */
int fss_stem_pack_dims(int kh, int kw, int cout, int dtype, int* k_pad, int* cout_pad);
int fss_stem_conv_u8(const uint8_t* x, const void* w_packed, const float* bias, void* y,
                     int n, int hr, int wr, int sub, int h, int w,
                     const float* mean3, const float* scale3, int cout, int kh, int kw,
                     int stride, int dilation, int pad_top, int pad_left, int ho, int wo,
                     int act, int dtype, int y_ld, void* stream);

/* The first TWO convolutions of a branch in one kernel, from the uint8 frame:
 *   y1 = act1(conv3x3/2(img, W1; pad_top1/pad_left1) + bias1)   [n,h1,w1,c1], rounded to `dtype`
 *   y  = act2(conv3x3/2(y1,  W2; pad_top2/pad_left2) + bias2)   [n,h2,w2,c2], pixel stride y_ld
 * with img as in fss_stem_conv_u8.  y1 never reaches memory: it is the largest
 * activation of ICNet (64 channels at 513 x 1025), and the result equals
 * fss_stem_conv_u8 followed by fss_conv2d_nhwc.  Replaces the first two
 * slim.conv2d + batch norm + activation of the full-resolution branch
 * (architectures/icnet_architecture.py:213-226 'Conv', 'Conv_1') together with the
 * preprocessing listed at fss_stem_conv_u8.  W1 packed as for fss_stem_conv_u8
 * (fss_stem_pack_dims(3, 3, c1)), W2 as for fss_conv2d_nhwc
 * (fss_conv_pack_dims(c1, 3, 3, c2)).  Limits: dtype bf16, c1 and c2 in {32, 64},
 * paddings <= 2, y_ld a multiple of 8. */
int fss_stem2_conv_u8(const uint8_t* x, const void* w1_packed, const float* bias1,
                      const void* w2_packed, const float* bias2, void* y, int n, int hr,
                      int wr, int sub, int h, int w, const float* mean3, const float* scale3,
                      int c1, int pad_top1, int pad_left1, int h1, int w1, int act1, int c2,
                      int pad_top2, int pad_left2, int h2, int w2, int act2, int dtype,
                      int y_ld, void* stream);

/* Bilinear (align_corners) upsample of float logits [n,h,w,c] (pixel stride
 * x_ld >= c floats) to [n,ho,wo] fused with argmax over channels (lowest index
 * wins ties); labels are uint8.  If `logits_out` is non-NULL the upsampled
 * logits [n,ho,wo,c] (dense) are stored too.
 * Replaces _dynamic_interpolation(z_factor) + tf.argmax(.., 3)
 * (architectures/icnet_architecture.py:142-144, libs/exporter.py:41,
 * libs/evaluator.py:65-72,85). */
int fss_upsample_argmax(const float* logits, uint8_t* labels, float* logits_out,
                        int n, int h, int wd, int c, int x_ld, int ho, int wo, void* stream);

/* y[n,i,j,:] = table[labels[n,i,j]] for i<ho, j<wo (top-left crop).
 * Replaces one_hot @ color_table + crop_to_bounding_box
 * (libs/exporter.py:14-29,71-73).  table: float [num_classes][channels]. */
int fss_label_lut(const uint8_t* labels, float* y, const float* table, int n,
                  int h, int wd, int ho, int wo, int channels, int num_classes,
                  void* stream);

/* cm[label*num_classes + pred] += 1 for every pixel whose label != ignore_label;
 * labels == ignore_label are dropped (weight 0).  cm is int64 and is
 * accumulated into (not cleared).  Replaces the ignore mask +
 * tf.contrib.metrics.streaming_mean_iou update op (libs/evaluator.py:143-154). */
int fss_confusion_accumulate(const uint8_t* pred, const uint8_t* labels,
                             size_t count, int num_classes, int ignore_label,
                             int64_t* cm, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FSS_H_ */
