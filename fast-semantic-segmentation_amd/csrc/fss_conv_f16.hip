// Half-precision (FSS_F16) instantiations of the implicit-GEMM convolution: f16 tensors on
// v_mfma_f32_16x16x32_f16, and the forms that join a half-precision stretch of a graph to the
// split-bfloat16 one (split-bf16 arithmetic with f16 output; f16 arithmetic with split-bf16 or
// float output).  Same kernel template, tile configurations and LDS byte layout as bfloat16.
#include "fss_conv_igemm.h"

namespace fss {

int dispatch_conv_half(ConvArgs& a, int in_dtype, int out_dtype, int config, hipStream_t stream) {
  if (in_dtype == FSS_F16 && out_dtype == FSS_F16) return dispatch_conv<f16_t, f16_t>(a, config, stream);
  if (in_dtype == FSS_F16 && out_dtype == FSS_F32) return dispatch_conv<f16_t, float>(a, config, stream);
  if (in_dtype == FSS_F16 && out_dtype == FSS_F32_BF16X3) return dispatch_conv<f16_t, f32s_t>(a, config, stream);
  if (in_dtype == FSS_F32_BF16X3 && out_dtype == FSS_F16) return dispatch_conv<f32s_t, f16_t>(a, config, stream);
  if (in_dtype == FSS_F16_W2 && out_dtype == FSS_F16) return dispatch_conv<f16w_t, f16_t>(a, config, stream);
  if (in_dtype == FSS_F16_W2 && out_dtype == FSS_F32_BF16X3) return dispatch_conv<f16w_t, f32s_t>(a, config, stream);
  set_error("fss_conv2d_nhwc: unsupported combination of FSS_F16 with another dtype");
  return 2;
}

int dispatch_conv_head_wide(ConvArgs& a, int config, hipStream_t stream) {
  return dispatch_conv_head<f16w_t>(a, config, stream);
}

}  // namespace fss
