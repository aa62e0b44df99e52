// Probe for the packed-f32 anomaly: fss_pyramid_add_nhwc built WITH hipcc's SLP vectorizer
// (v_pk_fma_f32 / v_pk_add_f32 blends), launched alone and beside other kernels on a second
// stream, outputs compared bit for bit with the first (solo) result.  No graphs, no Python.
//
// Build (two variants):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Ifast-semantic-segmentation_amd/csrc \
//         tools/pyramid_slp_probe.hip -o tools/pyramid_slp_probe
//   ... -fno-slp-vectorize ... -o tools/pyramid_noslp_probe
#include "fss_kernels.hip"

#include <string.h>
#include <vector>

namespace fss {
void set_error(const std::string& msg) { fprintf(stderr, "fss error: %s\n", msg.c_str()); }
}  // namespace fss

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { \
  fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(256) mfma_aggressor(float* sink, int iters) {
  bf8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
  f16v acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
  for (int it = 0; it < iters; ++it) {
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc3, 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
  if (s == 12345.678f) sink[0] = s;
}

// LDS traffic + VALU + global loads: the shape of a convolution's staging loop without MFMA
__global__ void __launch_bounds__(256) lds_aggressor(const float4* __restrict__ x, float* sink, size_t n,
                                                     int rounds) {
  __shared__ float4 tile[1024];
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
      tile[threadIdx.x] = x[i];
      __syncthreads();
      const float4 v = tile[(threadIdx.x * 7 + 3) & 255];
      acc += v.x * v.y + v.z * v.w;
      __syncthreads();
    }
  }
  if (acc == 12345.678f) sink[1] = acc;
}

int main() {
  const int n = 2, h = 33, w = 65, c = 2048, L = 4;
  const int bh[L] = {1, 2, 3, 4}, bw[L] = {1, 2, 3, 4};
  const size_t px = (size_t)n * h * w * c;
  unsigned s = 99u;
  auto rnd_bf16 = [&]() {
    s = s * 1664525u + 1013904223u;
    const float f = (float)((s >> 8) & 0xffff) / 8192.f;      // 0 .. 8, like a relu output
    unsigned u;
    memcpy(&u, &f, 4);
    return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
  };
  std::vector<unsigned short> hx(px);
  for (auto& v : hx) v = rnd_bf16();
  unsigned short *x, *y;
  void* p[L];
  CK(hipMalloc(&x, px * 2));
  CK(hipMalloc(&y, px * 2));
  CK(hipMemcpy(x, hx.data(), px * 2, hipMemcpyHostToDevice));
  for (int l = 0; l < L; ++l) {
    const size_t e = (size_t)n * bh[l] * bw[l] * c;
    std::vector<unsigned short> hp(e);
    for (auto& v : hp) v = rnd_bf16();
    CK(hipMalloc(&p[l], e * 2));
    CK(hipMemcpy(p[l], hp.data(), e * 2, hipMemcpyHostToDevice));
  }
  const size_t n_src = (size_t)1 << 22;
  float4* big;
  float* sink;
  CK(hipMalloc(&big, n_src * 16));
  CK(hipMemset(big, 0, n_src * 16));
  CK(hipMalloc(&sink, 64));
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));

  std::vector<unsigned short> ref(px), got(px);
  if (fss_pyramid_add_nhwc(x, p, bh, bw, L, y, n, h, w, c, FSS_BF16, s0)) return 1;
  CK(hipStreamSynchronize(s0));
  CK(hipMemcpy(ref.data(), y, px * 2, hipMemcpyDeviceToHost));

  const char* names[4] = {"alone", "beside mfma kernel", "beside lds/valu kernel", "beside both"};
  for (int mode = 0; mode < 4; ++mode) {
    size_t bad_runs = 0, bad_elems = 0;
    size_t by_e[8] = {0};
    const int runs = 300;
    for (int r = 0; r < runs; ++r) {
      if (r % 25 == 0) {      // aggressors of a few ms each, re-armed while the victims run
        CK(hipStreamSynchronize(s1));
        if (mode == 1 || mode == 3) hipLaunchKernelGGL(mfma_aggressor, dim3(512), dim3(256), 0, s1, sink, 60000);
        if (mode == 2 || mode == 3) hipLaunchKernelGGL(lds_aggressor, dim3(512), dim3(256), 0, s1, big, sink, n_src, 3);
      }
      CK(hipMemsetAsync(y, 0xff, px * 2, s0));
      if (fss_pyramid_add_nhwc(x, p, bh, bw, L, y, n, h, w, c, FSS_BF16, s0)) return 1;
      CK(hipMemcpyAsync(got.data(), y, px * 2, hipMemcpyDeviceToHost, s0));
      CK(hipStreamSynchronize(s0));
      size_t bad = 0;
      if (memcmp(got.data(), ref.data(), px * 2) != 0)
        for (size_t i = 0; i < px; ++i)
        if (got[i] != ref[i]) { ++bad; ++by_e[i & 7]; }
      if (bad) { ++bad_runs; bad_elems += bad; }
    }
    CK(hipDeviceSynchronize());
    printf("%-24s %d runs: %zu with wrong elements (%zu elements; by position in the 16-byte chunk:",
           names[mode], runs, bad_runs, bad_elems);
    for (int e = 0; e < 8; ++e) printf(" %zu", by_e[e]);
    printf(")\n");
    fflush(stdout);
  }
  return 0;
}
