// Pure-MFMA ceiling of the box: every SIMD issues independent v_mfma_f32_16x16x32_bf16 from
// registers, nothing else (no LDS, no memory).  Prints the sustained TFLOP/s for zero-filled and
// for random operands (data toggling lowers the clock the chip holds) and for 1, 2 and 4 waves per
// SIMD - the number every "fraction of peak" of the convolution kernels should be read against.
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o exp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__global__ void __launch_bounds__(256) mfma_loop(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
  const uint4 av = src[threadIdx.x], bv = src[256 + threadIdx.x];
  const bf16x8 a = __builtin_bit_cast(bf16x8, av), b = __builtin_bit_cast(bf16x8, bv);
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;          // keeps the loop alive
}

int main() {
  std::vector<unsigned short> h(512 * 8);
  uint4* d; float* o;
  if (hipMalloc((void**)&d, h.size() * 2) != hipSuccess || hipMalloc((void**)&o, 64) != hipSuccess) return 1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 40000;
  for (int random = 0; random < 2; ++random) {
    srand(1);
    for (auto& v : h) {
      // bf16 bit patterns of values in [-1, 1): sign, exponent 0x7e or below, random mantissa
      v = random ? (unsigned short)(((rand() & 1) << 15) | ((0x70 + (rand() % 15)) << 7) | (rand() & 0x7f)) : 0;
    }
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int wps = 1; wps <= 4; wps *= 2) {           // waves per SIMD
      const int blocks = 256 * wps;                   // 256 CUs x (4 waves per block)
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, d, o, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)blocks * 4 * iters * 32 * 16384.0;
        // cycles per MFMA per SIMD at full rate = 16: implied clock = mfma per SIMD / time * 16
        const double per_simd = (double)wps * iters * 32;
        if (rep == 3)
          printf("%s operands, %d wave(s) per SIMD: %8.1f TFLOP/s  (%.1f ms; clock if the pipe never idles: %.2f GHz)\n",
                 random ? "random" : "zero  ", wps, flop / ms / 1e9, ms, per_simd * 16 / (ms * 1e6));
      }
    }
  }
  return 0;
}
