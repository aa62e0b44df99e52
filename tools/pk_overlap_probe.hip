// Probe: does v_pk_fma_f32 give a wrong high half when its destination pair is also the
// source pair whose LOW register is broadcast to both halves (op_sel_hi:[0,1,1])?
//
//   v_pk_fma_f32 v[d:d+1], v[d:d+1], v[b:b+1], v[c:c+1] op_sel_hi:[0,1,1]
//     lo = v[d] * v[b]   + v[c]
//     hi = v[d] * v[b+1] + v[c+1]      <- reads v[d], which the low half overwrites
//
// hipcc's SLP vectorizer emits exactly this form (pyramid_add_kernel, one instance per
// pixel: the 4th channel pair of each 16-byte chunk).  The victim kernel executes the
// overlapping form and a non-overlapping control form on the same operands and counts
// lanes whose high half differs from the scalar result; it runs alone, beside an MFMA
// kernel and beside a streaming-copy kernel on a second stream.
//
// Build: hipcc -O2 --offload-arch=gfx950 tools/pk_overlap_probe.hip -o tools/pk_overlap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { \
  fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

// counts[0..63]: wrong high halves of the overlapping form per lane; [64..127]: of the control
__global__ void __launch_bounds__(256) victim(const float4* __restrict__ src, size_t n_src,
                                              unsigned long long* counts, int iters) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  unsigned bad_ov = 0, bad_ct = 0;
  size_t at = tid % n_src;
  for (int it = 0; it < iters; ++it) {
    // loads in flight while the packed ops issue, as in the real kernel
    const float4 q = src[at];
    at += (size_t)gridDim.x * blockDim.x;
    if (at >= n_src) at -= n_src;
    const float lx = q.x;
    const f2 b = {q.y, q.z};
    const f2 c = {q.w, q.x + q.y};
    const float want_lo = __builtin_fmaf(lx, b.x, c.x);
    const float want_hi = __builtin_fmaf(lx, b.y, c.y);
    f2 d = {lx, 123.f};
    asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[0,1,1]" : "+v"(d) : "v"(b), "v"(c));
    f2 a = {lx, 123.f}, e;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=&v"(e) : "v"(a), "v"(b), "v"(c));
    if (__float_as_uint(d.y) != __float_as_uint(want_hi) || __float_as_uint(d.x) != __float_as_uint(want_lo)) ++bad_ov;
    if (__float_as_uint(e.y) != __float_as_uint(want_hi) || __float_as_uint(e.x) != __float_as_uint(want_lo)) ++bad_ct;
  }
  if (bad_ov) atomicAdd(&counts[lane], (unsigned long long)bad_ov);
  if (bad_ct) atomicAdd(&counts[64 + lane], (unsigned long long)bad_ct);
}

// keeps the matrix pipe and the VGPR read ports of every CU busy for `iters` rounds
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
  if (s == 12345.678f) sink[0] = s;       // never true; keeps the loop alive
}

__global__ void __launch_bounds__(256) copy_aggressor(const float4* __restrict__ x, float4* __restrict__ y,
                                                      size_t n, int rounds) {
  for (int r = 0; r < rounds; ++r)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
      y[i] = x[i];
}

int main() {
  const size_t n_src = (size_t)1 << 22;          // 64 MB of float4
  std::vector<float> h(n_src * 4);
  unsigned s = 12345u;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (float)((s >> 8) & 0xffff) / 4096.f - 8.f; }
  float4 *src, *cpx, *cpy;
  float* sink;
  unsigned long long* counts;
  CK(hipMalloc(&src, n_src * 16));
  CK(hipMalloc(&cpx, n_src * 16));
  CK(hipMalloc(&cpy, n_src * 16));
  CK(hipMalloc(&sink, 64));
  CK(hipMalloc(&counts, 128 * 8));
  CK(hipMemcpy(src, h.data(), n_src * 16, hipMemcpyHostToDevice));
  CK(hipMemset(cpx, 0, n_src * 16));
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  const char* names[3] = {"alone", "beside mfma kernel", "beside copy kernel"};
  for (int mode = 0; mode < 3; ++mode) {
    CK(hipMemset(counts, 0, 128 * 8));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // aggressors occupy part of every CU (2 blocks of 4 waves per CU) for the whole run
    if (mode == 1) hipLaunchKernelGGL(mfma_aggressor, dim3(512), dim3(256), 0, s1, sink, 400000);
    if (mode == 2) hipLaunchKernelGGL(copy_aggressor, dim3(512), dim3(256), 0, s1, cpx, cpy, n_src, 40);
    CK(hipEventRecord(e0, s0));
    for (int rep = 0; rep < 20; ++rep)
      hipLaunchKernelGGL(victim, dim3(4096), dim3(256), 0, s0, src, n_src, counts, 64);
    CK(hipEventRecord(e1, s0));
    CK(hipGetLastError());
    CK(hipStreamSynchronize(s0));
    float ms_v = 0.f;
    CK(hipEventElapsedTime(&ms_v, e0, e1));
    const hipError_t still = hipStreamQuery(s1);   // hipErrorNotReady: the aggressor outlived the victims
    CK(hipDeviceSynchronize());
    unsigned long long hc[128];
    CK(hipMemcpy(hc, counts, sizeof(hc), hipMemcpyDeviceToHost));
    unsigned long long ov = 0, ct = 0;
    for (int i = 0; i < 64; ++i) { ov += hc[i]; ct += hc[64 + i]; }
    printf("%-20s victims %.2f ms, aggressor %s: overlapping form wrong %llu, control wrong %llu of %llu\n",
           names[mode], ms_v, mode == 0 ? "-" : (still == hipErrorNotReady ? "still running" : "finished early"),
           ov, ct, 20ull * 4096 * 256 * 64);
    if (ov) {
      printf("  wrong lanes:");
      for (int i = 0; i < 64; ++i) if (hc[i]) printf(" %d:%llu", i, hc[i]);
      printf("\n");
    }
    (void)hipGetLastError();
  }
  return 0;
}
