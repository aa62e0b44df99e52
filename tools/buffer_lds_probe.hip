// Probe: buffer_load_dwordx4 ... lds on gfx950 - in-range lanes copy 16 bytes each into
// consecutive LDS, lanes whose offset is past num_records must deposit zeros.
// build: hipcc -O2 --offload-arch=gfx950 tools/buffer_lds_probe.hip -o gpurun_out/buffer_lds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const void* p, unsigned bytes, unsigned* out) {
  extern __shared__ char smem[];
  for (int i = threadIdx.x; i < 512; i += 64) ((unsigned*)smem)[i] = 0xABABABABu;
  __syncthreads();
  unsigned long long a = (unsigned long long)p;
  i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  unsigned voff = (threadIdx.x & 1) ? 0xFFFFFFF0u : threadIdx.x * 16;     // odd lanes: out of range
  if (threadIdx.x == 62) voff = bytes - 8;                               // straddles the end
  unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(voff), "s"(r), "s"(lds) : "memory", "m0");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = ((unsigned*)smem)[i];
}
int main() {
  const unsigned bytes = 4096;
  std::vector<unsigned> h(bytes / 4);
  for (unsigned i = 0; i < bytes / 4; ++i) h[i] = 0x1000 + i;
  void* d; unsigned* o;
  hipMalloc(&d, bytes); hipMalloc((void**)&o, 1024);
  hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, bytes, o);
  std::vector<unsigned> r(256);
  hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane)
    for (int e = 0; e < 4; ++e) {
      unsigned want = (lane & 1) ? 0u : 0x1000 + lane * 4 + e;
      if (lane == 62) want = e < 2 ? 0x1000 + (bytes - 8) / 4 + e : 0u;
      if (r[lane * 4 + e] != want) { if (bad < 8) printf("lane %d e %d: got %08x want %08x\n", lane, e, r[lane * 4 + e], want); ++bad; }
    }
  printf("buffer_load lds probe: %d mismatches\n", bad);
  return bad != 0;
}
