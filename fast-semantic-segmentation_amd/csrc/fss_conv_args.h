// Argument block and LDS-DMA helpers shared by the convolution kernels.
#pragma once
#include <type_traits>

#include "fss_common.h"

namespace fss {

struct ConvArgs {
  const void* x;
  const void* x2;   // second input (1x1, stride 1, the output's spatial size), or null
  const void* w;
  const float* bias;
  const void* res;
  void* y;
  int N, H, W, Cin, Cout, KH, KW, stride, dil, pad_t, pad_l, Ho, Wo, act;
  int Cin2;         // channels of x2: its K segment follows the filter taps in the packed weights
  int M, K, Kpad, y_ld, res_ld, tiles_m, tiles_n;
  // split-K launches: the grid holds `splits` blocks per output tile, block z of a tile walks
  // its share of the tile's K tiles and stores raw sums into slab z of y ([splits][M][y_ld])
  int splits;
  // classifier head (HEAD kernels only): logits = act(conv) * w_head^T + bias_head
  const void* w_head;       // bf16 [32][Cout], rows >= cout_head zero
  const float* bias_head;   // [cout_head]
  int cout_head;
  // residual of the HEAD kernels given as the x2 zoom of a half-resolution tensor zy[N][zH][zW][Cout]
  // (+ the border substitutes of fss_zoom2_conv_commute), interpolated in the epilogue
  const void* zy;
  const void* zytb;
  const void* zylr;
  const void* zycn;
  int zH, zW;
};


// One LDS-DMA piece: 64 lanes x 16 bytes from per-lane global addresses to the
// 1 KiB of LDS starting at the wave-uniform byte address `lds_dst`.  M0 (the DMA's
// LDS base) is written and consumed inside the one statement and declared clobbered,
// so the compiler never keeps a value of its own (s_movrel / gpr-index setup) in it
// across the statement.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, off"
      :
      : "v"(gsrc), "s"(lds_dst)
      : "memory", "m0");
}

// Buffer addressing of the LDS-DMA: a 128-bit resource (base, size) in SGPRs and one 32-bit byte
// offset per lane.  A lane whose offset lies past the end of the buffer deposits ZEROS, which is
// how padding taps, rows past M and K tiles past the last one are filled: no zero page and no
// 64-bit address arithmetic per piece, one add and one select.  Every block describes only the
// window of a tensor it can touch (its images / its rows / its weight rows), so the 32-bit
// offsets hold for tensors of any size.
typedef int fss_i32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kDmaOutOfRange = 0xfffffff0u;     // past the end of every resource below
__device__ __forceinline__ fss_i32x4 dma_resource(const void* base, unsigned long long bytes) {
  const unsigned long long a = (unsigned long long)base;
  const unsigned n = bytes > 0xffffff00ull ? 0xffffff00u : (unsigned)bytes;
  fss_i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));   // stride 0: a raw buffer
  r.z = __builtin_amdgcn_readfirstlane((int)n);
  r.w = 0x00020000;                                                 // 32-bit raw data format
  return r;
}
// One piece: 64 lanes x 16 bytes from base + voff (per lane) to the 1 KiB of LDS at lds_dst.
__device__ __forceinline__ void blds16(unsigned voff, fss_i32x4 rsrc, unsigned lds_dst) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %0, %1, 0 offen lds"
      :
      : "v"(voff), "s"(rsrc), "s"(lds_dst)
      : "memory", "m0");
}

// Orders a wave's own LDS writes before its own later LDS reads (and the reverse): the LDS serves
// one wave's instructions in issue order, so data staged and read back by the SAME wave needs no
// workgroup barrier, only a point the compiler does not move memory operations across.
__device__ __forceinline__ void wave_lds_fence() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt out of range");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}


// f(integral_constant<int, I>) for I = FIRST .. LAST - 1, in order
template <int FIRST, int LAST, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (FIRST < LAST) {
    f(std::integral_constant<int, FIRST>{});
    static_for<FIRST + 1, LAST>(f);
  }
}

// The 256x256, 8-wave, 8-phase form of the bf16 convolution (fss_conv8.hip).  Returns -1 when the
// layer is outside what it covers (the caller then uses another configuration of the same MFMA
// shape, whose results are bit-identical).
int launch_conv8_bf16(ConvArgs& a, hipStream_t stream);

}  // namespace fss
