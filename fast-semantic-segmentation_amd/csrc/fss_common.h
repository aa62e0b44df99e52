// Shared helpers for the gfx950 kernels: error plumbing, dtype helpers,
// 16-byte vector access.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "fss.h"

namespace fss {

void set_error(const std::string& msg);
// Resolves, for the current device, the device address of the static scratch array behind
// fss_avgpool_pyramid_nhwc_static (defined next to the array, called by fss_set_device).
int pyramid_scratch_init();

#define FSS_CHECK_HIP(expr)                                                    \
  do {                                                                         \
    hipError_t e_ = (expr);                                                    \
    if (e_ != hipSuccess) {                                                    \
      ::fss::set_error(std::string(#expr) + ": " + hipGetErrorString(e_));     \
      return 1;                                                                \
    }                                                                          \
  } while (0)

#define FSS_REQUIRE(cond, msg)                                                 \
  do {                                                                         \
    if (!(cond)) {                                                             \
      ::fss::set_error(std::string(__func__) + ": " + (msg));                  \
      return 2;                                                                \
    }                                                                          \
  } while (0)

#define FSS_LAUNCH_CHECK()                                                     \
  do {                                                                         \
    hipError_t e_ = hipGetLastError();                                         \
    if (e_ != hipSuccess) {                                                    \
      ::fss::set_error(std::string(__func__) + " launch: " +                   \
                       hipGetErrorString(e_));                                 \
      return 3;                                                                \
    }                                                                          \
  } while (0)

typedef unsigned short bf16_t;  // storage type of bfloat16

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((unsigned)v) << 16);
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  // hipcc lowers the cast to v_cvt_pk_bf16_f32 (round to nearest even).
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}

// two floats -> two bf16 in one dword (lo in bits 0-15): one v_cvt_pk_bf16_f32
typedef float fss_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 fss_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  fss_f32x2 f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, fss_bf16x2));
}

// IEEE half precision: the compiler's own type as the storage tag (distinct from bf16_t for the
// template machinery; two bytes).  Conversions round to nearest even (v_cvt_pk_f16_f32).
typedef _Float16 f16_t;
// Tag of the convolution mode FSS_F16_W2: half-precision activations (stored as f16_t) against
// weights kept as two f16 numbers each.  Never the type of a tensor.
struct f16w_t { _Float16 v; };
typedef _Float16 fss_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_f16x2(float lo, float hi) {
  fss_f32x2 f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, fss_f16x2));
}

// What the kernels need to know about a 16-bit storage type: two floats -> one dword and back,
// the bit pattern of the pair (1.0, 1.0), one value -> its bits.
template <typename T> struct Half16;
template <> struct Half16<bf16_t> {
  static constexpr unsigned kOnePair = 0x3f803f80u;
  static __device__ __forceinline__ unsigned pack2(float lo, float hi) { return pack_bf16x2(lo, hi); }
  static __device__ __forceinline__ float lo(unsigned w) { return __uint_as_float(w << 16); }
  static __device__ __forceinline__ float hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
  static __device__ __forceinline__ unsigned bits(float v) { return f32_to_bf16(v); }
  static __device__ __forceinline__ float value(unsigned b) { return __uint_as_float(b << 16); }
};
template <> struct Half16<f16_t> {
  static constexpr unsigned kOnePair = 0x3c003c00u;
  static __device__ __forceinline__ unsigned pack2(float lo, float hi) { return pack_f16x2(lo, hi); }
  static __device__ __forceinline__ float lo(unsigned w) { return (float)__builtin_bit_cast(fss_f16x2, w)[0]; }
  static __device__ __forceinline__ float hi(unsigned w) { return (float)__builtin_bit_cast(fss_f16x2, w)[1]; }
  static __device__ __forceinline__ unsigned bits(float v) {
    return __builtin_bit_cast(unsigned short, (f16_t)v);
  }
  static __device__ __forceinline__ float value(unsigned b) {
    return (float)__builtin_bit_cast(f16_t, (unsigned short)b);
  }
};

// One 16x16x32 MFMA step on 8 + 8 values of a 16-bit type held as raw 16-byte vectors.
typedef float fss_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 fss_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 fss_f16x8 __attribute__((ext_vector_type(8)));
template <typename T>
__device__ __forceinline__ fss_f32x4 mfma_16x16x32(const uint4& a, const uint4& b, fss_f32x4 acc) {
  if constexpr (sizeof(T) == 2 && __is_same(T, f16_t))
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(fss_f16x8, a), __builtin_bit_cast(fss_f16x8, b),
                                                  acc, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(fss_bf16x8, a),
                                                   __builtin_bit_cast(fss_bf16x8, b), acc, 0, 0, 0);
}

// Split-bfloat16 tensors (dtype FSS_F32_BF16X3): one 4-byte slot per element, like float, but
// every 8 consecutive channels (32 bytes) hold their 8 bf16 high parts followed by their 8 bf16
// remainders: v = hi + lo with hi = bf16(v), lo = bf16(v - hi), about 16 significant bits.  The
// halves of such a group are ready-made operands of the bf16 matrix cores (a_hi*b_hi + a_hi*b_lo
// + a_lo*b_hi), so a convolution reading them splits nothing in its main loop; the producer
// splits each value once when it stores it.  The tag type stands for one slot.
struct f32s_t { float slot; };

// 8 floats -> their bf16 high parts and bf16 remainders, packed 8 per uint4 in element order.
// Round to nearest even, so that the remainders (and the dropped lo*lo terms) have no
// systematic sign: truncation would bias every product the same way.
__device__ __forceinline__ void split8(const float (&f)[8], uint4& hi, uint4& lo) {
  unsigned h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = f[2 * i], b = f[2 * i + 1];
    h[i] = pack_bf16x2(a, b);
    l[i] = pack_bf16x2(a - __uint_as_float(h[i] << 16), b - __uint_as_float(h[i] & 0xffff0000u));
  }
  hi = make_uint4(h[0], h[1], h[2], h[3]);
  lo = make_uint4(l[0], l[1], l[2], l[3]);
}
// ... and back: the 8 values of a group from its two 16-byte halves
__device__ __forceinline__ void join8(const uint4& hi, const uint4& lo, float (&f)[8]) {
  const unsigned h[4] = {hi.x, hi.y, hi.z, hi.w}, l[4] = {lo.x, lo.y, lo.z, lo.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(h[i] << 16) + __uint_as_float(l[i] << 16);
    f[2 * i + 1] = __uint_as_float(h[i] & 0xffff0000u) + __uint_as_float(l[i] & 0xffff0000u);
  }
}

// 16-byte store that does not keep the line in the caches (outputs far larger than L2 that
// the same kernel never reads back)
typedef unsigned fss_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16_nt(void* p, unsigned a, unsigned b, unsigned c, unsigned d) {
  fss_u32x4 v = {a, b, c, d};
  __builtin_nontemporal_store(v, reinterpret_cast<fss_u32x4*>(p));
}

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
  static constexpr int kPerChunk = 4;  // elements per 16-byte chunk
  static __device__ __forceinline__ float load(const float* p) { return *p; }
  static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct ElemTraits<bf16_t> {
  static constexpr int kPerChunk = 8;
  static __device__ __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

template <> struct ElemTraits<f16_t> {
  static constexpr int kPerChunk = 8;
  static __device__ __forceinline__ float load(const f16_t* p) { return (float)*p; }
  static __device__ __forceinline__ void store(f16_t* p, float v) {
    *p = (f16_t)fminf(fmaxf(v, -65504.f), 65504.f);   // saturates instead of overflowing
  }
};

// One element of a split-bfloat16 tensor by its slot address (slow paths only): the group
// starts at the slot address rounded down to 32 bytes, the halves are 16 bytes apart.
template <> struct ElemTraits<f32s_t> {
  static constexpr int kPerChunk = 8;
  static __device__ __forceinline__ float load(const f32s_t* p) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const unsigned short* g = reinterpret_cast<const unsigned short*>(a & ~(uintptr_t)31);
    const int i = (int)((a >> 2) & 7);
    return bf16_to_f32(g[i]) + bf16_to_f32(g[8 + i]);
  }
  static __device__ __forceinline__ void store(f32s_t* p, float v) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    unsigned short* g = reinterpret_cast<unsigned short*>(a & ~(uintptr_t)31);
    const int i = (int)((a >> 2) & 7);
    const bf16_t h = f32_to_bf16(v);
    g[i] = h;
    g[8 + i] = f32_to_bf16(v - bf16_to_f32(h));
  }
};

// A channel chunk of T viewed as floats: 16 bytes (4 floats, 8 bf16 / half values), or the
// 32-byte group of 8 channels of a split-bfloat16 tensor.
template <typename T> struct Chunk;
template <> struct Chunk<float> {
  static constexpr int N = 4;
  float v[4];
  __device__ __forceinline__ void load(const float* p) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  __device__ __forceinline__ void store(float* p) const {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
  __device__ __forceinline__ void store_nt(float* p) const {
    store16_nt(p, __float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
  }
};
template <> struct Chunk<bf16_t> {
  static constexpr int N = 8;
  float v[8];
  __device__ __forceinline__ void load(const bf16_t* p) {
    uint4 t = *reinterpret_cast<const uint4*>(p);
    unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(w[i] << 16);
      v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  __device__ __forceinline__ void store(bf16_t* p) const {
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  }
  __device__ __forceinline__ void store_nt(bf16_t* p) const {
    store16_nt(p, pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
               pack_bf16x2(v[6], v[7]));
  }
};

template <> struct Chunk<f32s_t> {
  static constexpr int N = 8;
  float v[8];
  __device__ __forceinline__ void load(const f32s_t* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    join8(q[0], q[1], v);
  }
  __device__ __forceinline__ void store(f32s_t* p) const {
    uint4 hi, lo;
    split8(v, hi, lo);
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = hi;
    q[1] = lo;
  }
  __device__ __forceinline__ void store_nt(f32s_t* p) const {
    uint4 hi, lo;
    split8(v, hi, lo);
    store16_nt(p, hi.x, hi.y, hi.z, hi.w);
    store16_nt(p + 4, lo.x, lo.y, lo.z, lo.w);
  }
};
template <> struct Chunk<f16_t> {
  static constexpr int N = 8;
  float v[8];
  __device__ __forceinline__ void load(const f16_t* p) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = Half16<f16_t>::lo(w[i]);
      v[2 * i + 1] = Half16<f16_t>::hi(w[i]);
    }
  }
  // (saturating, like every half-precision store of the library)
  static __device__ __forceinline__ unsigned sat2(float a, float b) {
    return Half16<f16_t>::pack2(__builtin_amdgcn_fmed3f(a, -65504.f, 65504.f),
                                __builtin_amdgcn_fmed3f(b, -65504.f, 65504.f));
  }
  __device__ __forceinline__ void store(f16_t* p) const {
    *reinterpret_cast<uint4*>(p) = make_uint4(sat2(v[0], v[1]), sat2(v[2], v[3]), sat2(v[4], v[5]), sat2(v[6], v[7]));
  }
  __device__ __forceinline__ void store_nt(f16_t* p) const {
    store16_nt(p, sat2(v[0], v[1]), sat2(v[2], v[3]), sat2(v[4], v[5]), sat2(v[6], v[7]));
  }
};
// bytes of a channel chunk of T (what the copy kernels move per thread)
template <typename T> struct ChunkBytes { static constexpr int value = Chunk<T>::N * (int)sizeof(T); };

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == FSS_ACT_RELU) return fmaxf(v, 0.f);
  if (act == FSS_ACT_RELU6) return fminf(fmaxf(v, 0.f), 6.f);
  return v;
}

static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace fss
