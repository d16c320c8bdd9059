// Shared device/host helpers for libflairhip (gfx950 / CDNA4 only).
//
// Conventions used by every kernel in this directory:
//   * activations are NHWC, channel pitch a multiple of 16 elements, dtype bf16 or f32
//   * a "k-step" is 32 bytes of channels per pixel (16 bf16 / 8 f32); one lane's MFMA
//     operand fragment is 16 bytes of it (bf16: 8 elems -> one v_mfma_f32_32x32x16_bf16,
//     f32: 4 elems -> four v_mfma_f32_32x32x2_f32), so all byte addressing is dtype-agnostic
//   * wave = 64 lanes, everywhere
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#define FFA_BF16 0
#define FFA_F32 1

#define FFA_OK 0
#define FFA_ERR_ARG (-1)
#define FFA_ERR_UNSUPPORTED (-2)
#define FFA_ERR_WORKSPACE (-3)

#define FFA_BCO_RING 0x1000  // `bco` flag: operand packed for conv3x3_ring_kernel (ffa_conv_plan)
#define FFA_BCO_THIN 0x2000  // `bco` flag: operand packed for conv3x3_thin_kernel (<= 32 input channels, <= 32 rows)
#define FFA_BCO_THIN32 0x4000  // with FFA_BCO_THIN: the operand multiplies 32-channel pixels (8-row tiles; else 16 ch, 16-row tiles)
#define FFA_BCO_STEM 0x8000   // `bco` flag: operand packed for conv7x7_stem_kernel (bf16 7x7 stride 2, <= 8 real input channels, 64 rows)

void ffa_set_error(const char* fmt, ...);
int ffa_check_launch(const char* what);
// kernel timing session (ffa_runtime.hip): true + an event pair when this thread has one open
bool ffa_ktime_next(int tag, hipEvent_t* start, hipEvent_t* stop);
#define FFA_KT_RING16_8x32 1
#define FFA_KT_RING16_16x16 2
#define FFA_KT_RING16_128CO 4
#define FFA_KT_WGRAD64 16

#define FFA_REQUIRE(cond, ...)                 \
  do {                                         \
    if (!(cond)) {                             \
      ffa_set_error(__VA_ARGS__);              \
      return FFA_ERR_ARG;                      \
    }                                          \
  } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 ffa_bf16x8;
typedef __attribute__((ext_vector_type(4))) float ffa_f32x4;
typedef __attribute__((ext_vector_type(16))) float ffa_f32x16;
typedef __attribute__((ext_vector_type(4))) short ffa_s16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t ffa_u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t ffa_u32x2;

struct ffa_bf16 {
  uint16_t v;
};

template <typename T>
struct ElemTraits;
template <>
struct ElemTraits<ffa_bf16> {
  static constexpr int kBytes = 2;
  static constexpr int kPerFrag = 8;   // elements in a 16-byte fragment
  static constexpr int kPerStep = 16;  // elements in a 32-byte k-step
  static constexpr int kDtype = FFA_BF16;
};
template <>
struct ElemTraits<float> {
  static constexpr int kBytes = 4;
  static constexpr int kPerFrag = 4;
  static constexpr int kPerStep = 8;
  static constexpr int kDtype = FFA_F32;
};

__device__ __forceinline__ float ffa_bf16_bits_to_f32(uint16_t b) {
  return __uint_as_float(((uint32_t)b) << 16);
}
// Plain cast: hipcc emits v_cvt_pk_bf16_f32 (round-nearest-even, NaN stays NaN).
__device__ __forceinline__ uint16_t ffa_f32_to_bf16_bits(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ uint32_t ffa_pack_bf16x2(float lo, float hi) {
  return (uint32_t)ffa_f32_to_bf16_bits(lo) | ((uint32_t)ffa_f32_to_bf16_bits(hi) << 16);
}

template <typename T>
__device__ __forceinline__ float ffa_load_elem(const T* p);
template <>
__device__ __forceinline__ float ffa_load_elem<ffa_bf16>(const ffa_bf16* p) {
  return ffa_bf16_bits_to_f32(p->v);
}
template <>
__device__ __forceinline__ float ffa_load_elem<float>(const float* p) {
  return *p;
}
template <typename T>
__device__ __forceinline__ void ffa_store_elem(T* p, float v);
template <>
__device__ __forceinline__ void ffa_store_elem<ffa_bf16>(ffa_bf16* p, float v) {
  p->v = ffa_f32_to_bf16_bits(v);
}
template <>
__device__ __forceinline__ void ffa_store_elem<float>(float* p, float v) {
  *p = v;
}

// 8 consecutive channels <-> 8 floats (one 16-byte bf16 vector or two 16-byte f32 vectors).
template <typename T>
__device__ __forceinline__ void ffa_load8(const T* p, float (&v)[8]);
template <>
__device__ __forceinline__ void ffa_load8<ffa_bf16>(const ffa_bf16* p, float (&v)[8]) {
  uint4 u = *reinterpret_cast<const uint4*>(p);
  v[0] = __uint_as_float(u.x << 16);
  v[1] = __uint_as_float(u.x & 0xffff0000u);
  v[2] = __uint_as_float(u.y << 16);
  v[3] = __uint_as_float(u.y & 0xffff0000u);
  v[4] = __uint_as_float(u.z << 16);
  v[5] = __uint_as_float(u.z & 0xffff0000u);
  v[6] = __uint_as_float(u.w << 16);
  v[7] = __uint_as_float(u.w & 0xffff0000u);
}
template <>
__device__ __forceinline__ void ffa_load8<float>(const float* p, float (&v)[8]) {
  float4 a = reinterpret_cast<const float4*>(p)[0];
  float4 b = reinterpret_cast<const float4*>(p)[1];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <typename T>
__device__ __forceinline__ void ffa_store8(T* p, const float (&v)[8]);
template <>
__device__ __forceinline__ void ffa_store8<ffa_bf16>(ffa_bf16* p, const float (&v)[8]) {
  uint4 u;
  u.x = ffa_pack_bf16x2(v[0], v[1]);
  u.y = ffa_pack_bf16x2(v[2], v[3]);
  u.z = ffa_pack_bf16x2(v[4], v[5]);
  u.w = ffa_pack_bf16x2(v[6], v[7]);
  *reinterpret_cast<uint4*>(p) = u;
}
template <>
__device__ __forceinline__ void ffa_store8<float>(float* p, const float (&v)[8]) {
  reinterpret_cast<float4*>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
  reinterpret_cast<float4*>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
}

__device__ __forceinline__ float ffa_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

static inline int ffa_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long long ffa_cdivll(long long a, long long b) { return (a + b - 1) / b; }
