// Shared device helpers for libcase_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "case_hip.h"

typedef unsigned short bf16_t;  // raw bf16 bits

int case_set_error(int code, const char* fmt, ...);
int case_check_launch(const char* what);
int case_device_cus();      // compute units of the current device (cached)
int case_persistent_cus();  // ... minus the reserved ones, rounded down to whole XCD rounds: the grid of the persistent kernels

#define CASE_REQUIRE(cond, ...)                          \
  do {                                                   \
    if (!(cond)) return case_set_error(CASE_E_ARG, __VA_ARGS__); \
  } while (0)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even, NaN stays a quiet NaN: gfx950's v_cvt_pk_bf16_f32 (one instruction per pair; the integer
// formulation costs ~7 VALU per element, which was a third of the GEMM epilogue at one wave per SIMD)
typedef float case_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 case_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t f32x2_to_bf16x2(float lo, float hi) {
  const case_f32x2 v = {lo, hi};
  const case_bf16x2 b = __builtin_convertvector(v, case_bf16x2);
  return *reinterpret_cast<const uint32_t*>(&b);
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(f32x2_to_bf16x2(f, 0.f) & 0xffffu); }

template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// 16-byte vector access: 4 x f32 or 8 x bf16 <-> float lanes
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static constexpr int N = 4;
  typedef uint4 raw;
  static __device__ __forceinline__ void unpack(const uint4& t, float (&v)[4]) {
    v[0] = __uint_as_float(t.x); v[1] = __uint_as_float(t.y); v[2] = __uint_as_float(t.z); v[3] = __uint_as_float(t.w);
  }
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};
template <> struct Vec16<bf16_t> {
  static constexpr int N = 8;
  typedef uint4 raw;
  static __device__ __forceinline__ void unpack(const uint4& t, float (&v)[8]) {
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(w[i] << 16);
      v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
#ifdef CASE_VEC16_NT
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
    typedef unsigned int u32x4_nt __attribute__((ext_vector_type(4)));
    const u32x4_nt t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt*>(p));
    unpack(make_uint4(t[0], t[1], t[2], t[3]), v);
  }
#else
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) { unpack(*reinterpret_cast<const uint4*>(p), v); }
#endif
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = f32x2_to_bf16x2(v[2 * i], v[2 * i + 1]);
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  }
};

// 8-byte vector access for bf16 rows whose width is a multiple of 256 but not of 512 elements (64 lanes x 4): widths 256, 768, 1280 of the
// reference's default geometry and of cfg 5
template <typename T> struct Vec8;
template <> struct Vec8<bf16_t> {
  static constexpr int N = 4;
  typedef uint2 raw;
  static __device__ __forceinline__ void unpack(const uint2& t, float (&v)[4]) {
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
  }
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[4]) { unpack(*reinterpret_cast<const uint2*>(p), v); }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[4]) {
    *reinterpret_cast<uint2*>(p) = make_uint2(f32x2_to_bf16x2(v[0], v[1]), f32x2_to_bf16x2(v[2], v[3]));
  }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide reductions for <= 1024 threads; `red` is a 32-float LDS scratch
__device__ __forceinline__ float block_sum(float v, float* red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = (lane < nw) ? red[lane] : 0.f;
  t = wave_sum(t);
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_max(v);
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = (lane < nw) ? red[lane] : -INFINITY;
  t = wave_max(t);
  return t;
}

// ABI 600: the per-step base of the dropout counter stream may live in caller-owned device memory (CaseStepState.rng_base) so that a
// captured hipGraph draws new masks on every replay.  The pointer is a kernel argument (uniform) and nobody writes the struct while a
// kernel that reads it runs, so the load goes through the CONSTANT address space: one s_load_dwordx2 into scalar registers, no VGPR.
__device__ __forceinline__ uint64_t rng_base_of(const CaseStepState* st) {
  if (st == nullptr) return 0;
  typedef const __attribute__((address_space(4))) uint64_t* const_u64_ptr;
  return *(const_u64_ptr)(uintptr_t)st;  // rng_base is the first member
}

// Counter-based RNG: uniform in [0,1) from (seed, 64-bit element index), keyed by element index so forward and backward
// (and the fused / unfused attention paths) regenerate the same keep mask.  One 32-bit hash serves TWO consecutive elements
// (16 bits each, element index >> 1 is hashed): 32-bit integer multiplies are quarter rate on gfx950 and the two rounds of
// multiply-xorshift per element were costing the dropout GEMM epilogues as much as a K = 512 main loop.  A keep test
// `u >= p` therefore resolves p to 1/65536.  rng_uniform() is the per-element definition; rng_uniform2() returns the pair
// (idx_even, idx_even + 1) from one hash for the vectorised kernels.
__device__ __forceinline__ uint32_t rng_hash(uint64_t seed, uint64_t pair) {
  const uint32_t lo = (uint32_t)pair, hi = (uint32_t)(pair >> 32);
  uint32_t h = lo ^ ((hi << 16) | (hi >> 16)) ^ (uint32_t)seed;
  h *= 0x9E3779B1u;
  h ^= (h >> 15) ^ (uint32_t)(seed >> 32);
  h *= 0x85EBCA77u;
  h ^= h >> 13;
  return h;
}
__device__ __forceinline__ float rng_uniform(uint64_t seed, uint64_t idx) {
  const uint32_t h = rng_hash(seed, idx >> 1);
  return (float)((idx & 1) ? (h >> 16) : (h & 0xffffu)) * (1.0f / 65536.0f);
}
__device__ __forceinline__ void rng_uniform2(uint64_t seed, uint64_t idx_even, float& u0, float& u1) {
  const uint32_t h = rng_hash(seed, idx_even >> 1);
  u0 = (float)(h & 0xffffu) * (1.0f / 65536.0f);
  u1 = (float)(h >> 16) * (1.0f / 65536.0f);
}
// keep-or-zero for 8 consecutive elements starting at idx0 (any parity)
__device__ __forceinline__ void dropout8(float (&x)[8], uint64_t seed, uint64_t idx0, float p, float scale) {
  if ((idx0 & 1) == 0) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      float u0, u1;
      rng_uniform2(seed, idx0 + e, u0, u1);
      x[e] = u0 >= p ? x[e] * scale : 0.f;
      x[e + 1] = u1 >= p ? x[e + 1] * scale : 0.f;
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = rng_uniform(seed, idx0 + e) >= p ? x[e] * scale : 0.f;
  }
}

// ---- dropout of ATTENTION PROBABILITIES (softmax kernels and the fused attention kernels draw the same mask) -------------
// Element (row, col) of the [rows, Lk] probability matrix, row = (n * heads + head) * Lq + q.  A row key -- one full hash of
// (seed, offset + row) -- is computed once per row; the two 16-bit uniforms of the column pair (2 j, 2 j + 1) come from ONE
// multiply: h = (key ^ j C1) C2, h ^= h >> 16, low half = even column, high half = odd column.  keep <=> field >= ceil(p 65536).
// The fused kernels get 4-7 vector instructions per element out of this instead of ~15 (64-bit element indices, two
// quarter-rate multiplies per pair): they are VALU-bound at head_dim 64.  (j C1 is a per-lane or scalar constant there: the
// multiplication distributes over the sum of a uniform and a per-lane part.)  Statistics checked offline: mean, row / column
// dispersion and neighbour correlations at distances 1..8 are at the sampling-noise level; dropping the C1 pre-multiply is NOT
// (correlation 0.1 at distances 2, 4, 8).
constexpr uint32_t RNG_C1 = 0x9E3779B1u, RNG_C2 = 0x85EBCA77u;
__device__ __forceinline__ uint32_t rng_row_key(uint64_t seed, uint64_t offset_plus_row) { return rng_hash(seed, offset_plus_row); }
__device__ __forceinline__ uint32_t rng_pair_bits_pre(uint32_t key, uint32_t jc1) {  // jc1 = pair index * RNG_C1
  const uint32_t h = (key ^ jc1) * RNG_C2;
  return h ^ (h >> 16);
}
__device__ __forceinline__ uint32_t rng_pair_bits(uint32_t key, uint32_t pair) { return rng_pair_bits_pre(key, pair * RNG_C1); }
__device__ __forceinline__ uint32_t rng_threshold(float p) { return (uint32_t)ceilf(p * 65536.f); }
__device__ __forceinline__ bool attn_keep(uint32_t key, uint32_t col, uint32_t thr) {
  const uint32_t h = rng_pair_bits(key, col >> 1);
  return ((col & 1) ? (h >> 16) : (h & 0xffffu)) >= thr;
}
// keep-or-zero for 8 consecutive columns starting at the EVEN column c0
__device__ __forceinline__ void attn_dropout8(float (&x)[8], uint32_t key, uint32_t c0, uint32_t thr, float scale) {
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const uint32_t h = rng_pair_bits(key, (c0 + e) >> 1);
    x[e] = (h & 0xffffu) >= thr ? x[e] * scale : 0.f;
    x[e + 1] = (h >> 16) >= thr ? x[e + 1] * scale : 0.f;
  }
}

// GELU (erf form, as F.gelu) and its derivative.  Phi(x) = 0.5 (1 + erf(x / sqrt 2)) through Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7 on erf): with z = |x| / sqrt 2, t = 1 / (1 + p z), q = (a1 t + ... + a5 t^5) exp(-z^2) the tail
// 1 - erf(z) IS q, so Phi(-|x|) = q / 2 carries no cancellation, and exp(-z^2) = exp(-x^2 / 2) is also the density the
// derivative needs.  Two transcendentals (v_rcp, v_exp) + 10 FMAs; libm's erff is ~3x that, and in a GEMM epilogue at one wave
// per SIMD nothing hides it.
struct GeluTerms {
  float cdf, pdf;
};
__device__ __forceinline__ GeluTerms gelu_terms(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
  const float e = __expf(-z * z);
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float q = 0.5f * p * t * e;  // Phi(-|x|)
  GeluTerms r;
  r.cdf = x >= 0.f ? 1.f - q : q;
  r.pdf = 0.39894228040143268f * e;
  return r;
}
__device__ __forceinline__ float gelu_f(float x) { return x * gelu_terms(x).cdf; }
__device__ __forceinline__ float dgelu_f(float x) {
  const GeluTerms g = gelu_terms(x);
  return g.cdf + x * g.pdf;
}

static inline int grid_for(int64_t n, int block, int per_thread = 1, int cap = 256 * 8) {
  int64_t g = (n + (int64_t)block * per_thread - 1) / ((int64_t)block * per_thread);
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}
