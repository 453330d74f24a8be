// K7 additive-attention scores (fused tanh, never materialising [B,T,S,H]), K11 copy/pointer
// scatter-add, K12 NLL gather, K13 row argmax.
//
// K7 is transcendental-issue bound (B*T*S*H tanh per pass; 2.5 G at cfg2), not MFMA or HBM work:
//   forward   lanes = source position j (the H-sum stays in a register), uh tile transposed through LDS
//   backward  two sweeps with lanes = h (coalesced uh reads, no cross-lane reductions, no atomics on the
//             large tensors): sweep 1 owns (j, h) and sums over t -> d_uh and the d_v partials;
//             sweep 2 owns (t, h) and sums over j -> d_wq.  tanh is recomputed in both.
#include "common.h"

namespace {

template <bool FAST>
__device__ __forceinline__ float tanh_t(float x) {
  if constexpr (FAST) {
    // 1 - 2 / (exp(2x) + 1); saturates cleanly for |x| large (exp -> inf gives 1, exp -> 0 gives -1)
    const float e = __expf(2.f * x);
    return 1.f - __fdividef(2.f, e + 1.f);
  } else {
    return tanhf(x);
  }
}

// Fast path (bf16 operands): with xs = 2 log2(e) (w + u) folded into the operands when they are staged,
//   r = 1 / (2^xs + 1),  tanh = 1 - 2 r,  1 - tanh^2 = 4 r (1 - r):
// one add, v_exp, one add, v_rcp per element; the affine parts (1 - 2 r, the factor 4) are applied to the sums.
// Round 5: the exponential is FACTORED, 2^(w + u) = 2^w 2^u: 2^{ws} is taken once per (target row, feature) and 2^{us} once per (source
// position, feature) where the operands are staged, so the inner loops pay one FMA and ONE quarter-rate instruction (v_rcp) per element
// instead of two (v_exp + v_rcp): 28 -> 16 issue cycles per element in the forward sweep.  Each prescaled argument is clamped at
// +-EXP2_CLAMP so that the product of the two factors stays finite and normal (2^+-124); that alters tanh(wq + uh) only where |wq| or
// |uh| exceeds 21.5 (bf16 fast path only; the f32 parity path calls tanhf).
constexpr float TANH_PRESCALE = 2.8853900817779268f;  // 2 / ln 2
constexpr float EXP2_CLAMP = 62.f;
__device__ __forceinline__ float exp2_factor(float xs) { return __builtin_amdgcn_exp2f(fminf(fmaxf(xs, -EXP2_CLAMP), EXP2_CLAMP)); }
__device__ __forceinline__ float half_sigmoid_prod(float ew, float eu) { return __builtin_amdgcn_rcpf(fmaf(ew, eu, 1.f)); }

constexpr int AJ = 64;   // source positions per workgroup (forward)
constexpr int AH = 64;   // h chunk staged in LDS
constexpr int ATW = 8;   // target rows per wave per chunk (4 waves -> 32 per chunk)

template <typename T, bool FAST>
__global__ __launch_bounds__(256) void additive_fwd_kernel(const float* __restrict__ wq, const T* __restrict__ uh,
                                                           const float* __restrict__ v, float* __restrict__ s, int64_t Tn,
                                                           int64_t S, int64_t H) {
  __shared__ float U[AJ][AH + 1];
  __shared__ float W[4 * ATW][AH];
  __shared__ float V[AH];
  const int64_t b = blockIdx.y, j0 = (int64_t)blockIdx.x * AJ;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t tc = (int64_t)blockIdx.z * 4 * ATW; tc < Tn; tc += (int64_t)gridDim.z * 4 * ATW) {  // (target chunks over blockIdx.z: short memories)
    constexpr float PS = FAST ? TANH_PRESCALE : 1.f;
    float acc[ATW];
#pragma unroll
    for (int i = 0; i < ATW; ++i) acc[i] = 0.f;
    float vsum = 0.f;
    for (int64_t hc = 0; hc < H; hc += AH) {
      __syncthreads();
      for (int e = threadIdx.x; e < AJ * AH; e += 256) {
        const int jj = e / AH, hh = e % AH;
        const int64_t j = j0 + jj, h = hc + hh;
        const float uv = (j < S && h < H) ? PS * Elem<T>::ld(uh + (b * S + j) * H + h) : 0.f;
        U[jj][hh] = FAST ? exp2_factor(uv) : uv;
      }
      for (int e = threadIdx.x; e < 4 * ATW * AH; e += 256) {
        const int tt = e / AH, hh = e % AH;
        const int64_t t = tc + tt, h = hc + hh;
        const float wv = (t < Tn && h < H) ? PS * wq[(b * Tn + t) * H + h] : 0.f;
        W[tt][hh] = FAST ? exp2_factor(wv) : wv;
      }
      if (threadIdx.x < AH) V[threadIdx.x] = (hc + threadIdx.x < H) ? v[hc + threadIdx.x] : 0.f;
      __syncthreads();
#pragma unroll 4
      for (int hh = 0; hh < AH; ++hh) {
        const float u = U[lane][hh], vv = V[hh];
        if constexpr (FAST) {
          vsum += vv;  // sum_h v_h (1 - 2 r_h) = sum_h v_h - 2 sum_h v_h r_h
#pragma unroll
          for (int i = 0; i < ATW; ++i) acc[i] += vv * half_sigmoid_prod(W[wave * ATW + i][hh], u);
        } else {
#pragma unroll
          for (int i = 0; i < ATW; ++i) acc[i] += vv * tanh_t<FAST>(W[wave * ATW + i][hh] + u);
        }
      }
    }
    const int64_t j = j0 + lane;
    if (j < S) {
#pragma unroll
      for (int i = 0; i < ATW; ++i) {
        const int64_t t = tc + wave * ATW + i;
        if (t < Tn) s[(b * Tn + t) * S + j] = FAST ? vsum - 2.f * acc[i] : acc[i];
      }
    }
  }
}

// Few target rows (greedy decoding: T = 1 per step): one wave per source position j, lanes = h (16-byte coalesced read of
// the uh row, no LDS transpose), shuffle reduction of the H-sum.  The tiled kernel above would spend 31/32 of its tanh work
// on absent target rows; this one is bound by streaming uh once (B*S*H*esz bytes).
template <typename T, bool FAST, int TT, int NCH>
__global__ __launch_bounds__(256) void additive_fwd_rowwise_kernel(const float* __restrict__ wq, const T* __restrict__ uh,
                                                                   const float* __restrict__ v, float* __restrict__ s,
                                                                   int64_t Tn, int64_t S, int64_t H) {
  constexpr int E = Vec16<T>::N;
  const int64_t b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  // this lane's slice of the query projection and of v stays in registers for every source position
  float wreg[NCH][TT][E], vreg[NCH][E];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int64_t h0 = ((int64_t)c * 64 + lane) * E;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      vreg[c][e] = h0 < H ? v[h0 + e] : 0.f;
#pragma unroll
      for (int t = 0; t < TT; ++t) wreg[c][t][e] = (h0 < H && t < Tn) ? wq[(b * Tn + t) * H + h0 + e] : 0.f;
    }
  }
  for (int64_t j = wave; j < S; j += nwaves) {
    float acc[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) acc[t] = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int64_t h0 = ((int64_t)c * 64 + lane) * E;
      if (h0 < H) {
        float u[E];
        Vec16<T>::load(uh + (b * S + j) * H + h0, u);
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
          for (int t = 0; t < TT; ++t) acc[t] += vreg[c][e] * tanh_t<FAST>(wreg[c][t][e] + u[e]);
      }
    }
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      const float r = wave_sum(acc[t]);
      if (lane == 0 && t < Tn) s[(b * Tn + t) * S + j] = r;
    }
  }
}

// sweep 1: thread owns h (256 per workgroup), workgroup owns BJ source positions; sums over t.
constexpr int BJ = 16;
constexpr int BTC = 32;  // t chunk staged in LDS

template <typename T, bool FAST>
__global__ __launch_bounds__(256) void additive_bwd_uh_kernel(const float* __restrict__ ds, const float* __restrict__ wq,
                                                              const T* __restrict__ uh, const float* __restrict__ v,
                                                              float* __restrict__ d_uh, float* __restrict__ d_v, int64_t Tn,
                                                              int64_t S, int64_t H) {
  __shared__ float DS[BTC][BJ];
  const int64_t b = blockIdx.z, j0 = (int64_t)blockIdx.x * BJ, h = (int64_t)blockIdx.y * 256 + threadIdx.x;
  const bool h_ok = h < H;
  float u[BJ], acc[BJ];
#pragma unroll
  for (int jj = 0; jj < BJ; ++jj) {
    const int64_t j = j0 + jj;
    u[jj] = (h_ok && j < S) ? (FAST ? TANH_PRESCALE : 1.f) * Elem<T>::ld(uh + (b * S + j) * H + h) : 0.f;
    if constexpr (FAST) u[jj] = exp2_factor(u[jj]);  // 2^{us}: the factored form (see exp2_factor)
    acc[jj] = 0.f;
  }
  float dv = 0.f, gsum = 0.f;
  for (int64_t tc = 0; tc < Tn; tc += BTC) {
    __syncthreads();
    for (int e = threadIdx.x; e < BTC * BJ; e += 256) {
      const int tt = e / BJ, jj = e % BJ;
      const int64_t t = tc + tt, j = j0 + jj;
      DS[tt][jj] = (t < Tn && j < S) ? ds[(b * Tn + t) * S + j] : 0.f;
    }
    __syncthreads();
    const int tmax = (int)((Tn - tc) < BTC ? (Tn - tc) : BTC);
    for (int tt = 0; tt < tmax; ++tt) {
      float w = h_ok ? (FAST ? TANH_PRESCALE : 1.f) * wq[(b * Tn + tc + tt) * H + h] : 0.f;
      if constexpr (FAST) w = exp2_factor(w);
#pragma unroll
      for (int jj = 0; jj < BJ; ++jj) {
        const float g = DS[tt][jj];
        if constexpr (FAST) {
          const float r = half_sigmoid_prod(w, u[jj]);
          const float gr = g * r;
          acc[jj] += gr - gr * r;  // g r (1 - r); the factor 4 is applied once at the end
          dv += gr;                // sum g tanh = sum g - 2 sum g r
          gsum += g;
        } else {
          const float th = tanh_t<FAST>(w + u[jj]);
          acc[jj] += g * (1.f - th * th);
          dv += g * th;
        }
      }
    }
  }
  if (h_ok) {
    const float vh = FAST ? 4.f * v[h] : v[h];
#pragma unroll
    for (int jj = 0; jj < BJ; ++jj) {
      const int64_t j = j0 + jj;
      if (j < S) d_uh[(b * S + j) * H + h] = vh * acc[jj];
    }
    atomicAdd(d_v + h, FAST ? gsum - 2.f * dv : dv);
  }
}

// sweep 2: thread owns h, workgroup owns CT target rows; sums over all j.
constexpr int CT = 8;
constexpr int CJC = 64;  // j chunk staged in LDS

template <typename T, bool FAST>
__global__ __launch_bounds__(256) void additive_bwd_wq_kernel(const float* __restrict__ ds, const float* __restrict__ wq,
                                                              const T* __restrict__ uh, const float* __restrict__ v,
                                                              float* __restrict__ d_wq, int64_t Tn, int64_t S, int64_t H,
                                                              int tblocks, int64_t j_per) {
  // blockIdx.x = (source-position chunk, target-row block): with few target rows (T = 40 -> 5 blocks) one workgroup per
  // (t block, h block, b) left the chip at one wave per SIMD; the j range is split and partial sums are added atomically
  // into the pre-zeroed d_wq when there is more than one chunk.
  __shared__ float DS[CT][CJC];
  const int64_t b = blockIdx.z, t0 = (int64_t)(blockIdx.x % tblocks) * CT, h = (int64_t)blockIdx.y * 256 + threadIdx.x;
  const int64_t j_begin = (int64_t)(blockIdx.x / tblocks) * j_per, j_end = (j_begin + j_per < S) ? j_begin + j_per : S;
  const bool split = j_per < S;
  const bool h_ok = h < H;
  float w[CT], acc[CT];
#pragma unroll
  for (int tt = 0; tt < CT; ++tt) {
    const int64_t t = t0 + tt;
    w[tt] = (h_ok && t < Tn) ? (FAST ? TANH_PRESCALE : 1.f) * wq[(b * Tn + t) * H + h] : 0.f;
    if constexpr (FAST) w[tt] = exp2_factor(w[tt]);
    acc[tt] = 0.f;
  }
  for (int64_t jc = j_begin; jc < j_end; jc += CJC) {
    __syncthreads();
    for (int e = threadIdx.x; e < CT * CJC; e += 256) {
      const int tt = e / CJC, jj = e % CJC;
      const int64_t t = t0 + tt, j = jc + jj;
      DS[tt][jj] = (t < Tn && j < j_end) ? ds[(b * Tn + t) * S + j] : 0.f;
    }
    __syncthreads();
    const int jmax = (int)((j_end - jc) < CJC ? (j_end - jc) : CJC);
    for (int jj = 0; jj < jmax; ++jj) {
      float u = h_ok ? (FAST ? TANH_PRESCALE : 1.f) * Elem<T>::ld(uh + (b * S + jc + jj) * H + h) : 0.f;
      if constexpr (FAST) u = exp2_factor(u);
#pragma unroll
      for (int tt = 0; tt < CT; ++tt) {
        if constexpr (FAST) {
          const float r = half_sigmoid_prod(w[tt], u);
          const float gr = DS[tt][jj] * r;
          acc[tt] += gr - gr * r;
        } else {
          const float th = tanh_t<FAST>(w[tt] + u);
          acc[tt] += DS[tt][jj] * (1.f - th * th);
        }
      }
    }
  }
  if (h_ok) {
    const float vh = FAST ? 4.f * v[h] : v[h];
#pragma unroll
    for (int tt = 0; tt < CT; ++tt) {
      const int64_t t = t0 + tt;
      if (t < Tn) {
        if (split) atomicAdd(d_wq + (b * Tn + t) * H + h, vh * acc[tt]);
        else d_wq[(b * Tn + t) * H + h] = vh * acc[tt];
      }
    }
  }
}

// ---- K22 (round 5): the greedy step's additive attention as ONE launch ------------------------------------------------------------
// tanh(a + b) = 1 - 2 / (e^{2a} e^{2b} + 1): with EU = e^{2 uh} cached per (source position, feature) when the memory is encoded (uh is
// the same in all T steps) and EW = e^{2 wq} computed once per (item, feature) per step, an element costs one FMA, ONE reciprocal and
// one FMA -- the exponentials move from O(T S H) to O((T + S) H).  Both arguments are clamped at +-EXP_CLAMP so that the product stays
// finite and normal in f32 (e^{+-86}); that changes tanh(wq + uh) only where |wq| or |uh| exceeds 21.5, far outside what a linear map of
// LayerNorm outputs produces (and there tanh is saturated unless the two nearly cancel).
// One workgroup of 16 waves per item: (1) scores over the EU rows into LDS, (2) masked softmax in LDS (+ the copy prior's
// renormalisation p w / (1e-8 + sum p w), CaSE/Model.py:81-82), (3) ctx = sum_j p_j value_j over the memory rows -- the score sweep, the
// softmax, the cast and the [1, S] x [S, H] product of the four-launch form, and the four ATen launches of the prior.  Rows of masked
// source positions are never loaded.  HBM-bound: S H (2 + 2) bytes per item.
constexpr float EXP_CLAMP = 43.f;
constexpr int PA_WAVES = 16, PA_H = 512;

__global__ __launch_bounds__(256) void additive_key_exp_kernel(const float* __restrict__ uh, bf16_t* __restrict__ eu, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const float4 a = reinterpret_cast<const float4*>(uh)[2 * i], b = reinterpret_cast<const float4*>(uh)[2 * i + 1];
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    float y[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] = __expf(fminf(fmaxf(2.f * x[e], -EXP_CLAMP), EXP_CLAMP));
    Vec16<bf16_t>::store(eu + 8 * i, y);
  }
}

// sums of four per-lane values over the 64 lanes, all at once: two v_permlane32_swap + one v_permlane16_swap fold the wave's four
// 16-lane rows so that lane row g keeps value g, four rotate-adds inside the row finish it.  Returns the total of r<g> in every lane of
// lane row g (lanes 16 g .. 16 g + 15).
__device__ __forceinline__ float row_sum4(float r0, float r1, float r2, float r3) {
  const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(r0), __float_as_uint(r2), false, false);  // [r0.lo | r2.lo], [r0.hi | r2.hi]
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(r1), __float_as_uint(r3), false, false);
  const float A = __uint_as_float(a[0]) + __uint_as_float(a[1]);  // lanes 0-31: r0 over {l, l + 32}; lanes 32-63: r2
  const float B = __uint_as_float(b[0]) + __uint_as_float(b[1]);  // r1 | r3
  const auto c = __builtin_amdgcn_permlane16_swap(__float_as_uint(A), __float_as_uint(B), false, false);
  float t = __uint_as_float(c[0]) + __uint_as_float(c[1]);        // lane row 0: r0, 1: r1, 2: r2, 3: r3 (16 partial sums each)
  t += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(t), 0x128, 0xf, 0xf, false));  // row_ror:8
  t += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(t), 0x124, 0xf, 0xf, false));  // row_ror:4
  t += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(t), 0x122, 0xf, 0xf, false));  // row_ror:2
  t += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(t), 0x121, 0xf, 0xf, false));  // row_ror:1
  return t;
}

// K22's two sweeps (e^{2 uh} rows, then the value rows: 1 GB each at cfg 4, every byte once per launch) are read NON-TEMPORAL for the same
// reason as K21's stream (attn_mqa.hip): the step's reusable data stays in the Infinity Cache.  -DCASE_STREAM_DEFAULT_POLICY: default policy (A/B).
#ifndef CASE_STREAM_DEFAULT_POLICY
typedef unsigned int pa_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 pa_nt_load(const void* p) {
  const pa_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const pa_u32x4*>(p));
  return make_uint4(v[0], v[1], v[2], v[3]);
}
#define PA_STREAM_LOAD(P) pa_nt_load(P)
#else
#define PA_STREAM_LOAD(P) (*reinterpret_cast<const uint4*>(P))
#endif
__global__ __launch_bounds__(64 * PA_WAVES) void pointer_attend_decode_kernel(
    const float* __restrict__ wq, const float* __restrict__ wq_add, const bf16_t* __restrict__ eu, const float* __restrict__ v, const bf16_t* __restrict__ mem,
    const uint8_t* __restrict__ col_valid, const uint8_t* __restrict__ row_valid, const float* __restrict__ prior, bf16_t* __restrict__ ctx,
    float* __restrict__ p_out, float* __restrict__ copy_out, const int64_t S) {
  extern __shared__ __attribute__((aligned(16))) float pa_smem[];
  float* sc = pa_smem;                              // [S] scores, then probabilities
  float* red = pa_smem + ((S + 3) & ~(int64_t)3);   // [PA_WAVES][PA_H] partial contexts; its head doubles as the reduction scratch
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t b = blockIdx.x;
  const uint8_t* cv = col_valid ? col_valid + b * S : nullptr;
  const bool row_ok = row_valid ? row_valid[b] != 0 : true;
  float ew[8], vv[8];
  {
    float w8[8];
    Vec16<float>::load(wq + b * PA_H + 8 * lane, *reinterpret_cast<float(*)[4]>(w8));
    Vec16<float>::load(wq + b * PA_H + 8 * lane + 4, *reinterpret_cast<float(*)[4]>(w8 + 4));
    if (wq_add) {  // the step-invariant part of the query projection (its feature columns and the bias), kept in f32
      float c8[8];
      Vec16<float>::load(wq_add + b * PA_H + 8 * lane, *reinterpret_cast<float(*)[4]>(c8));
      Vec16<float>::load(wq_add + b * PA_H + 8 * lane + 4, *reinterpret_cast<float(*)[4]>(c8 + 4));
#pragma unroll
      for (int e = 0; e < 8; ++e) w8[e] += c8[e];
    }
    Vec16<float>::load(v + 8 * lane, *reinterpret_cast<float(*)[4]>(vv));
    Vec16<float>::load(v + 8 * lane + 4, *reinterpret_cast<float(*)[4]>(vv + 4));
#pragma unroll
    for (int e = 0; e < 8; ++e) ew[e] = __expf(fminf(fmaxf(2.f * w8[e], -EXP_CLAMP), EXP_CLAMP));
  }
  float vs = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) vs += vv[e];
  vs = wave_sum(vs);
  // (1) s_j = sum_h v_h tanh(wq_h + uh_jh) = sum_h v_h - 2 sum_h v_h / (EW_h EU_jh + 1).  A wave takes four rows at a time (16 bytes of
  // each per lane) and sums them with ONE transposing reduction (row_sum4): six ds_bpermute steps per row were costing the stage
  // more than its reciprocals -- the first build spent 0.28 ms per launch in them, above its HBM time.  The next four rows are
  // requested before the current four are evaluated.
  const bf16_t* eub = eu + b * S * PA_H + 8 * lane;
  uint4 raw[4], nxt[4];
  bool ok[4], nok[4];
#define PA_LOAD(R, OK, J0)                                                     \
  _Pragma("unroll") for (int u = 0; u < 4; ++u) {                              \
    const int64_t j = (J0) + u * PA_WAVES;                                     \
    OK[u] = j < S && (!cv || cv[j]); /* wave-uniform */                        \
    if (OK[u]) R[u] = PA_STREAM_LOAD(eub + j * PA_H);                          \
  }
  PA_LOAD(raw, ok, (int64_t)wave)
  for (int64_t j0 = wave; j0 < S; j0 += 4 * PA_WAVES) {
    PA_LOAD(nxt, nok, j0 + 4 * PA_WAVES)
    float r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      r[u] = 0.f;
      if (ok[u]) {
        float x[8];
        Vec16<bf16_t>::unpack(raw[u], x);
#pragma unroll
        for (int e = 0; e < 8; ++e) r[u] += vv[e] * __builtin_amdgcn_rcpf(fmaf(ew[e], x[e], 1.f));
      }
    }
    const float tot = row_sum4(r[0], r[1], r[2], r[3]);  // lane row g holds the sum of r[g]
    const int g = lane >> 4;
    const int64_t j = j0 + g * PA_WAVES;
    const bool okg = g == 0 ? ok[0] : g == 1 ? ok[1] : g == 2 ? ok[2] : ok[3];
    if ((lane & 15) == 0 && j < S) sc[j] = okg ? vs - 2.f * tot : -INFINITY;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      raw[u] = nxt[u];
      ok[u] = nok[u];
    }
  }
#undef PA_LOAD
  __syncthreads();
  // (2) masked softmax over the S scores (exact zeros for a row without a valid key or an invalid target row)
  float mx = -INFINITY;
  for (int64_t j = threadIdx.x; j < S; j += 64 * PA_WAVES) mx = fmaxf(mx, sc[j]);
  mx = block_max(mx, red);
  float sum = 0.f;
  for (int64_t j = threadIdx.x; j < S; j += 64 * PA_WAVES) {
    const float e = mx == -INFINITY ? 0.f : __expf(sc[j] - mx);
    sc[j] = e;
    sum += e;
  }
  sum = block_sum(sum, red);
  const float inv = (row_ok && sum > 0.f) ? 1.f / sum : 0.f;
  float wsum = 0.f;
  for (int64_t j = threadIdx.x; j < S; j += 64 * PA_WAVES) {
    const float pj = sc[j] * inv;
    sc[j] = pj;
    p_out[b * S + j] = pj;
    if (prior) wsum += pj * prior[b * S + j];
  }
  if (prior) {
    wsum = block_sum(wsum, red);
    const float winv = 1.f / (1e-8f + wsum);
    for (int64_t j = threadIdx.x; j < S; j += 64 * PA_WAVES) copy_out[b * S + j] = sc[j] * prior[b * S + j] * winv;
  }
  __syncthreads();
  // (3) ctx = sum_j p_j value_j; rows with p_j == 0 (masked positions) are not read
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  const bf16_t* mb = mem + b * S * PA_H + 8 * lane;
  // as in sweep (1): the next four rows are requested before the current four are accumulated
  uint4 vraw[4], vnxt[4];
  float pj[4], pn[4];
#define PA_VLOAD(R, P, J0)                                                     \
  _Pragma("unroll") for (int u = 0; u < 4; ++u) {                              \
    const int64_t j = (J0) + u * PA_WAVES;                                     \
    P[u] = j < S ? sc[j] : 0.f; /* wave-uniform */                             \
    if (P[u] != 0.f) R[u] = PA_STREAM_LOAD(mb + j * PA_H);                     \
  }
  PA_VLOAD(vraw, pj, (int64_t)wave)
  for (int64_t j0 = wave; j0 < S; j0 += 4 * PA_WAVES) {
    PA_VLOAD(vnxt, pn, j0 + 4 * PA_WAVES)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (pj[u] != 0.f) {
        float x[8];
        Vec16<bf16_t>::unpack(vraw[u], x);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(pj[u], x[e], acc[e]);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      vraw[u] = vnxt[u];
      pj[u] = pn[u];
    }
  }
#undef PA_VLOAD
#pragma unroll
  for (int e = 0; e < 8; ++e) red[wave * PA_H + 8 * lane + e] = acc[e];
  __syncthreads();
  if (threadIdx.x < PA_H) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < PA_WAVES; ++w) t += red[w * PA_H + threadIdx.x];  // fixed order: bit-identical from launch to launch
    ctx[b * PA_H + threadIdx.x] = f32_to_bf16(t);
  }
}

// ---- K23 (round 5): the greedy step's pointer-generator head in one launch ----------------------------------------------------------
// gen = softmax(logits); pm = softmax(mix logits); dist = pm_0 gen + scatter(pm_k copy_k over the sorted source keys); id = argmax dist
// (CaSE/Model.py:34-48, :112-117, common/Utils.py:156-168 with the lowest index on ties).  One workgroup per answer row keeps the whole
// vocabulary row in LDS (V <= 36 000 floats): the five passes over the [B, V] matrix of the separate launches (softmax, p0 x gen,
// zero-fill + scatter, sum, argmax: ~0.2 ms per step at B 256 x V 30 522) become one read of the logits and one write of each output.
constexpr int PH_THREADS = 1024, PH_MAX_MEM = 4;
struct HeadArgs {
  const float* logits;      // [B, V]
  const float* mix_logits;  // [B, 1 + nmem]
  const uint32_t* keys;     // [B, S] sorted (token << 15 | position), 0xFFFFFFFF = no token
  const float* copy[PH_MAX_MEM];  // copy_k [B, len_k]: pointer weights of memory k (positions offset by the lengths before it)
  int64_t len[PH_MAX_MEM];
  float* gen;               // [B, V] or null
  float* dist;              // [B, V]
  int64_t* ids;             // [B]
  float* top;               // [B] or null: dist[id]
  int64_t V, S;
  int nmem;
};

__global__ __launch_bounds__(PH_THREADS) void pointer_head_decode_kernel(const HeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) float ph_smem[];
  float* row = ph_smem;                                  // [V]
  uint32_t* tk = reinterpret_cast<uint32_t*>(ph_smem + ((a.V + 3) & ~(int64_t)3));  // [PH_THREADS + 1]
  float* tv = reinterpret_cast<float*>(tk + PH_THREADS + 4);                       // [PH_THREADS]
  float* red = tv + PH_THREADS;                                                    // [64] reduction scratch
  __shared__ int64_t best_i[16];
  __shared__ float best_v[16];
  const int64_t b = blockIdx.x, V = a.V;
  const int tid = threadIdx.x;
  // the mixing probabilities (1 + nmem <= 5 values; every thread computes them)
  float pm[1 + PH_MAX_MEM];
  {
    float mm = -INFINITY, ss = 0.f;
#pragma unroll
    for (int k = 0; k <= PH_MAX_MEM; ++k) pm[k] = k <= a.nmem ? a.mix_logits[b * (a.nmem + 1) + k] : -INFINITY;
#pragma unroll
    for (int k = 0; k <= PH_MAX_MEM; ++k) mm = fmaxf(mm, pm[k]);
#pragma unroll
    for (int k = 0; k <= PH_MAX_MEM; ++k) {
      pm[k] = __expf(pm[k] - mm);  // exp(-inf) = 0 for the slots behind nmem
      ss += pm[k];
    }
#pragma unroll
    for (int k = 0; k <= PH_MAX_MEM; ++k) pm[k] /= ss;
  }
  const float* lg = a.logits + b * V;
  float mx = -INFINITY;
  for (int64_t i = tid; i < V; i += PH_THREADS) {
    const float x = lg[i];
    row[i] = x;
    mx = fmaxf(mx, x);
  }
  mx = block_max(mx, red);
  float sum = 0.f;
  for (int64_t i = tid; i < V; i += PH_THREADS) {
    const float e = __expf(row[i] - mx);
    row[i] = e;
    sum += e;
  }
  sum = block_sum(sum, red);
  const float inv = 1.f / sum;
  for (int64_t i = tid; i < V; i += PH_THREADS) {
    const float g = row[i] * inv;
    if (a.gen) a.gen[b * V + i] = g;
    row[i] = pm[0] * g;
  }
  __syncthreads();
  // pointer mass: the keys of a row are sorted by token, so a token's positions form runs; the last lane of a run inside a chunk sums it
  // (fixed order) and adds it to the row -- a token's run in a later chunk is added by a later iteration: no atomics, scheduling-free
  const uint32_t* k = a.keys + b * a.S;
  for (int64_t base = 0; base < a.S; base += PH_THREADS) {
    const int64_t i = base + tid;
    const uint32_t key = i < a.S ? k[i] : 0xFFFFFFFFu;
    const bool valid = key != 0xFFFFFFFFu;
    float w = 0.f;
    if (valid) {
      int64_t pos = key & 0x7FFFu;
      bool done = false;
#pragma unroll
      for (int m = 0; m < PH_MAX_MEM; ++m) {  // (compile-time m: pm[] stays in registers)
        if (m < a.nmem && !done) {
          if (pos < a.len[m]) {
            w = pm[m + 1] * a.copy[m][b * a.len[m] + pos];
            done = true;
          }
          pos -= a.len[m];
        }
      }
    }
    tk[tid] = valid ? (key >> 15) : 0xFFFFFFFFu;
    tv[tid] = w;
    if (tid == 0) tk[PH_THREADS] = 0xFFFFFFFEu;
    __syncthreads();
    const uint32_t tok = tk[tid];
    if (valid && tok != tk[tid + 1]) {
      int j = tid;
      while (j > 0 && tk[j - 1] == tok) --j;
      float s2 = 0.f;
      for (; j <= tid; ++j) s2 += tv[j];
      if (s2 != 0.f && tok < (uint32_t)V) row[tok] += s2;
    }
    __syncthreads();
  }
  // outputs: the distribution row and its argmax (lowest index on ties)
  float bv = -INFINITY;
  int64_t bi = V;
  for (int64_t i = tid; i < V; i += PH_THREADS) {
    const float x = row[i];
    if (a.dist) a.dist[b * V + i] = x;
    if (x > bv) {  // indices ascend within a thread: the first maximum is kept
      bv = x;
      bi = i;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int64_t oi = __shfl_xor(bi, o, 64);
    if (ov > bv || (ov == bv && oi < bi)) {
      bv = ov;
      bi = oi;
    }
  }
  if ((tid & 63) == 0) {
    best_v[tid >> 6] = bv;
    best_i[tid >> 6] = bi;
  }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < PH_THREADS / 64; ++w)
      if (best_v[w] > bv || (best_v[w] == bv && best_i[w] < bi)) {
        bv = best_v[w];
        bi = best_i[w];
      }
    a.ids[b] = bi < V ? bi : 0;
    if (a.top) a.top[b] = bv;
  }
}

// ---- K11 ---------------------------------------------------------------------------------------
__global__ void copy_scatter_fwd_kernel(const int64_t* __restrict__ src, const float* __restrict__ w,
                                        float* __restrict__ dist, int64_t B, int64_t Tn, int64_t S, int64_t V) {
  const int64_t n = B * Tn * S;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t sidx = i % S, bt = i / S, b = bt / Tn;
    const int64_t tok = src[b * S + sidx];
    const float val = w[i];
    if (tok >= 0 && tok < V && val != 0.f) atomicAdd(dist + bt * V + tok, val);
  }
}

__global__ void copy_scatter_bwd_kernel(const int64_t* __restrict__ src, const float* __restrict__ d_dist,
                                        float* __restrict__ d_w, int64_t B, int64_t Tn, int64_t S, int64_t V) {
  const int64_t n = B * Tn * S;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t sidx = i % S, bt = i / S, b = bt / Tn;
    const int64_t tok = src[b * S + sidx];
    d_w[i] = (tok >= 0 && tok < V) ? d_dist[bt * V + tok] : 0.f;
  }
}

// ---- K11, sorted form (the source map sorted on the device once per batch) -----------
// keys[b, i] = (token << 15 | position), ascending; out-of-vocabulary ids become 0xFFFFFFFF and sort behind every valid key.
// One workgroup per batch row, bitonic network over the next power of two in LDS (S <= 32768: 128 KiB).
__global__ __launch_bounds__(1024) void source_sort_kernel(const int64_t* __restrict__ src, uint32_t* __restrict__ keys, int64_t S,
                                                            int64_t V, int n) {
  extern __shared__ uint32_t sk[];
  const int64_t b = blockIdx.x;
  for (int i = threadIdx.x; i < n; i += 1024) {
    uint32_t k = 0xFFFFFFFFu;
    if (i < S) {
      const int64_t tok = src[b * S + i];
      if (tok >= 0 && tok < V) k = ((uint32_t)tok << 15) | (uint32_t)i;
    }
    sk[i] = k;
  }
  __syncthreads();
  for (int k = 2; k <= n; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < n; i += 1024) {
        const int p = i ^ j;
        if (p > i) {
          const uint32_t x = sk[i], y = sk[p];
          if ((x > y) == ((i & k) == 0)) {
            sk[i] = y;
            sk[p] = x;
          }
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < S; i += 1024) keys[b * S + i] = sk[i];
}

// dist[b, t, token] += sum of w[b, t, position] over each run of equal tokens in the sorted keys: one workgroup per (b, t) row
// walks the keys in chunks of 256; the last lane of a run sums it (fixed order) and adds it to the row with a plain
// read-modify-write -- a token's run in a later chunk is added by a later iteration of the same workgroup, so there are no
// atomics and the result does not depend on scheduling.
__global__ __launch_bounds__(256) void copy_scatter_sorted_kernel(const uint32_t* __restrict__ keys, const float* __restrict__ w,
                                                                  float* __restrict__ dist, int64_t Tn, int64_t S, int64_t V) {
  __shared__ uint32_t tk[257];
  __shared__ float tv[256];
  const int64_t bt = blockIdx.x, b = bt / Tn;
  const uint32_t* k = keys + b * S;
  const float* wr = w + bt * S;
  float* d = dist + bt * V;
  for (int64_t base = 0; base < S; base += 256) {
    const int64_t i = base + threadIdx.x;
    const uint32_t key = i < S ? k[i] : 0xFFFFFFFFu;
    const bool valid = key != 0xFFFFFFFFu;
    tk[threadIdx.x] = valid ? (key >> 15) : 0xFFFFFFFFu;
    tv[threadIdx.x] = valid ? wr[key & 0x7FFFu] : 0.f;
    if (threadIdx.x == 0) tk[256] = 0xFFFFFFFEu;  // differs from every token and from the invalid marker: lane 255 always ends a run
    __syncthreads();
    const uint32_t tok = tk[threadIdx.x];
    if (valid && tok != tk[threadIdx.x + 1]) {
      int j = threadIdx.x;
      while (j > 0 && tk[j - 1] == tok) --j;
      float sum = 0.f;
      for (; j <= (int)threadIdx.x; ++j) sum += tv[j];
      if (sum != 0.f) d[tok] += sum;
    }
    __threadfence_block();
    __syncthreads();
  }
}

// ---- K12 / K13 ---------------------------------------------------------------------------------
__global__ void nll_fwd_kernel(const float* __restrict__ dist, const int64_t* __restrict__ target,
                               float* __restrict__ per_row, int64_t rows, int64_t V) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
    const int64_t y = target[r];
    per_row[r] = (y > 0 && y < V) ? -logf(dist[r * V + y] + 1e-8f) : 0.f;
  }
}

__global__ void nll_bwd_kernel(const float* __restrict__ dist, const int64_t* __restrict__ target,
                               const float* __restrict__ g_row, float* __restrict__ d_dist, int64_t rows, int64_t V) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
    const int64_t y = target[r];
    if (y > 0 && y < V) d_dist[r * V + y] = -g_row[r] / (dist[r * V + y] + 1e-8f);
  }
}

__global__ __launch_bounds__(256) void row_argmax_kernel(const float* __restrict__ x, int64_t* __restrict__ idx,
                                                         float* __restrict__ val, int64_t rows, int64_t cols, int64_t ld) {
  __shared__ float sv[4];
  __shared__ int64_t si[4];
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
    float best = -INFINITY;
    int64_t bi = cols;  // sentinel larger than any index so the lowest index wins ties
    for (int64_t c = threadIdx.x; c < cols; c += 256) {
      const float vv = x[r * ld + c];
      if (vv > best || (vv == best && c < bi)) {
        best = vv;
        bi = c;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int64_t oi = __shfl_xor(bi, o, 64);
      if (ob > best || (ob == best && oi < bi)) {
        best = ob;
        bi = oi;
      }
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
      sv[threadIdx.x >> 6] = best;
      si[threadIdx.x >> 6] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < 4; ++w)
        if (sv[w] > best || (sv[w] == best && si[w] < bi)) {
          best = sv[w];
          bi = si[w];
        }
      idx[r] = bi < cols ? bi : 0;
      if (val) val[r] = best;
    }
  }
}

}  // namespace

extern "C" int case_additive_scores_fwd(const float* wq, const void* uh, const float* v, float* s, int64_t B, int64_t T,
                                        int64_t S, int64_t H, int32_t dtype, case_stream_t stream) {
  CASE_REQUIRE(wq && uh && v && s && B > 0 && T > 0 && S > 0 && H > 0 && B < 65536, "case_additive_scores_fwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int ev = dtype == CASE_F32 ? 4 : 8;
  if (T <= 2 && H % ev == 0 && H <= 2 * 64 * ev && (uintptr_t)uh % 16 == 0) {  // decode steps (T = 1)
    const dim3 g((unsigned)((S + 15) / 16 < 64 ? (S + 15) / 16 : 64), (unsigned)B);
    const bool one = H <= 64 * ev;
    // T = 1 (every greedy step) gets its own instantiation: the kernel is bound by the tanh issue rate, and the two-row form
    // evaluates the absent second row as well
#define ROWWISE(TY, FAST, TT, NCH) hipLaunchKernelGGL((additive_fwd_rowwise_kernel<TY, FAST, TT, NCH>), g, dim3(256), 0, st, wq, (const TY*)uh, v, s, T, S, H)
    if (dtype == CASE_F32) {
      if (T == 1) { if (one) ROWWISE(float, false, 1, 1); else ROWWISE(float, false, 1, 2); }
      else { if (one) ROWWISE(float, false, 2, 1); else ROWWISE(float, false, 2, 2); }
    } else {
      if (T == 1) { if (one) ROWWISE(bf16_t, true, 1, 1); else ROWWISE(bf16_t, true, 1, 2); }
      else { if (one) ROWWISE(bf16_t, true, 2, 1); else ROWWISE(bf16_t, true, 2, 2); }
    }
#undef ROWWISE
    return case_check_launch("case_additive_scores_fwd");
  }
  // a short memory (the 64-token query: one j block per item) leaves most CUs idle: spread the target chunks over workgroups too
  const int64_t jb = (S + AJ - 1) / AJ, tchunks = (T + 4 * ATW - 1) / (4 * ATW);
  const dim3 grid((unsigned)jb, (unsigned)B, (unsigned)(jb * B < 512 ? tchunks : 1));
  if (dtype == CASE_F32)
    hipLaunchKernelGGL((additive_fwd_kernel<float, false>), grid, dim3(256), 0, st, wq, (const float*)uh, v, s, T, S, H);
  else
    hipLaunchKernelGGL((additive_fwd_kernel<bf16_t, true>), grid, dim3(256), 0, st, wq, (const bf16_t*)uh, v, s, T, S, H);
  return case_check_launch("case_additive_scores_fwd");
}

extern "C" int case_additive_scores_bwd(const float* ds, const float* wq, const void* uh, const float* v, float* d_wq,
                                        float* d_uh, float* d_v, int64_t B, int64_t T, int64_t S, int64_t H, int32_t dtype,
                                        case_stream_t stream) {
  CASE_REQUIRE(ds && wq && uh && v && d_wq && d_uh && d_v && B > 0 && T > 0 && S > 0 && H > 0 && B < 65536,
               "case_additive_scores_bwd: bad argument");
  const unsigned hb = (unsigned)((H + 255) / 256);
  const int tblocks = (int)((T + CT - 1) / CT);
  // split the source positions of sweep 2 until ~2048 workgroups exist (whole CJC chunks per workgroup)
  const int64_t base_wgs = (int64_t)tblocks * hb * B, j_chunks = (S + CJC - 1) / CJC;
  int64_t sch = base_wgs >= 2048 ? 1 : (2048 + base_wgs - 1) / base_wgs;
  if (sch > j_chunks) sch = j_chunks;
  const int64_t j_per = ((j_chunks + sch - 1) / sch) * CJC;
  sch = (S + j_per - 1) / j_per;
  const dim3 g1((unsigned)((S + BJ - 1) / BJ), hb, (unsigned)B), g2((unsigned)(tblocks * sch), hb, (unsigned)B);
  hipStream_t st = (hipStream_t)stream;
  if (sch > 1 && hipMemsetAsync(d_wq, 0, (size_t)(B * T * H) * sizeof(float), st) != hipSuccess)
    return case_set_error(CASE_E_LAUNCH, "case_additive_scores_bwd: hipMemsetAsync failed");
  if (dtype == CASE_F32) {
    hipLaunchKernelGGL((additive_bwd_uh_kernel<float, false>), g1, dim3(256), 0, st, ds, wq, (const float*)uh, v, d_uh, d_v, T, S, H);
    hipLaunchKernelGGL((additive_bwd_wq_kernel<float, false>), g2, dim3(256), 0, st, ds, wq, (const float*)uh, v, d_wq, T, S, H, tblocks, j_per);
  } else {
    hipLaunchKernelGGL((additive_bwd_uh_kernel<bf16_t, true>), g1, dim3(256), 0, st, ds, wq, (const bf16_t*)uh, v, d_uh, d_v, T, S, H);
    hipLaunchKernelGGL((additive_bwd_wq_kernel<bf16_t, true>), g2, dim3(256), 0, st, ds, wq, (const bf16_t*)uh, v, d_wq, T, S, H, tblocks, j_per);
  }
  return case_check_launch("case_additive_scores_bwd");
}

extern "C" int case_additive_key_exp(const float* uh, void* eu, int64_t n, case_stream_t stream) {
  CASE_REQUIRE(uh && eu && n > 0 && n % 8 == 0 && (uintptr_t)uh % 16 == 0 && (uintptr_t)eu % 16 == 0, "case_additive_key_exp: bad argument");
  hipLaunchKernelGGL(additive_key_exp_kernel, dim3(grid_for(n / 8, 256, 1, 256 * 32)), dim3(256), 0, (hipStream_t)stream, uh,
                     reinterpret_cast<bf16_t*>(eu), n / 8);
  return case_check_launch("case_additive_key_exp");
}

extern "C" int case_pointer_attend_decode(const float* wq, const float* wq_add, const void* eu, const float* v, const void* value, const uint8_t* col_valid,
                                          const uint8_t* row_valid, const float* prior, void* ctx, float* p, float* copy, int64_t B, int64_t S,
                                          int64_t H, case_stream_t stream) {
  CASE_REQUIRE(wq && eu && v && value && ctx && p && B > 0 && S > 0 && B < (1ll << 31) && (copy != nullptr) == (prior != nullptr),
               "case_pointer_attend_decode: bad argument");
  if (H != PA_H || S > 28000)
    return case_set_error(CASE_E_UNSUPPORTED, "case_pointer_attend_decode: built for H = %d and S <= 28000 (run case_additive_scores_fwd + "
                                              "case_softmax_fwd + case_gemm)", PA_H);
  for (const void* q : {(const void*)wq, (const void*)wq_add, eu, (const void*)v, value})
    CASE_REQUIRE((reinterpret_cast<uintptr_t>(q) & 15) == 0, "case_pointer_attend_decode: tensors must be 16-byte aligned");
  const size_t lds = (size_t)(((S + 3) & ~(int64_t)3) + PA_WAVES * PA_H) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pointer_attend_decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
        hipSuccess)
      return case_set_error(CASE_E_LAUNCH, "case_pointer_attend_decode: cannot reserve LDS");
    attr = true;
  }
  hipLaunchKernelGGL(pointer_attend_decode_kernel, dim3((unsigned)B), dim3(64 * PA_WAVES), lds, (hipStream_t)stream, wq, wq_add,
                     reinterpret_cast<const bf16_t*>(eu), v, reinterpret_cast<const bf16_t*>(value), col_valid, row_valid, prior,
                     reinterpret_cast<bf16_t*>(ctx), p, copy, S);
  return case_check_launch("case_pointer_attend_decode");
}

extern "C" int case_pointer_head_decode(const float* logits, const float* mix_logits, const uint32_t* keys, const float* const* copies,
                                        const int64_t* lens, int32_t nmem, float* gen, float* dist, int64_t* ids, float* top, int64_t B, int64_t V,
                                        int64_t S, case_stream_t stream) {
  CASE_REQUIRE(logits && mix_logits && keys && copies && lens && ids && B > 0 && V > 0 && S > 0 && nmem >= 1 && B < (1ll << 31),
               "case_pointer_head_decode: bad argument");
  if (nmem > PH_MAX_MEM || V > 36000 || S > 32768)
    return case_set_error(CASE_E_UNSUPPORTED, "case_pointer_head_decode: built for <= %d memories, V <= 36000, S <= 32768 (run the softmax / "
                                              "scatter / argmax launches)", PH_MAX_MEM);
  HeadArgs a;
  a.logits = logits;
  a.mix_logits = mix_logits;
  a.keys = keys;
  int64_t total = 0;
  for (int m = 0; m < PH_MAX_MEM; ++m) {
    a.copy[m] = m < nmem ? copies[m] : nullptr;
    a.len[m] = m < nmem ? lens[m] : 0;
    total += a.len[m];
    CASE_REQUIRE(m >= nmem || (copies[m] && lens[m] > 0), "case_pointer_head_decode: null copy weights");
  }
  CASE_REQUIRE(total == S, "case_pointer_head_decode: the memories hold %lld positions, the source map %lld", (long long)total, (long long)S);
  a.gen = gen;
  a.dist = dist;
  a.ids = ids;
  a.top = top;
  a.V = V;
  a.S = S;
  a.nmem = nmem;
  const size_t lds = (size_t)(((V + 3) & ~(int64_t)3) + (PH_THREADS + 4) + PH_THREADS + 64) * 4;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pointer_head_decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024) !=
        hipSuccess)
      return case_set_error(CASE_E_LAUNCH, "case_pointer_head_decode: cannot reserve LDS");
    attr = true;
  }
  hipLaunchKernelGGL(pointer_head_decode_kernel, dim3((unsigned)B), dim3(PH_THREADS), lds, (hipStream_t)stream, a);
  return case_check_launch("case_pointer_head_decode");
}

extern "C" int case_copy_scatter_fwd(const int64_t* src, const float* w, float* dist, int64_t B, int64_t T, int64_t S,
                                     int64_t V, case_stream_t stream) {
  CASE_REQUIRE(src && w && dist && B > 0 && T > 0 && S > 0 && V > 0, "case_copy_scatter_fwd: bad argument");
  hipLaunchKernelGGL(copy_scatter_fwd_kernel, dim3(grid_for(B * T * S, 256, 2)), dim3(256), 0, (hipStream_t)stream, src, w,
                     dist, B, T, S, V);
  return case_check_launch("case_copy_scatter_fwd");
}

extern "C" int case_copy_scatter_bwd(const int64_t* src, const float* d_dist, float* d_w, int64_t B, int64_t T, int64_t S,
                                     int64_t V, case_stream_t stream) {
  CASE_REQUIRE(src && d_dist && d_w && B > 0 && T > 0 && S > 0 && V > 0, "case_copy_scatter_bwd: bad argument");
  hipLaunchKernelGGL(copy_scatter_bwd_kernel, dim3(grid_for(B * T * S, 256, 2)), dim3(256), 0, (hipStream_t)stream, src,
                     d_dist, d_w, B, T, S, V);
  return case_check_launch("case_copy_scatter_bwd");
}

extern "C" int case_source_sort(const int64_t* src, uint32_t* keys, int64_t B, int64_t S, int64_t V, case_stream_t stream) {
  CASE_REQUIRE(src && keys && B > 0 && S > 0 && V > 0, "case_source_sort: bad argument");
  CASE_REQUIRE(S <= 32768 && V <= 131071, "case_source_sort: S <= 32768 and V <= 131071 (token << 15 | position must fit 32 bits)");
  int n = 2;
  while (n < S) n <<= 1;
  // opt-in to > 64 KiB of dynamic LDS: idempotent host call, once per batch (no library state kept)
  if (n > 16384 && hipFuncSetAttribute((const void*)source_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 32768 * 4) != hipSuccess)
    return case_set_error(CASE_E_LAUNCH, "case_source_sort: cannot reserve 128 KiB of LDS");
  hipLaunchKernelGGL(source_sort_kernel, dim3((unsigned)B), dim3(1024), (size_t)n * 4, (hipStream_t)stream, src, keys, S, V, n);
  return case_check_launch("case_source_sort");
}

extern "C" int case_copy_scatter_sorted_fwd(const uint32_t* keys, const float* w, float* dist, int64_t B, int64_t T, int64_t S,
                                            int64_t V, case_stream_t stream) {
  CASE_REQUIRE(keys && w && dist && B > 0 && T > 0 && S > 0 && V > 0 && B * T < (1ll << 31), "case_copy_scatter_sorted_fwd: bad argument");
  CASE_REQUIRE(S <= 32768 && V <= 131071, "case_copy_scatter_sorted_fwd: S <= 32768 and V <= 131071");
  hipLaunchKernelGGL(copy_scatter_sorted_kernel, dim3((unsigned)(B * T)), dim3(256), 0, (hipStream_t)stream, keys, w, dist, T, S, V);
  return case_check_launch("case_copy_scatter_sorted_fwd");
}

extern "C" int case_nll_gather_fwd(const float* dist, const int64_t* target, float* per_row, int64_t rows, int64_t V,
                                   case_stream_t stream) {
  CASE_REQUIRE(dist && target && per_row && rows > 0 && V > 0, "case_nll_gather_fwd: bad argument");
  hipLaunchKernelGGL(nll_fwd_kernel, dim3(grid_for(rows, 256)), dim3(256), 0, (hipStream_t)stream, dist, target, per_row, rows, V);
  return case_check_launch("case_nll_gather_fwd");
}

extern "C" int case_nll_gather_bwd(const float* dist, const int64_t* target, const float* g_row, float* d_dist,
                                   int64_t rows, int64_t V, case_stream_t stream) {
  CASE_REQUIRE(dist && target && g_row && d_dist && rows > 0 && V > 0, "case_nll_gather_bwd: bad argument");
  hipLaunchKernelGGL(nll_bwd_kernel, dim3(grid_for(rows, 256)), dim3(256), 0, (hipStream_t)stream, dist, target, g_row, d_dist, rows, V);
  return case_check_launch("case_nll_gather_bwd");
}

namespace {
// ---- greedy post-processing (common/Utils.py:200-217 to_sentence): per answer row drop BOS / PAD, stop at the first EOS ----
// one thread per row (T <= a few hundred); out [B, T] receives the kept ids front-packed (pad behind), len [B] their count
__global__ void sentence_compact_kernel(const int64_t* __restrict__ ids, int64_t* __restrict__ out, int32_t* __restrict__ len,
                                        int64_t B, int64_t T, int64_t bos, int64_t pad, int64_t eos) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int n = 0;
  bool open = true;
  for (int64_t t = 0; t < T; ++t) {
    const int64_t id = ids[b * T + t];
    if (open && id == eos) open = false;
    if (open && id != bos && id != pad) out[b * T + n++] = id;
  }
  len[b] = n;
  for (int64_t t = n; t < T; ++t) out[b * T + t] = pad;
}
}  // namespace

extern "C" int case_sentence_compact(const int64_t* ids, int64_t* out, int32_t* len, int64_t B, int64_t T, int64_t bos, int64_t pad,
                                     int64_t eos, case_stream_t stream) {
  CASE_REQUIRE(ids && out && len && B > 0 && T > 0, "case_sentence_compact: bad argument");
  hipLaunchKernelGGL(sentence_compact_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, (hipStream_t)stream, ids, out, len, B, T, bos,
                     pad, eos);
  return case_check_launch("case_sentence_compact");
}

extern "C" int case_row_argmax(const float* x, int64_t* idx, float* val, int64_t rows, int64_t cols, int64_t ld,
                               case_stream_t stream) {
  CASE_REQUIRE(x && idx && rows > 0 && cols > 0 && ld >= cols, "case_row_argmax: bad argument");
  hipLaunchKernelGGL(row_argmax_kernel, dim3(grid_for(rows, 1)), dim3(256), 0, (hipStream_t)stream, x, idx, val, rows, cols, ld);
  return case_check_launch("case_row_argmax");
}
