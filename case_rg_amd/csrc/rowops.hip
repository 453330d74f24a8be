// K2 LayerNorm and the masked row softmax (softmax stage of K4/K5/K6 and of K7/K8/K10/K11).
// HBM-bound row kernels: one wave (C <= 1024) or one 256-thread workgroup per row, grid-stride over rows,
// wavefront shuffles for the reductions, f32 statistics regardless of the storage dtype.
#include <stdlib.h>

// The row kernels of this file read every tensor once (the LayerNorm input and its incoming gradient are not touched again before the backward pass
// / at all), so their 16-byte loads are non-temporal: the rows they WRITE -- the next GEMM's operand -- keep the Infinity Cache.  Training step
// 95.14 / 94.95 -> 94.89 / 94.68 ms (alternating pairs, one box).  -DCASE_STREAM_DEFAULT_POLICY: the default policy (A/B builds).
#ifndef CASE_STREAM_DEFAULT_POLICY
#define CASE_VEC16_NT
#endif
#include "common.h"

namespace {

constexpr int LN_THREADS = 256;
constexpr int LN_MAXPT = 16;  // columns per thread cached in registers by the backward (cols <= 4096)

template <typename T>
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_kernel(const T* __restrict__ x, const T* __restrict__ x2,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            int64_t rows, int64_t cols, float eps) {
  __shared__ float red[32];
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
    const T* xr = x + r * cols;
    const T* x2r = x2 ? x2 + r * cols : nullptr;
    float s = 0.f;
    for (int64_t c = threadIdx.x; c < cols; c += LN_THREADS) s += Elem<T>::ld(xr + c) + (x2r ? Elem<T>::ld(x2r + c) : 0.f);
    const float mu = block_sum(s, red) / (float)cols;
    float q = 0.f;
    for (int64_t c = threadIdx.x; c < cols; c += LN_THREADS) {
      const float d = Elem<T>::ld(xr + c) + (x2r ? Elem<T>::ld(x2r + c) : 0.f) - mu;
      q += d * d;
    }
    const float rs = rsqrtf(block_sum(q, red) / (float)cols + eps);
    if (threadIdx.x == 0) {
      mean[r] = mu;
      rstd[r] = rs;
    }
    T* yr = y + r * cols;
    for (int64_t c = threadIdx.x; c < cols; c += LN_THREADS) {
      const float v = Elem<T>::ld(xr + c) + (x2r ? Elem<T>::ld(x2r + c) : 0.f);
      Elem<T>::st(yr + c, (v - mu) * rs * gamma[c] + beta[c]);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(LN_THREADS) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const T* __restrict__ x2, const float* __restrict__ gamma,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, T* __restrict__ dx,
                                                            const T* __restrict__ dx_add, float* __restrict__ d_gamma,
                                                            float* __restrict__ d_beta, int64_t rows, int64_t cols) {
  __shared__ float red[32];
  float acc_g[LN_MAXPT], acc_b[LN_MAXPT];
#pragma unroll
  for (int i = 0; i < LN_MAXPT; ++i) acc_g[i] = acc_b[i] = 0.f;
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
    const T* xr = x + r * cols;
    const T* x2r = x2 ? x2 + r * cols : nullptr;
    const T* dyr = dy + r * cols;
    const float mu = mean[r], rs = rstd[r];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXPT; ++i) {
      const int64_t c = threadIdx.x + (int64_t)i * LN_THREADS;
      if (c < cols) {
        const float xh = (Elem<T>::ld(xr + c) + (x2r ? Elem<T>::ld(x2r + c) : 0.f) - mu) * rs;
        const float d = Elem<T>::ld(dyr + c);
        const float g = d * gamma[c];
        s1 += g;
        s2 += g * xh;
        acc_g[i] += d * xh;
        acc_b[i] += d;
      }
    }
    const float m1 = block_sum(s1, red) / (float)cols;
    const float m2 = block_sum(s2, red) / (float)cols;
    T* dxr = dx + r * cols;
#pragma unroll
    for (int i = 0; i < LN_MAXPT; ++i) {
      const int64_t c = threadIdx.x + (int64_t)i * LN_THREADS;
      if (c < cols) {
        const float xh = (Elem<T>::ld(xr + c) + (x2r ? Elem<T>::ld(x2r + c) : 0.f) - mu) * rs;
        const float g = Elem<T>::ld(dyr + c) * gamma[c];
        Elem<T>::st(dxr + c, rs * (g - m1 - xh * m2) + (dx_add ? Elem<T>::ld(dx_add + r * cols + c) : 0.f));
      }
    }
  }
#pragma unroll
  for (int i = 0; i < LN_MAXPT; ++i) {
    const int64_t c = threadIdx.x + (int64_t)i * LN_THREADS;
    if (c < cols) {
      atomicAdd(d_gamma + c, acc_g[i]);
      atomicAdd(d_beta + c, acc_b[i]);
    }
  }
}

// ---- vectorised LayerNorm: one wave per row, the row lives in registers (16-byte loads, <= 8 vectors per lane) ------
constexpr int LNV_MAX = 8;

template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_fwd_vec_kernel(const T* __restrict__ x, const T* __restrict__ x2,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         T* __restrict__ y, float* __restrict__ mean,
                                                         float* __restrict__ rstd, int64_t rows, int64_t cols, float eps) {
  constexpr int E = Vec16<T>::N;
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  for (int64_t r = wave; r < rows; r += nwaves) {
    float v[NV][E];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int64_t c = ((int64_t)i * 64 + lane) * E;
      if (c < cols) {
        Vec16<T>::load(x + r * cols + c, v[i]);
        if (x2) {
          float w[E];
          Vec16<T>::load(x2 + r * cols + c, w);
#pragma unroll
          for (int e = 0; e < E; ++e) v[i][e] += w[e];
        }
#pragma unroll
        for (int e = 0; e < E; ++e) s += v[i][e];
      }
    }
    const float mu = wave_sum(s) / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (((int64_t)i * 64 + lane) * E < cols) {
#pragma unroll
        for (int e = 0; e < E; ++e) q += (v[i][e] - mu) * (v[i][e] - mu);
      }
    const float rs = rsqrtf(wave_sum(q) / (float)cols + eps);
    if (lane == 0) {
      mean[r] = mu;
      rstd[r] = rs;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int64_t c = ((int64_t)i * 64 + lane) * E;
      if (c < cols) {
        float g[E], b[E], o[E];
#pragma unroll
        for (int e = 0; e < E; e += 4) {
          const float4 gg = *reinterpret_cast<const float4*>(gamma + c + e), bb = *reinterpret_cast<const float4*>(beta + c + e);
          g[e] = gg.x; g[e + 1] = gg.y; g[e + 2] = gg.z; g[e + 3] = gg.w;
          b[e] = bb.x; b[e + 1] = bb.y; b[e + 2] = bb.z; b[e + 3] = bb.w;
        }
#pragma unroll
        for (int e = 0; e < E; ++e) o[e] = (v[i][e] - mu) * rs * g[e] + b[e];
        Vec16<T>::store(y + r * cols + c, o);
      }
    }
  }
}

// Backward for row widths that are not whole 64-lane chunks (768, 3840 ... of cfg 5): the guarded form (one row in flight per
// wave, per-lane column partials in registers).  Slower than the two kernels below; kept for generality.
template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_bwd_vec_guarded_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                         const T* __restrict__ x2, const float* __restrict__ gamma,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         T* __restrict__ dx, const T* __restrict__ dx_add,
                                                         float* __restrict__ d_gamma, float* __restrict__ d_beta, int64_t rows,
                                                         int64_t cols) {
  constexpr int E = Vec16<T>::N;
  extern __shared__ float part[];  // [2][cols] partial d_gamma / d_beta of waves 1..3, added to wave 0's
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t wave = (int64_t)blockIdx.x * 4 + wid, nwaves = (int64_t)gridDim.x * 4;
  float ag[NV][E], ab[NV][E], g[NV][E];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int64_t c = ((int64_t)i * 64 + lane) * E;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      ag[i][e] = ab[i][e] = 0.f;
      g[i][e] = c < cols ? gamma[c + e] : 0.f;
    }
  }
  for (int64_t r = wave; r < rows; r += nwaves) {
    const float mu = mean[r], rs = rstd[r];
    float xh[NV][E], gy[NV][E];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int64_t c = ((int64_t)i * 64 + lane) * E;
      if (c < cols) {
        float d[E];
        Vec16<T>::load(x + r * cols + c, xh[i]);
        if (x2) {
          float w[E];
          Vec16<T>::load(x2 + r * cols + c, w);
#pragma unroll
          for (int e = 0; e < E; ++e) xh[i][e] += w[e];
        }
        Vec16<T>::load(dy + r * cols + c, d);
#pragma unroll
        for (int e = 0; e < E; ++e) {
          xh[i][e] = (xh[i][e] - mu) * rs;
          gy[i][e] = d[e] * g[i][e];
          s1 += gy[i][e];
          s2 += gy[i][e] * xh[i][e];
          ag[i][e] += d[e] * xh[i][e];
          ab[i][e] += d[e];
        }
      }
    }
    const float m1 = wave_sum(s1) / (float)cols, m2 = wave_sum(s2) / (float)cols;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int64_t c = ((int64_t)i * 64 + lane) * E;
      if (c < cols) {
        float o[E];
#pragma unroll
        for (int e = 0; e < E; ++e) o[e] = rs * (gy[i][e] - m1 - xh[i][e] * m2);
        if (dx_add) {  // a second gradient of the same tensor (its residual use), summed here instead of by a separate pass
          float w[E];
          Vec16<T>::load(dx_add + r * cols + c, w);
#pragma unroll
          for (int e = 0; e < E; ++e) o[e] += w[e];
        }
        Vec16<T>::store(dx + r * cols + c, o);
      }
    }
  }
  // workgroup reduction of the column partials through LDS, then one atomic per column per workgroup
  for (int w = 1; w < 4; ++w) {
    if (wid == w) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int64_t c = ((int64_t)i * 64 + lane) * E;
        if (c < cols) {
#pragma unroll
          for (int e = 0; e < E; ++e) {
            part[c + e] = ag[i][e];
            part[cols + c + e] = ab[i][e];
          }
        }
      }
    }
    __syncthreads();
    if (wid == 0) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int64_t c = ((int64_t)i * 64 + lane) * E;
        if (c < cols) {
#pragma unroll
          for (int e = 0; e < E; ++e) {
            ag[i][e] += part[c + e];
            ab[i][e] += part[cols + c + e];
          }
        }
      }
    }
    __syncthreads();
  }
  if (wid == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int64_t c = ((int64_t)i * 64 + lane) * E;
      if (c < cols) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
          atomicAdd(d_gamma + c + e, ag[i][e]);
          atomicAdd(d_beta + c + e, ab[i][e]);
        }
      }
    }
  }
}

// Optional SECOND output of the vector backward kernels (case_layernorm_bwd_dropout): the LayerNorm's input was y = dropout(x W^T + b) + r,
// so the Linear's backward needs mask * dx / (1 - p) next to dx itself (which is the gradient of r).  Written here, from the registers that
// hold dx, it costs one store; as the separate case_dropout pass it replaces it cost a read and a store of the whole tensor.  The mask is
// case_dropout's: element index = row * cols + column behind (seed, offset), applied to dx AFTER its rounding to T (what the separate
// pass read), so both forms produce the same bits.
struct LnDrop {
  void* out;
  float p;
  uint64_t seed, offset;
  const CaseStepState* state;  // nullable: offset += state->rng_base (ABI 600)
};
template <typename T, typename V = Vec16<T>>
__device__ __forceinline__ void ln_store_dropped(const LnDrop& dr, int64_t idx, const float (&o)[V::N]) {
  constexpr int E = V::N;
  float q[E];
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int i = 0; i < E; i += 2) {  // the value the separate pass would have read back: dx rounded to bf16
      const uint32_t w = f32x2_to_bf16x2(o[i], o[i + 1]);
      q[i] = __uint_as_float(w << 16);
      q[i + 1] = __uint_as_float(w & 0xffff0000u);
    }
  } else {
#pragma unroll
    for (int e = 0; e < E; ++e) q[e] = o[e];
  }
  const float scale = 1.f / (1.f - dr.p);
#pragma unroll
  for (int e = 0; e < E; e += 2) {  // idx is a multiple of E: even
    float u0, u1;
    rng_uniform2(dr.seed, dr.offset + rng_base_of(dr.state) + (uint64_t)(idx + e), u0, u1);
    q[e] = u0 >= dr.p ? q[e] * scale : 0.f;
    q[e + 1] = u1 >= dr.p ? q[e + 1] * scale : 0.f;
  }
  V::store(reinterpret_cast<T*>(dr.out) + idx, q);
}

// Backward.  One wave per row, the row in registers; the raw 16-byte vectors of the wave's NEXT row (x, dy and the optional
// x2 / dx_add streams) are requested before the current row's reductions, so every wave keeps two rows of loads in flight:
// the row is a dependent chain (load -> two wave reductions -> store) and with one row in flight per wave the kernel ran at
// 1.4-1.7 TB/s on streams that do not fit the Infinity Cache (copy ceiling 6.3 TB/s).  The loop body is branch-free (cols ==
// NV * 64 * E, optional streams are template flags): behind a conditional load hipcc waits vmcnt(0) and drains the prefetch.
template <typename T, int NV, bool HAS_X2, bool HAS_ADD, bool DROP2 = false, typename V = Vec16<T>>
__global__ __launch_bounds__(256) void ln_bwd_vec_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                         const T* __restrict__ x2, const float* __restrict__ gamma,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         T* __restrict__ dx, const T* __restrict__ dx_add,
                                                         float* __restrict__ d_gamma, float* __restrict__ d_beta, int64_t rows,
                                                         int64_t cols, const LnDrop dr = LnDrop()) {
  constexpr int E = V::N;
  typedef typename V::raw raw_t;
  extern __shared__ float part[];  // [4 waves][2][cols] column partials, summed and added to d_gamma / d_beta at the end
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t wave = (int64_t)blockIdx.x * 4 + wid, nwaves = (int64_t)gridDim.x * 4;
  float ag[NV][E], ab[NV][E], g[NV][E];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int64_t c = ((int64_t)i * 64 + lane) * E;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      ag[i][e] = ab[i][e] = 0.f;
      g[i][e] = gamma[c + e];
    }
  }
  struct Raw {  // a row stays packed (16 bytes per stream and chunk) between the two passes
    raw_t x[NV], d[NV], x2[HAS_X2 ? NV : 1], add[HAS_ADD ? NV : 1];
    float mu, rs;
  };
  auto request = [&](Raw& w, int64_t r) {
    const int64_t base = r * cols + (int64_t)lane * E;
    w.mu = mean[r];
    w.rs = rstd[r];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int64_t o = base + (int64_t)i * 64 * E;
      w.x[i] = *reinterpret_cast<const raw_t*>(x + o);
      w.d[i] = *reinterpret_cast<const raw_t*>(dy + o);
      if constexpr (HAS_X2) w.x2[i] = *reinterpret_cast<const raw_t*>(x2 + o);
      if constexpr (HAS_ADD) w.add[i] = *reinterpret_cast<const raw_t*>(dx_add + o);
    }
  };
  auto normalised = [&](const Raw& w, int i, float (&xh)[E]) {
    V::unpack(w.x[i], xh);
    if constexpr (HAS_X2) {
      float t[E];
      V::unpack(w.x2[i], t);
#pragma unroll
      for (int e = 0; e < E; ++e) xh[e] += t[e];
    }
#pragma unroll
    for (int e = 0; e < E; ++e) xh[e] = (xh[e] - w.mu) * w.rs;
  };
  Raw cur, nxt;
  request(cur, wave < rows ? wave : rows - 1);
  for (int64_t r = wave; r < rows; r += nwaves) {
    request(nxt, r + nwaves < rows ? r + nwaves : rows - 1);  // unconditional: the last one re-reads a row and is dropped
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float xh[E], d[E];
      normalised(cur, i, xh);
      V::unpack(cur.d[i], d);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const float gy = d[e] * g[i][e];
        s1 += gy;
        s2 = fmaf(gy, xh[e], s2);  // explicit FMAs: the DROP2 and plain instantiations must round alike (-ffp-contract=fast may not)
        ag[i][e] = fmaf(d[e], xh[e], ag[i][e]);
        ab[i][e] += d[e];
      }
    }
    const float m1 = wave_sum(s1) / (float)cols, m2 = wave_sum(s2) / (float)cols;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float xh[E], d[E], o[E];
      normalised(cur, i, xh);
      V::unpack(cur.d[i], d);
#pragma unroll
      for (int e = 0; e < E; ++e) o[e] = cur.rs * fmaf(-xh[e], m2, fmaf(d[e], g[i][e], -m1));
      if constexpr (HAS_ADD) {  // a second gradient of the same tensor (its residual use), summed here instead of by a separate pass
        float w[E];
        V::unpack(cur.add[i], w);
#pragma unroll
        for (int e = 0; e < E; ++e) o[e] += w[e];
      }
      V::store(dx + r * cols + ((int64_t)i * 64 + lane) * E, o);
      if constexpr (DROP2) ln_store_dropped<T, V>(dr, r * cols + ((int64_t)i * 64 + lane) * E, o);
    }
    cur = nxt;
  }
  // column partials: every wave parks its sums in LDS, then thread t adds columns t, t + 256, ... -- 256 contiguous bytes per
  // atomic wave-instruction (a lane adding its own 8 columns is a 32-byte-strided scatter: one line per lane)
  float* mine = part + (int64_t)wid * 2 * cols;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int64_t c = ((int64_t)i * 64 + lane) * E;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      mine[c + e] = ag[i][e];
      mine[cols + c + e] = ab[i][e];
    }
  }
  __syncthreads();
  for (int64_t c = threadIdx.x; c < 2 * cols; c += 256) {
    const float t = part[c] + part[2 * cols + c] + part[4 * cols + c] + part[6 * cols + c];
    atomicAdd(c < cols ? d_gamma + c : d_beta + (c - cols), t);
  }
}

// Backward for WIDE rows (5H = 2560 columns: the first TransformerBlock of every stack).  With a whole row per wave the
// per-lane gamma / beta partial sums alone take 2 x 40 registers next to two passes' worth of row data: 250+ registers, one
// wave per SIMD, 1.7 TB/s.  Here a workgroup of NW waves shares each row: wave w owns columns [512 w, 512 w + 512) (8 per
// lane), the two row statistics are combined through LDS (R rows per barrier, slots double-buffered -> one barrier per
// group), and the loads of the next group are requested before the barrier.  ~70 registers, 4+ workgroups per CU.
template <typename T, int R, bool HAS_X2, bool HAS_ADD, bool DROP2 = false, typename V = Vec16<T>, int MAXW = 8>
__global__ __launch_bounds__(64 * MAXW) void ln_bwd_split_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           const T* __restrict__ x2, const float* __restrict__ gamma,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           T* __restrict__ dx, const T* __restrict__ dx_add,
                                                           float* __restrict__ d_gamma, float* __restrict__ d_beta, int64_t rows,
                                                           int64_t cols, const LnDrop dr = LnDrop()) {
  constexpr int E = V::N;
  typedef typename V::raw raw_t;
  __shared__ float part[2][R][MAXW][2];  // [buffer][row of the group][wave][s1, s2]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int64_t col = ((int64_t)wid * 64 + lane) * E;  // cols == nw * 64 * E (launcher)
  float ag[E], ab[E], g[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    ag[e] = ab[e] = 0.f;
    g[e] = gamma[col + e];
  }
  struct Raw {  // the rows stay packed (16 bytes per stream) between the two passes; x^ and dy gamma are recomputed
    raw_t x[R], d[R], x2[HAS_X2 ? R : 1], add[HAS_ADD ? R : 1];
    float mu[R], rs[R];
  };
  auto request = [&](Raw& w, int64_t row0) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int64_t r = row0 + j < rows ? row0 + j : rows - 1;  // clamp: the tail group recomputes the last row, stores are guarded
      const int64_t o = r * cols + col;
      w.mu[j] = mean[r];
      w.rs[j] = rstd[r];
      w.x[j] = *reinterpret_cast<const raw_t*>(x + o);
      w.d[j] = *reinterpret_cast<const raw_t*>(dy + o);
      if constexpr (HAS_X2) w.x2[j] = *reinterpret_cast<const raw_t*>(x2 + o);
      if constexpr (HAS_ADD) w.add[j] = *reinterpret_cast<const raw_t*>(dx_add + o);
    }
  };
  auto normalised = [&](const Raw& w, int j, float (&xh)[E]) {
    V::unpack(w.x[j], xh);
    if constexpr (HAS_X2) {
      float t[E];
      V::unpack(w.x2[j], t);
#pragma unroll
      for (int e = 0; e < E; ++e) xh[e] += t[e];
    }
#pragma unroll
    for (int e = 0; e < E; ++e) xh[e] = (xh[e] - w.mu[j]) * w.rs[j];
  };
  const int64_t stride = (int64_t)gridDim.x * R;
  Raw cur, nxt;
  int buf = 0;
  if ((int64_t)blockIdx.x * R < rows) request(cur, (int64_t)blockIdx.x * R);
  for (int64_t row0 = (int64_t)blockIdx.x * R; row0 < rows; row0 += stride, buf ^= 1) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
      float xh[E], d[E];
      normalised(cur, j, xh);
      V::unpack(cur.d[j], d);
      float s1 = 0.f, s2 = 0.f;
      const bool real = row0 + j < rows;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const float gy = d[e] * g[e];
        s1 += gy;
        s2 = fmaf(gy, xh[e], s2);
        if (real) {
          ag[e] = fmaf(d[e], xh[e], ag[e]);
          ab[e] += d[e];
        }
      }
      s1 = wave_sum(s1);
      s2 = wave_sum(s2);
      if (lane == 0) {
        part[buf][j][wid][0] = s1;
        part[buf][j][wid][1] = s2;
      }
    }
    if (row0 + stride < rows) request(nxt, row0 + stride);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < R; ++j) {
      float m1 = 0.f, m2 = 0.f;
      for (int w = 0; w < nw; ++w) {
        m1 += part[buf][j][w][0];
        m2 += part[buf][j][w][1];
      }
      m1 /= (float)cols;
      m2 /= (float)cols;
      if (row0 + j < rows) {
        float xh[E], d[E], o[E];
        normalised(cur, j, xh);
        V::unpack(cur.d[j], d);
#pragma unroll
        for (int e = 0; e < E; ++e) o[e] = cur.rs[j] * fmaf(-xh[e], m2, fmaf(d[e], g[e], -m1));
        if constexpr (HAS_ADD) {
          float w[E];
          V::unpack(cur.add[j], w);
#pragma unroll
          for (int e = 0; e < E; ++e) o[e] += w[e];
        }
        V::store(dx + (row0 + j) * cols + col, o);
        if constexpr (DROP2) ln_store_dropped<T, V>(dr, (row0 + j) * cols + col, o);
      }
    }
    cur = nxt;
  }
  // contiguous atomics (see ln_bwd_vec_kernel): the lanes' 8-column partials go through LDS first
  extern __shared__ float cpart[];  // [2][cols]
  __syncthreads();
#pragma unroll
  for (int e = 0; e < E; ++e) {
    cpart[col + e] = ag[e];
    cpart[cols + col + e] = ab[e];
  }
  __syncthreads();
  for (int64_t c = threadIdx.x; c < 2 * cols; c += blockDim.x) atomicAdd(c < cols ? d_gamma + c : d_beta + (c - cols), cpart[c]);
}

template <typename T>
bool ln_vec_ok(const void* a, const void* b, const void* c, int64_t cols) {
  constexpr int E = Vec16<T>::N;
  auto al = [](const void* p) { return p == nullptr || ((uintptr_t)p % 16 == 0); };
  return cols % E == 0 && cols <= (int64_t)LNV_MAX * 64 * E && al(a) && al(b) && al(c);
}

template <typename T>
void ln_fwd_vec_launch(const void* x, const void* x2, const float* gamma, const float* beta, void* y, float* mean,
                       float* rstd, int64_t rows, int64_t cols, float eps, hipStream_t s) {
  constexpr int E = Vec16<T>::N;
  const int nv = (int)((cols + 64 * E - 1) / (64 * E));
  const int grid = grid_for(rows, 4, 1, 256 * 16);
#define LNF(NVV) hipLaunchKernelGGL((ln_fwd_vec_kernel<T, NVV>), dim3(grid), dim3(256), 0, s, (const T*)x, (const T*)x2, gamma, beta, (T*)y, mean, rstd, rows, cols, eps)
  switch (nv) { case 1: LNF(1); break; case 2: LNF(2); break; case 3: LNF(3); break; case 4: LNF(4); break;
                case 5: LNF(5); break; case 6: LNF(6); break; case 7: LNF(7); break; default: LNF(8); break; }
#undef LNF
}

template <typename T>
void ln_bwd_vec_launch(const void* dy, const void* x, const void* x2, const float* gamma, const float* mean,
                       const float* rstd, void* dx, const void* dx_add, float* dg, float* db, int64_t rows, int64_t cols,
                       hipStream_t s) {
  constexpr int E = Vec16<T>::N;
  const int nv = (int)((cols + 64 * E - 1) / (64 * E));
  if constexpr (sizeof(T) == 2) {
    // bf16 rows of 256, 768 or 1280 elements (odd multiples of 256: hidden 256 of the reference's defaults and its 5H rows, 768 of cfg 5): the
    // same pipelined one-wave-per-row kernel on 8-byte vectors (64 lanes x 4 elements per chunk) instead of the guarded form below
    if (cols % 256 == 0 && cols % 512 != 0 && cols <= 1280) {
      const int grid = grid_for(rows, 4, 8, 256 * 8);
      const size_t lds = 8 * cols * sizeof(float);
#define LN8(NVV, X2, ADD) hipLaunchKernelGGL((ln_bwd_vec_kernel<T, NVV, X2, ADD, false, Vec8<T>>), dim3(grid), dim3(256), lds, s, (const T*)dy, (const T*)x, (const T*)x2, gamma, mean, rstd, (T*)dx, (const T*)dx_add, dg, db, rows, cols)
#define LN8B(NVV) do { if (x2 && dx_add) LN8(NVV, true, true); else if (x2) LN8(NVV, true, false); else if (dx_add) LN8(NVV, false, true); else LN8(NVV, false, false); } while (0)
      if (cols == 256) LN8B(1);
      else if (cols == 768) LN8B(3);
      else LN8B(5);
#undef LN8B
#undef LN8
      return;
    }
    if (cols % 256 == 0 && cols % 512 != 0 && cols <= 15 * 256) {  // 3840 = cfg 5's 5H rows: the row-split kernel, fifteen waves of 256 columns
      constexpr int R = 2;
      const int nw = (int)(cols / 256);
      const int grid = grid_for(rows, 1, R * 16, 256 * 4);
#define LNS8(X2, ADD) hipLaunchKernelGGL((ln_bwd_split_kernel<T, R, X2, ADD, false, Vec8<T>, 16>), dim3(grid), dim3(64 * nw), 2 * cols * sizeof(float), s, (const T*)dy, (const T*)x, (const T*)x2, gamma, mean, rstd, (T*)dx, (const T*)dx_add, dg, db, rows, cols)
      if (x2 && dx_add) LNS8(true, true);
      else if (x2) LNS8(true, false);
      else if (dx_add) LNS8(false, true);
      else LNS8(false, false);
#undef LNS8
      return;
    }
  }
  if (cols != (int64_t)nv * 64 * E || nv > 8) {  // ragged last chunk: guarded kernel
    const int grid = grid_for(rows, 4, 16, 256 * 4);
    const size_t lds = 2 * cols * sizeof(float);
#define LNG(NVV) hipLaunchKernelGGL((ln_bwd_vec_guarded_kernel<T, NVV>), dim3(grid), dim3(256), lds, s, (const T*)dy, (const T*)x, (const T*)x2, gamma, mean, rstd, (T*)dx, (const T*)dx_add, dg, db, rows, cols)
    switch (nv) { case 1: LNG(1); break; case 2: LNG(2); break; case 3: LNG(3); break; case 4: LNG(4); break;
                  case 5: LNG(5); break; case 6: LNG(6); break; case 7: LNG(7); break; default: LNG(8); break; }
#undef LNG
    return;
  }
  if (nv >= 3) {  // wide rows: one workgroup of nv waves per row group
    constexpr int R = 2;
    const int grid = grid_for(rows, 1, R * 16, 256 * 4);  // >= 16 groups per workgroup (2 cols atomics each at the end)
#define LNS(X2, ADD) hipLaunchKernelGGL((ln_bwd_split_kernel<T, R, X2, ADD>), dim3(grid), dim3(64 * nv), 2 * cols * sizeof(float), s, (const T*)dy, (const T*)x, (const T*)x2, gamma, mean, rstd, (T*)dx, (const T*)dx_add, dg, db, rows, cols)
    if (x2 && dx_add) LNS(true, true);
    else if (x2) LNS(true, false);
    else if (dx_add) LNS(false, true);
    else LNS(false, false);
#undef LNS
    return;
  }
  // narrow rows: one wave per row, >= 8 rows per wave (2 cols atomics per workgroup at the end), up to 8 workgroups per CU
  const int grid = grid_for(rows, 4, 8, 256 * 8);
  const size_t lds = 8 * cols * sizeof(float);
#define LNB(NVV, X2, ADD) hipLaunchKernelGGL((ln_bwd_vec_kernel<T, NVV, X2, ADD>), dim3(grid), dim3(256), lds, s, (const T*)dy, (const T*)x, (const T*)x2, gamma, mean, rstd, (T*)dx, (const T*)dx_add, dg, db, rows, cols)
#define LNB2(NVV) do { if (x2 && dx_add) LNB(NVV, true, true); else if (x2) LNB(NVV, true, false); else if (dx_add) LNB(NVV, false, true); else LNB(NVV, false, false); } while (0)
  if (nv == 1) LNB2(1);
  else LNB2(2);
#undef LNB2
#undef LNB
}

// ---- softmax ----------------------------------------------------------------------------------
template <int TPR>  // threads per row: 64 (wave) or 256 (workgroup)
__device__ __forceinline__ float row_max(float v, float* red) {
  if constexpr (TPR == 64) return wave_max(v);
  else return block_max(v, red);
}
template <int TPR>
__device__ __forceinline__ float row_sum(float v, float* red) {
  if constexpr (TPR == 64) return wave_sum(v);
  else return block_sum(v, red);
}

template <typename TI, typename TO, int TPR>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const CaseSoftmaxDesc d, const TI* __restrict__ x,
                                                          const uint8_t* __restrict__ col_valid,
                                                          const uint8_t* __restrict__ row_valid, TO* __restrict__ p_out,
                                                          TO* __restrict__ y_out) {
  __shared__ float red[32];
  const int rows_per_block = 256 / TPR;
  const int sub = threadIdx.x / TPR, t = threadIdx.x % TPR;
  const int64_t total = d.outer * d.inner * d.R;
  const float keep_scale = d.drop_p > 0.f ? 1.f / (1.f - d.drop_p) : 1.f;
  const uint32_t thr = rng_threshold(d.drop_p);
  for (int64_t base = (int64_t)blockIdx.x * rows_per_block; base < total; base += (int64_t)gridDim.x * rows_per_block) {
    const int64_t row = base + sub;
    const bool live = row < total;  // keep every wave in the block-level reductions
    const int64_t rr = live ? row : 0;
    const int64_t r = rr % d.R, o = rr / (d.R * d.inner);
    const TI* xr = x + rr * d.C;
    const uint8_t* cv = col_valid ? col_valid + o * d.C : nullptr;
    const bool row_ok = live && (!row_valid || row_valid[o * d.R + r]);
    const int64_t cmax = d.causal ? (r + 1 < d.C ? r + 1 : d.C) : d.C;
    float m = -INFINITY;
    if (row_ok)
      for (int64_t c = t; c < cmax; c += TPR)
        if (!cv || cv[c]) m = fmaxf(m, Elem<TI>::ld(xr + c));
    m = row_max<TPR>(m, red);
    float s = 0.f;
    if (row_ok && m > -INFINITY)
      for (int64_t c = t; c < cmax; c += TPR)
        if (!cv || cv[c]) s += expf(Elem<TI>::ld(xr + c) - m);
    s = row_sum<TPR>(s, red);
    if (!live) continue;
    const float inv = s > 0.f ? 1.f / s : 0.f;
    const uint32_t rkey = d.drop_p > 0.f ? rng_row_key(d.seed, d.offset + rng_base_of(d.state) + (uint64_t)rr) : 0u;
    for (int64_t c = t; c < d.C; c += TPR) {
      float p = 0.f;
      if (row_ok && c < cmax && (!cv || cv[c]) && inv > 0.f) p = expf(Elem<TI>::ld(xr + c) - m) * inv;
      Elem<TO>::st(p_out + rr * d.C + c, p);
      if (d.drop_p > 0.f) Elem<TO>::st(y_out + rr * d.C + c, attn_keep(rkey, (uint32_t)c, thr) ? p * keep_scale : 0.f);
    }
  }
}

template <typename TI, typename TO, int TPR>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const CaseSoftmaxDesc d, const TO* __restrict__ dy,
                                                          const TO* __restrict__ p, TI* __restrict__ dx) {
  __shared__ float red[32];
  const int rows_per_block = 256 / TPR;
  const int sub = threadIdx.x / TPR, t = threadIdx.x % TPR;
  const int64_t total = d.outer * d.inner * d.R;
  const float keep_scale = d.drop_p > 0.f ? 1.f / (1.f - d.drop_p) : 1.f;
  const uint32_t thr = rng_threshold(d.drop_p);
  for (int64_t base = (int64_t)blockIdx.x * rows_per_block; base < total; base += (int64_t)gridDim.x * rows_per_block) {
    const int64_t row = base + sub;
    const bool live = row < total;
    const int64_t rr = live ? row : 0;
    const uint32_t rkey = d.drop_p > 0.f ? rng_row_key(d.seed, d.offset + rng_base_of(d.state) + (uint64_t)rr) : 0u;
    float dot = 0.f;
    if (live)
      for (int64_t c = t; c < d.C; c += TPR) {
        float g = Elem<TO>::ld(dy + rr * d.C + c);
        if (d.drop_p > 0.f) g = attn_keep(rkey, (uint32_t)c, thr) ? g * keep_scale : 0.f;
        dot += g * Elem<TO>::ld(p + rr * d.C + c);
      }
    dot = row_sum<TPR>(dot, red);
    if (!live) continue;
    for (int64_t c = t; c < d.C; c += TPR) {
      float g = Elem<TO>::ld(dy + rr * d.C + c);
      if (d.drop_p > 0.f) g = attn_keep(rkey, (uint32_t)c, thr) ? g * keep_scale : 0.f;
      const float pv = Elem<TO>::ld(p + rr * d.C + c);
      Elem<TI>::st(dx + rr * d.C + c, pv * (g - dot));
    }
  }
}

// bf16 rows of up to 1024 columns (C % 8 == 0): one wave per row, the row lives in registers (NCH 16-byte chunks per lane),
// one global read and one write per element.  The scalar kernels above read every row three times with 2-byte accesses
// (1.3 TB/s on the 384-wide attention rows of the unfused head_dim-320 backward).
// The forward also takes f32 rows (the unfused attention keeps its scores in f32 and writes bf16 probabilities).
template <typename TI, int NCH>
__global__ __launch_bounds__(256) void softmax_fwd_vec_kernel(const CaseSoftmaxDesc d, const TI* __restrict__ x,
                                                              const uint8_t* __restrict__ col_valid,
                                                              const uint8_t* __restrict__ row_valid, bf16_t* __restrict__ p_out,
                                                              bf16_t* __restrict__ y_out) {
  const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
  const int64_t total = d.outer * d.inner * d.R;
  const float keep_scale = d.drop_p > 0.f ? 1.f / (1.f - d.drop_p) : 1.f;
  for (int64_t row = (int64_t)blockIdx.x * 4 + sub; row < total; row += (int64_t)gridDim.x * 4) {
    const int64_t r = row % d.R, o = row / (d.R * d.inner);
    const uint8_t* cv = col_valid ? col_valid + o * d.C : nullptr;
    const bool row_ok = !row_valid || row_valid[o * d.R + r];
    const int64_t cmax = d.causal ? (r + 1 < d.C ? r + 1 : d.C) : d.C;
    float v[NCH][8];
    bool ok[NCH][8];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int64_t c0 = (int64_t)(lane + 64 * j) * 8;
      const bool in = c0 < d.C;
      if (in) {
        if constexpr (sizeof(TI) == 2) {
          Vec16<bf16_t>::load(x + row * d.C + c0, v[j]);
        } else {
          float lo[4], hi[4];
          Vec16<float>::load(x + row * d.C + c0, lo);
          Vec16<float>::load(x + row * d.C + c0 + 4, hi);
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[j][e] = lo[e]; v[j][4 + e] = hi[e]; }
        }
      }
      uint64_t cvw = ~0ull;
      if (in && cv) cvw = *reinterpret_cast<const uint64_t*>(cv + c0);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        ok[j][e] = in && row_ok && (c0 + e < cmax) && ((cvw >> (8 * e)) & 0xff);
        if (ok[j][e]) m = fmaxf(m, v[j][e]);
      }
    }
    m = wave_max(m);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[j][e] = ok[j][e] ? expf(v[j][e] - m) : 0.f;
        s += v[j][e];
      }
    s = wave_sum(s);
    const float inv = s > 0.f ? 1.f / s : 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int64_t c0 = (int64_t)(lane + 64 * j) * 8;
      if (c0 >= d.C) continue;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[j][e] *= inv;
      Vec16<bf16_t>::store(p_out + row * d.C + c0, v[j]);
      if (d.drop_p > 0.f) {
        attn_dropout8(v[j], rng_row_key(d.seed, d.offset + rng_base_of(d.state) + (uint64_t)row), (uint32_t)c0, rng_threshold(d.drop_p), keep_scale);
        Vec16<bf16_t>::store(y_out + row * d.C + c0, v[j]);
      }
    }
  }
}

template <int NCH>
__global__ __launch_bounds__(256) void softmax_bwd_vec_kernel(const CaseSoftmaxDesc d, const bf16_t* __restrict__ dy,
                                                              const bf16_t* __restrict__ p, bf16_t* __restrict__ dx) {
  const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
  const int64_t total = d.outer * d.inner * d.R;
  const float keep_scale = d.drop_p > 0.f ? 1.f / (1.f - d.drop_p) : 1.f;
  for (int64_t row = (int64_t)blockIdx.x * 4 + sub; row < total; row += (int64_t)gridDim.x * 4) {
    float g[NCH][8], pv[NCH][8];
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int64_t c0 = (int64_t)(lane + 64 * j) * 8;
      if (c0 < d.C) {
        Vec16<bf16_t>::load(dy + row * d.C + c0, g[j]);
        Vec16<bf16_t>::load(p + row * d.C + c0, pv[j]);
        if (d.drop_p > 0.f) attn_dropout8(g[j], rng_row_key(d.seed, d.offset + rng_base_of(d.state) + (uint64_t)row), (uint32_t)c0, rng_threshold(d.drop_p), keep_scale);
#pragma unroll
        for (int e = 0; e < 8; ++e) dot += g[j][e] * pv[j][e];
      }
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int64_t c0 = (int64_t)(lane + 64 * j) * 8;
      if (c0 >= d.C) continue;
#pragma unroll
      for (int e = 0; e < 8; ++e) g[j][e] = pv[j][e] * (g[j][e] - dot);
      Vec16<bf16_t>::store(dx + row * d.C + c0, g[j]);
    }
  }
}

static inline bool aligned16(const void* p) { return p == nullptr || ((uintptr_t)p % 16) == 0; }

template <typename TI, typename TO>
int softmax_launch(const CaseSoftmaxDesc* d, bool fwd, const void* a, const uint8_t* cv, const uint8_t* rv, void* b,
                   void* c, hipStream_t s) {
  const int64_t total = d->outer * d->inner * d->R;
  if constexpr (sizeof(TO) == 2) {
    if ((fwd || sizeof(TI) == 2) && d->C % 8 == 0 && d->C <= 1024 && aligned16(a) && aligned16(b) && aligned16(c)) {
      const int grid = grid_for(total, 4, 1, 256 * 16);
      if (d->C <= 512) {
        if (fwd) hipLaunchKernelGGL((softmax_fwd_vec_kernel<TI, 1>), dim3(grid), dim3(256), 0, s, *d, (const TI*)a, cv, rv, (bf16_t*)b, (bf16_t*)c);
        else hipLaunchKernelGGL((softmax_bwd_vec_kernel<1>), dim3(grid), dim3(256), 0, s, *d, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)c);
      } else {
        if (fwd) hipLaunchKernelGGL((softmax_fwd_vec_kernel<TI, 2>), dim3(grid), dim3(256), 0, s, *d, (const TI*)a, cv, rv, (bf16_t*)b, (bf16_t*)c);
        else hipLaunchKernelGGL((softmax_bwd_vec_kernel<2>), dim3(grid), dim3(256), 0, s, *d, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)c);
      }
      return case_check_launch(fwd ? "case_softmax_fwd" : "case_softmax_bwd");
    }
  }
  if (d->C <= 1024) {
    const int grid = grid_for(total, 4, 1, 256 * 16);
    if (fwd) hipLaunchKernelGGL((softmax_fwd_kernel<TI, TO, 64>), dim3(grid), dim3(256), 0, s, *d, (const TI*)a, cv, rv, (TO*)b, (TO*)c);
    else hipLaunchKernelGGL((softmax_bwd_kernel<TI, TO, 64>), dim3(grid), dim3(256), 0, s, *d, (const TO*)a, (const TO*)b, (TI*)c);
  } else {
    const int grid = grid_for(total, 1, 1, 256 * 8);
    if (fwd) hipLaunchKernelGGL((softmax_fwd_kernel<TI, TO, 256>), dim3(grid), dim3(256), 0, s, *d, (const TI*)a, cv, rv, (TO*)b, (TO*)c);
    else hipLaunchKernelGGL((softmax_bwd_kernel<TI, TO, 256>), dim3(grid), dim3(256), 0, s, *d, (const TO*)a, (const TO*)b, (TI*)c);
  }
  return case_check_launch(fwd ? "case_softmax_fwd" : "case_softmax_bwd");
}

template <bool FWD>
int softmax_dispatch(const CaseSoftmaxDesc* d, const void* a, const uint8_t* cv, const uint8_t* rv, void* b, void* c,
                     hipStream_t s) {
  const int i = d->in_dtype, o = d->out_dtype;
  if (i == CASE_F32 && o == CASE_F32) return softmax_launch<float, float>(d, FWD, a, cv, rv, b, c, s);
  if (i == CASE_BF16 && o == CASE_BF16) return softmax_launch<bf16_t, bf16_t>(d, FWD, a, cv, rv, b, c, s);
  if (i == CASE_BF16 && o == CASE_F32) return softmax_launch<bf16_t, float>(d, FWD, a, cv, rv, b, c, s);
  if (i == CASE_F32 && o == CASE_BF16) return softmax_launch<float, bf16_t>(d, FWD, a, cv, rv, b, c, s);
  return case_set_error(CASE_E_UNSUPPORTED, "case_softmax: dtype combination");
}

}  // namespace

extern "C" int case_layernorm_fwd(const void* x, const void* x2, const float* gamma, const float* beta, void* y,
                                  float* mean, float* rstd, int64_t rows, int64_t cols, float eps, int32_t dtype,
                                  case_stream_t stream) {
  CASE_REQUIRE(x && gamma && beta && y && mean && rstd && rows > 0 && cols > 0, "case_layernorm_fwd: bad argument");
  const int grid = grid_for(rows, 1);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CASE_F32 && ln_vec_ok<float>(x, x2, y, cols) && (uintptr_t)gamma % 16 == 0 && (uintptr_t)beta % 16 == 0) {
    ln_fwd_vec_launch<float>(x, x2, gamma, beta, y, mean, rstd, rows, cols, eps, s);
    return case_check_launch("case_layernorm_fwd");
  }
  if (dtype == CASE_BF16 && ln_vec_ok<bf16_t>(x, x2, y, cols) && (uintptr_t)gamma % 16 == 0 && (uintptr_t)beta % 16 == 0) {
    ln_fwd_vec_launch<bf16_t>(x, x2, gamma, beta, y, mean, rstd, rows, cols, eps, s);
    return case_check_launch("case_layernorm_fwd");
  }
  if (dtype == CASE_F32)
    hipLaunchKernelGGL(ln_fwd_kernel<float>, dim3(grid), dim3(LN_THREADS), 0, s, (const float*)x, (const float*)x2, gamma,
                       beta, (float*)y, mean, rstd, rows, cols, eps);
  else
    hipLaunchKernelGGL(ln_fwd_kernel<bf16_t>, dim3(grid), dim3(LN_THREADS), 0, s, (const bf16_t*)x, (const bf16_t*)x2,
                       gamma, beta, (bf16_t*)y, mean, rstd, rows, cols, eps);
  return case_check_launch("case_layernorm_fwd");
}

typedef float f32x4_r __attribute__((ext_vector_type(4)));
// The 5H-wide rows of case_layernorm_bwd_dropout in the same one-wave-per-row form (the row-split kernel above: 0.68 ms on 122 880 rows,
// two barriers' worth of synchronisation per row pair): dx and the dropout-masked copy, 10 vectors per lane and row in, 10 out.
__global__ __launch_bounds__(256) void ln_bwd_drop_rows5_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                                const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, bf16_t* __restrict__ dx,
                                                                float* __restrict__ d_gamma, float* __restrict__ d_beta, int64_t rows,
                                                                const LnDrop dr) {
  typedef bf16_t T;
  constexpr int E = 8, H = 512, COLS = 5 * H;
  __shared__ float sh[2 * COLS];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int c = threadIdx.x; c < COLS; c += 256) sh[c] = gamma[c];
  __syncthreads();
  const int64_t wave = (int64_t)blockIdx.x * 4 + wid, nwaves = (int64_t)gridDim.x * 4;
  const int pc = lane * E;
  float ag[5][E], ab[5][E];
#pragma unroll
  for (int p = 0; p < 5; ++p)
#pragma unroll
    for (int e = 0; e < E; ++e) ag[p][e] = ab[p][e] = 0.f;
  for (int64_t r = wave; r < rows; r += nwaves) {
    uint4 xv[5], dv[5];
#pragma unroll
    for (int p = 0; p < 5; ++p) {
      xv[p] = *reinterpret_cast<const uint4*>(x + r * COLS + pc + p * H);
      dv[p] = *reinterpret_cast<const uint4*>(dy + r * COLS + pc + p * H);
    }
    const float mu = mean[r], rs = rstd[r];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int p = 0; p < 5; ++p) {
      float xh[E], d[E];
      Vec16<T>::unpack(xv[p], xh);
      Vec16<T>::unpack(dv[p], d);
      const f32x4_r g0 = *reinterpret_cast<const f32x4_r*>(sh + p * H + pc), g1 = *reinterpret_cast<const f32x4_r*>(sh + p * H + pc + 4);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const float xn = (xh[e] - mu) * rs, gy = d[e] * (e < 4 ? g0[e & 3] : g1[e & 3]);
        s1 += gy;
        s2 += gy * xn;
        ag[p][e] += d[e] * xn;
        ab[p][e] += d[e];
      }
    }
    const float m1 = wave_sum(s1) * (1.f / COLS), m2 = wave_sum(s2) * (1.f / COLS);
#pragma unroll
    for (int p = 0; p < 5; ++p) {
      float xh[E], d[E], o[E];
      Vec16<T>::unpack(xv[p], xh);
      Vec16<T>::unpack(dv[p], d);
      const f32x4_r g0 = *reinterpret_cast<const f32x4_r*>(sh + p * H + pc), g1 = *reinterpret_cast<const f32x4_r*>(sh + p * H + pc + 4);
#pragma unroll
      for (int e = 0; e < E; ++e) o[e] = rs * (d[e] * (e < 4 ? g0[e & 3] : g1[e & 3]) - m1 - (xh[e] - mu) * rs * m2);
      Vec16<T>::store(dx + r * COLS + pc + p * H, o);
      ln_store_dropped<T>(dr, r * COLS + pc + p * H, o);
    }
  }
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wid == w) {
#pragma unroll
      for (int p = 0; p < 5; ++p)
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int c = p * H + pc + e;
          sh[c] = (w == 0 ? 0.f : sh[c]) + ag[p][e];
          sh[COLS + c] = (w == 0 ? 0.f : sh[COLS + c]) + ab[p][e];
        }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * COLS; c += 256) atomicAdd(c < COLS ? d_gamma + c : d_beta + (c - COLS), sh[c]);
}

// the dual-output form: full 64-lane chunks only (cols = nv * 64 * E, nv <= 8), no second input, no carried gradient
template <typename T>
bool ln_bwd_drop_launch(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx, float* dg,
                        float* db, int64_t rows, int64_t cols, const LnDrop& dr, hipStream_t s) {
  constexpr int E = Vec16<T>::N;
  const int nv = (int)((cols + 64 * E - 1) / (64 * E));
  const T* nul = nullptr;
  if constexpr (sizeof(T) == 2) {
    if (cols % 256 == 0 && cols % 512 != 0 && cols <= 1280) {  // 8-byte-vector form, see ln_bwd_vec_launch
      const int grid = grid_for(rows, 4, 8, 256 * 8);
      const size_t lds = 8 * cols * sizeof(float);
#define LN8D(NVV) hipLaunchKernelGGL((ln_bwd_vec_kernel<T, NVV, false, false, true, Vec8<T>>), dim3(grid), dim3(256), lds, s, (const T*)dy, (const T*)x, nul, gamma, mean, rstd, (T*)dx, nul, dg, db, rows, cols, dr)
      if (cols == 256) LN8D(1);
      else if (cols == 768) LN8D(3);
      else LN8D(5);
#undef LN8D
      return true;
    }
    if (cols % 256 == 0 && cols % 512 != 0 && cols <= 15 * 256) {
      constexpr int R = 2;
      hipLaunchKernelGGL((ln_bwd_split_kernel<T, R, false, false, true, Vec8<T>, 16>), dim3(grid_for(rows, 1, R * 16, 256 * 4)), dim3(64 * (int)(cols / 256)),
                         2 * cols * sizeof(float), s, (const T*)dy, (const T*)x, nul, gamma, mean, rstd, (T*)dx, nul, dg, db, rows, cols, dr);
      return true;
    }
  }
  if (cols != (int64_t)nv * 64 * E || nv > 8) return false;
  if constexpr (sizeof(T) == 2) {
    if (nv == 5) {  // the 5H rows of the CaSE / Masque blocks: one wave per row (see ln_bwd_drop_rows5_kernel)
      hipLaunchKernelGGL(ln_bwd_drop_rows5_kernel, dim3(grid_for(rows, 4, 8, 256 * 8)), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, gamma,
                         mean, rstd, (bf16_t*)dx, dg, db, rows, dr);
      return true;
    }
  }
  if (nv >= 3) {
    constexpr int R = 2;
    const int grid = grid_for(rows, 1, R * 16, 256 * 4);
    hipLaunchKernelGGL((ln_bwd_split_kernel<T, R, false, false, true>), dim3(grid), dim3(64 * nv), 2 * cols * sizeof(float), s, (const T*)dy,
                       (const T*)x, nul, gamma, mean, rstd, (T*)dx, nul, dg, db, rows, cols, dr);
    return true;
  }
  const int grid = grid_for(rows, 4, 8, 256 * 8);
  const size_t lds = 8 * cols * sizeof(float);
  if (nv == 1)
    hipLaunchKernelGGL((ln_bwd_vec_kernel<T, 1, false, false, true>), dim3(grid), dim3(256), lds, s, (const T*)dy, (const T*)x, nul, gamma, mean,
                       rstd, (T*)dx, nul, dg, db, rows, cols, dr);
  else
    hipLaunchKernelGGL((ln_bwd_vec_kernel<T, 2, false, false, true>), dim3(grid), dim3(256), lds, s, (const T*)dy, (const T*)x, nul, gamma, mean,
                       rstd, (T*)dx, nul, dg, db, rows, cols, dr);
  return true;
}

// LayerNorm backward of the 5H rows G = [E | A1 | A2 | E o A1 | E o A2] (common/Interaction.py:65-72 -> common/TransformerBlock.py:25) FUSED with the
// backward of that concatenation:
//     dE = dG0 + dG3 o A1 + dG4 o A2,   dA1 = dG1 + dG3 o E,   dA2 = dG2 + dG4 o E        (0 on padded rows; dG rounded to bf16 first,
// as the two-kernel path stores it) -- dG is never written: 630 MB less to write and 630 MB less to read per call at cfg 2.
// ONE WAVE PER ROW and no barrier in the row loop: lane l owns columns 8 l .. 8 l + 7 of EVERY piece, so the concatenation's backward is
// lane-local and the only cross-lane traffic is the two row sums.  15 vectors per lane and row in flight (x, dy, dx_add of five pieces);
// the per-lane gamma / beta partial sums (80 registers) leave no room for a prefetched second row (256 registers, one wave per SIMD) and
// it does not need one: 0.48 ms at 122 880 rows = 4.7 TB/s, against 0.88 ms for case_layernorm_bwd + case_concat5_bwd.  (A row-split
// form -- five waves per row, pieces exchanged through LDS, two barriers per row pair -- ran at 2.2 TB/s: 1.02 ms.)  gamma sits in LDS.
template <bool HAS_ADD>
__global__ __launch_bounds__(256) void ln_bwd_concat5_rows_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                                  const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd, const bf16_t* __restrict__ dx_add,
                                                                  const uint8_t* __restrict__ row_valid, bf16_t* __restrict__ de,
                                                                  bf16_t* __restrict__ da1, bf16_t* __restrict__ da2,
                                                                  float* __restrict__ d_gamma, float* __restrict__ d_beta, int64_t rows) {
  typedef bf16_t T;
  constexpr int E = 8, H = 512, COLS = 5 * H;
  __shared__ float sh[2 * COLS];  // gamma during the row loop (first half), the column sums afterwards
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int c = threadIdx.x; c < COLS; c += 256) sh[c] = gamma[c];
  __syncthreads();
  const int64_t wave = (int64_t)blockIdx.x * 4 + wid, nwaves = (int64_t)gridDim.x * 4;
  const int pc = lane * E;
  float ag[5][E], ab[5][E];
#pragma unroll
  for (int p = 0; p < 5; ++p)
#pragma unroll
    for (int e = 0; e < E; ++e) ag[p][e] = ab[p][e] = 0.f;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const bf16_t* xr = x + r * COLS + pc;
    const bf16_t* dr = dy + r * COLS + pc;
    uint4 xv[5], dv[5], av[HAS_ADD ? 5 : 1];
#pragma unroll
    for (int p = 0; p < 5; ++p) {
      xv[p] = *reinterpret_cast<const uint4*>(xr + p * H);
      dv[p] = *reinterpret_cast<const uint4*>(dr + p * H);
      if constexpr (HAS_ADD) av[p] = *reinterpret_cast<const uint4*>(dx_add + r * COLS + pc + p * H);
    }
    const float mu = mean[r], rs = rstd[r];
    const bool ok = row_valid[r] != 0;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int p = 0; p < 5; ++p) {
      float xh[E], d[E];
      Vec16<T>::unpack(xv[p], xh);
      Vec16<T>::unpack(dv[p], d);
      const f32x4_r g0 = *reinterpret_cast<const f32x4_r*>(sh + p * H + pc), g1 = *reinterpret_cast<const f32x4_r*>(sh + p * H + pc + 4);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const float xn = (xh[e] - mu) * rs, gy = d[e] * (e < 4 ? g0[e & 3] : g1[e & 3]);
        s1 += gy;
        s2 += gy * xn;
        ag[p][e] += d[e] * xn;
        ab[p][e] += d[e];
      }
    }
    const float m1 = wave_sum(s1) * (1.f / COLS), m2 = wave_sum(s2) * (1.f / COLS);
    float og[5][E];  // dG of the lane's columns, rounded to bf16 (what the two-kernel path stores and reads back)
#pragma unroll
    for (int p = 0; p < 5; ++p) {
      float xh[E], d[E];
      Vec16<T>::unpack(xv[p], xh);
      Vec16<T>::unpack(dv[p], d);
      const f32x4_r g0 = *reinterpret_cast<const f32x4_r*>(sh + p * H + pc), g1 = *reinterpret_cast<const f32x4_r*>(sh + p * H + pc + 4);
      float o[E];
#pragma unroll
      for (int e = 0; e < E; ++e) o[e] = rs * (d[e] * (e < 4 ? g0[e & 3] : g1[e & 3]) - m1 - (xh[e] - mu) * rs * m2);
      if constexpr (HAS_ADD) {
        float w[E];
        Vec16<T>::unpack(av[p], w);
#pragma unroll
        for (int e = 0; e < E; ++e) o[e] += w[e];
      }
      Vec16<T>::unpack(make_uint4(f32x2_to_bf16x2(o[0], o[1]), f32x2_to_bf16x2(o[2], o[3]), f32x2_to_bf16x2(o[4], o[5]), f32x2_to_bf16x2(o[6], o[7])),
                       og[p]);
    }
    float ev[E], x1[E], x2[E], oe[E], o1[E], o2[E];
    Vec16<T>::unpack(xv[0], ev);
    Vec16<T>::unpack(xv[1], x1);
    Vec16<T>::unpack(xv[2], x2);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      oe[e] = ok ? og[0][e] + og[3][e] * x1[e] + og[4][e] * x2[e] : 0.f;
      o1[e] = ok ? og[1][e] + og[3][e] * ev[e] : 0.f;
      o2[e] = ok ? og[2][e] + og[4][e] * ev[e] : 0.f;
    }
    Vec16<T>::store(de + r * H + pc, oe);
    Vec16<T>::store(da1 + r * H + pc, o1);
    Vec16<T>::store(da2 + r * H + pc, o2);
  }
  // column sums: the four waves add their partials into LDS one after the other, then contiguous atomics
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wid == w) {
#pragma unroll
      for (int p = 0; p < 5; ++p)
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int c = p * H + pc + e;
          sh[c] = (w == 0 ? 0.f : sh[c]) + ag[p][e];
          sh[COLS + c] = (w == 0 ? 0.f : sh[COLS + c]) + ab[p][e];
        }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * COLS; c += 256) atomicAdd(c < COLS ? d_gamma + c : d_beta + (c - COLS), sh[c]);
}

extern "C" int case_layernorm_bwd_concat5(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                          const void* dx_add, const void* e, const void* a1, const void* a2, const uint8_t* row_valid, void* de,
                                          void* da1, void* da2, float* d_gamma, float* d_beta, int64_t rows, int64_t H, int32_t dtype,
                                          case_stream_t stream) {
  CASE_REQUIRE(dy && x && gamma && mean && rstd && e && a1 && a2 && row_valid && de && da1 && da2 && d_gamma && d_beta && rows > 0,
               "case_layernorm_bwd_concat5: bad argument");
  if (dtype != CASE_BF16 || H != 512)
    return case_set_error(CASE_E_UNSUPPORTED, "case_layernorm_bwd_concat5: built for bf16 rows of 5 x 512 (run case_layernorm_bwd + case_concat5_bwd)");
  for (const void* p : {dy, x, dx_add, (const void*)de, (const void*)da1, (const void*)da2})
    CASE_REQUIRE((reinterpret_cast<uintptr_t>(p) & 15) == 0, "case_layernorm_bwd_concat5: tensors must be 16-byte aligned");
  (void)e; (void)a1; (void)a2;  // the pieces are read back from the columns of x = G that hold them
  hipStream_t s = (hipStream_t)stream;
  const int grid_r = grid_for(rows, 4, 8, 256 * 8);
  if (dx_add)
    hipLaunchKernelGGL(ln_bwd_concat5_rows_kernel<true>, dim3(grid_r), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, gamma, mean, rstd,
                       (const bf16_t*)dx_add, row_valid, (bf16_t*)de, (bf16_t*)da1, (bf16_t*)da2, d_gamma, d_beta, rows);
  else
    hipLaunchKernelGGL(ln_bwd_concat5_rows_kernel<false>, dim3(grid_r), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, gamma, mean, rstd,
                       (const bf16_t*)nullptr, row_valid, (bf16_t*)de, (bf16_t*)da1, (bf16_t*)da2, d_gamma, d_beta, rows);
  return case_check_launch("case_layernorm_bwd_concat5");
}

extern "C" int case_layernorm_bwd_dropout(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx,
                                          void* dx_dropped, float* d_gamma, float* d_beta, int64_t rows, int64_t cols, float p, uint64_t seed,
                                          uint64_t offset, const CaseStepState* state, int32_t dtype, case_stream_t stream) {
  CASE_REQUIRE(dy && x && gamma && mean && rstd && dx && dx_dropped && d_gamma && d_beta && rows > 0 && cols > 0 && p > 0.f && p < 1.f,
               "case_layernorm_bwd_dropout: bad argument");
  // the dual-output kernels draw one 64-bit hash per element PAIR (offset + even index); case_dropout does the same only for even offsets
  CASE_REQUIRE((offset & 1) == 0, "case_layernorm_bwd_dropout: the RNG offset must be even (the mask is drawn per element pair)");
  const LnDrop dr = {dx_dropped, p, seed, offset, state};
  hipStream_t s = (hipStream_t)stream;
  bool ok = false;
  if (dtype == CASE_F32 && ln_vec_ok<float>(dy, x, dx, cols) && ln_vec_ok<float>(dx_dropped, nullptr, nullptr, cols))
    ok = ln_bwd_drop_launch<float>(dy, x, gamma, mean, rstd, dx, d_gamma, d_beta, rows, cols, dr, s);
  else if (dtype == CASE_BF16 && ln_vec_ok<bf16_t>(dy, x, dx, cols) && ln_vec_ok<bf16_t>(dx_dropped, nullptr, nullptr, cols))
    ok = ln_bwd_drop_launch<bf16_t>(dy, x, gamma, mean, rstd, dx, d_gamma, d_beta, rows, cols, dr, s);
  if (!ok)
    return case_set_error(CASE_E_UNSUPPORTED, "case_layernorm_bwd_dropout: needs 16-byte aligned rows of a whole number of 64-lane chunks "
                                              "(cols = %d x k, k <= 8); run case_layernorm_bwd + case_dropout", dtype == CASE_BF16 ? 512 : 256);
  return case_check_launch("case_layernorm_bwd_dropout");
}

extern "C" int case_layernorm_bwd(const void* dy, const void* x, const void* x2, const float* gamma, const float* mean,
                                  const float* rstd, void* dx, const void* dx_add, float* d_gamma, float* d_beta,
                                  int64_t rows, int64_t cols, int32_t dtype, case_stream_t stream) {
  CASE_REQUIRE(dy && x && gamma && mean && rstd && dx && d_gamma && d_beta && rows > 0 && cols > 0,
               "case_layernorm_bwd: bad argument");
  CASE_REQUIRE(cols <= (int64_t)LN_THREADS * LN_MAXPT, "case_layernorm_bwd: cols %lld > %d", (long long)cols,
               LN_THREADS * LN_MAXPT);
  const int grid = grid_for(rows, 1, 8, 256 * 2);  // few workgroups -> few atomics on d_gamma / d_beta
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CASE_F32 && ln_vec_ok<float>(dy, x, dx, cols) && ln_vec_ok<float>(x2, dx_add, nullptr, cols)) {
    ln_bwd_vec_launch<float>(dy, x, x2, gamma, mean, rstd, dx, dx_add, d_gamma, d_beta, rows, cols, s);
    return case_check_launch("case_layernorm_bwd");
  }
  if (dtype == CASE_BF16 && ln_vec_ok<bf16_t>(dy, x, dx, cols) && ln_vec_ok<bf16_t>(x2, dx_add, nullptr, cols)) {
    ln_bwd_vec_launch<bf16_t>(dy, x, x2, gamma, mean, rstd, dx, dx_add, d_gamma, d_beta, rows, cols, s);
    return case_check_launch("case_layernorm_bwd");
  }
  if (dtype == CASE_F32)
    hipLaunchKernelGGL(ln_bwd_kernel<float>, dim3(grid), dim3(LN_THREADS), 0, s, (const float*)dy, (const float*)x,
                       (const float*)x2, gamma, mean, rstd, (float*)dx, (const float*)dx_add, d_gamma, d_beta, rows, cols);
  else
    hipLaunchKernelGGL(ln_bwd_kernel<bf16_t>, dim3(grid), dim3(LN_THREADS), 0, s, (const bf16_t*)dy, (const bf16_t*)x,
                       (const bf16_t*)x2, gamma, mean, rstd, (bf16_t*)dx, (const bf16_t*)dx_add, d_gamma, d_beta, rows, cols);
  return case_check_launch("case_layernorm_bwd");
}

extern "C" int case_softmax_fwd(const CaseSoftmaxDesc* d, const void* x, const uint8_t* col_valid,
                                const uint8_t* row_valid, void* p_out, void* y_out, case_stream_t stream) {
  CASE_REQUIRE(d && x && p_out && d->outer > 0 && d->inner > 0 && d->R > 0 && d->C > 0, "case_softmax_fwd: bad argument");
  CASE_REQUIRE(d->drop_p == 0.f || y_out, "case_softmax_fwd: dropout needs y_out");
  CASE_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f, "case_softmax_fwd: drop_p out of range");
  return softmax_dispatch<true>(d, x, col_valid, row_valid, p_out, y_out, (hipStream_t)stream);
}

extern "C" int case_softmax_bwd(const CaseSoftmaxDesc* d, const void* dy, const void* p, void* dx, case_stream_t stream) {
  CASE_REQUIRE(d && dy && p && dx && d->outer > 0 && d->inner > 0 && d->R > 0 && d->C > 0, "case_softmax_bwd: bad argument");
  return softmax_dispatch<false>(d, dy, nullptr, nullptr, (void*)p, dx, (hipStream_t)stream);
}
