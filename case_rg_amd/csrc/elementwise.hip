// HBM-bound elementwise / gather / small-reduction kernels: K1 embed+pos, K9 masked mean, Interaction
// feature assembly (K8 epilogue), Highway gate (K14 epilogue), residual add, dropout, row masking,
// bias-gradient column sums and dtype casts.  Grid-stride, consecutive lanes on consecutive addresses.
#include "common.h"

namespace {

constexpr int EW_THREADS = 256;

template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    Elem<T>::st(o + i, Elem<T>::ld(a + i) + Elem<T>::ld(b + i));
}

template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, float p, uint64_t seed,
                               uint64_t offset, const CaseStepState* state) {
  offset += rng_base_of(state);
  const float scale = 1.f / (1.f - p);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    Elem<T>::st(y + i, rng_uniform(seed, offset + (uint64_t)i) >= p ? Elem<T>::ld(x + i) * scale : 0.f);
}

template <typename T>
__global__ void mask_rows_kernel(const T* __restrict__ x, const uint8_t* __restrict__ valid, T* __restrict__ y,
                                 int64_t rows, int64_t cols) {
  const int64_t n = rows * cols;
  const bool in_place = x == y;  // only the invalid rows are touched
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const bool ok = valid[i / cols] != 0;
    if (in_place) {
      if (!ok) Elem<T>::st(y + i, 0.f);
    } else {
      Elem<T>::st(y + i, ok ? Elem<T>::ld(x + i) : 0.f);
    }
  }
}

// column sums: workgroup = 64 columns x 4 row-lanes; each workgroup walks a strided slice of the rows
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* __restrict__ out, int64_t rows,
                                                     int64_t cols, int row_splits) {
  __shared__ float part[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int64_t col = (int64_t)(blockIdx.x / row_splits) * 64 + cx;
  const int split = blockIdx.x % row_splits;
  float s = 0.f;
  if (col < cols)
    for (int64_t r = (int64_t)split * 4 + ry; r < rows; r += (int64_t)row_splits * 4) s += Elem<T>::ld(x + r * cols + col);
  part[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && col < cols) atomicAdd(out + col, part[0][cx] + part[1][cx] + part[2][cx] + part[3][cx]);
}

// ---- 16-byte vector variants (taken when the pointers are 16-byte aligned and the row width is a multiple of the vector) --
// column sums: workgroup = 32 column chunks (of E columns) x 8 row lanes
template <typename T>
__global__ __launch_bounds__(256) void colsum_vec_kernel(const T* __restrict__ x, float* __restrict__ out, int64_t rows,
                                                         int64_t cols, int row_splits) {
  constexpr int E = Vec16<T>::N;
  __shared__ float part[8][32 * E + 1];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int64_t col = ((int64_t)(blockIdx.x / row_splits) * 32 + cx) * E;
  const int split = blockIdx.x % row_splits;
  float s[E];
#pragma unroll
  for (int e = 0; e < E; ++e) s[e] = 0.f;
  if (col < cols)
    for (int64_t r = (int64_t)split * 8 + ry; r < rows; r += (int64_t)row_splits * 8) {
      float v[E];
      Vec16<T>::load(x + r * cols + col, v);
#pragma unroll
      for (int e = 0; e < E; ++e) s[e] += v[e];
    }
#pragma unroll
  for (int e = 0; e < E; ++e) part[ry][cx * E + e] = s[e];
  __syncthreads();
  for (int c = threadIdx.x; c < 32 * E; c += 256) {
    const int64_t gc = (int64_t)(blockIdx.x / row_splits) * 32 * E + c;
    if (gc < cols) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) t += part[r][c];
      atomicAdd(out + gc, t);
    }
  }
}

template <typename T>
__global__ void add_vec_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o, int64_t nvec) {
  constexpr int E = Vec16<T>::N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    float x[E], y[E];
    Vec16<T>::load(a + i * E, x);
    Vec16<T>::load(b + i * E, y);
#pragma unroll
    for (int e = 0; e < E; ++e) x[e] += y[e];
    Vec16<T>::store(o + i * E, x);
  }
}

// out = src[0] + ... + src[n - 1] (2 <= n <= 8), summed in f32 and rounded once: the gradient of a tensor with n consumers in one pass
// (autograd's own accumulation is n - 1 binary adds: 3 (n - 1) tensor passes against n + 1 here)
struct AddNArgs {
  const void* src[8];
  int n;
};
template <typename T>
__global__ void add_n_vec_kernel(const AddNArgs a, T* __restrict__ o, int64_t nvec) {
  constexpr int E = Vec16<T>::N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    float acc[E], y[E];
    Vec16<T>::load(reinterpret_cast<const T*>(a.src[0]) + i * E, acc);
#pragma unroll
    for (int j = 1; j < 8; ++j) {
      if (j < a.n) {
        Vec16<T>::load(reinterpret_cast<const T*>(a.src[j]) + i * E, y);
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] += y[e];
      }
    }
    Vec16<T>::store(o + i * E, acc);
  }
}

template <typename T>
__global__ void dropout_vec_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t nvec, float p, uint64_t seed,
                                   uint64_t offset, const CaseStepState* state) {
  offset += rng_base_of(state);
  constexpr int E = Vec16<T>::N;
  const float scale = 1.f / (1.f - p);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    float v[E];
    Vec16<T>::load(x + i * E, v);
    const uint64_t idx0 = offset + (uint64_t)(i * E);
    if ((idx0 & 1) == 0) {  // one hash per element pair
#pragma unroll
      for (int e = 0; e < E; e += 2) {
        float u0, u1;
        rng_uniform2(seed, idx0 + e, u0, u1);
        v[e] = u0 >= p ? v[e] * scale : 0.f;
        v[e + 1] = u1 >= p ? v[e + 1] * scale : 0.f;
      }
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = rng_uniform(seed, idx0 + e) >= p ? v[e] * scale : 0.f;
    }
    Vec16<T>::store(y + i * E, v);
  }
}

template <typename T>
__global__ void mask_rows_vec_kernel(const T* __restrict__ x, const uint8_t* __restrict__ valid, T* __restrict__ y,
                                     int64_t rows, int64_t cols) {
  constexpr int E = Vec16<T>::N;
  const int64_t vpr = cols / E, nvec = rows * vpr;
  const bool in_place = x == y;  // y aliases x: a valid row is neither read nor written -- the pass costs one flag byte per 16-byte vector
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    const bool ok = valid[i / vpr] != 0;
    if (in_place && ok) continue;
    float v[E];
    if (ok) Vec16<T>::load(x + i * E, v);
    else {
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = 0.f;
    }
    Vec16<T>::store(y + i * E, v);
  }
}

// ---- single-output linear (Linear(H, 1): passage / token scorers, Interaction rank-1 terms) ---------------------------
// y[r] = x[r, :] . w (+ b): one wave per row, 16-byte loads, shuffle reduction -- an N = 1 GEMM tile would waste 127/128
// of the MFMA work and cannot use vector loads for its gradient operand.
template <typename T>
__global__ __launch_bounds__(256) void rowdot_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ b, float* __restrict__ y, int64_t rows,
                                                         int64_t cols) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  for (int64_t r = wave; r < rows; r += nwaves) {
    float acc = 0.f;
    for (int64_t c = lane; c < cols; c += 64) acc += Elem<T>::ld(x + r * cols + c) * w[c];
    acc = wave_sum(acc);
    if (lane == 0) y[r] = acc + (b ? b[0] : 0.f);
  }
}

// ---- few-output linear over a column-wise CONCATENATION of up to four inputs (the greedy step's mixing logits: Linear(3H, 1 + nmem) on
// [dec_out | ctx_q | ctx_p], CaSE/Model.py:116, Masque/Model.py:42): y[r, o] = sum_k x_k[r, :] . w[o, off_k : off_k + width_k] + b[o].
// One wave per row, NOUT <= 8 accumulators per lane, f32 weights; neither the concatenated row nor an N = 3 GEMM tile exists.
struct SkinnyArgs {
  const void* x[4];
  int64_t width[4];
  const float* w; const float* b; float* y;
  int64_t rows, cols;
  int nseg, nout;
};
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void linear_skinny_kernel(const SkinnyArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  constexpr int NV = Vec16<T>::N;  // elements per 16-byte load: 4 (f32) or 8 (bf16)
  for (int64_t r = wave; r < a.rows; r += nwaves) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int64_t off = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < a.nseg) {
        const T* xr = reinterpret_cast<const T*>(a.x[k]) + r * a.width[k];
        if constexpr (VEC) {  // every width a multiple of NV, every base 16-byte aligned: one 16-byte load of x and NV / 4 of each weight row per step
#pragma unroll 2
          for (int64_t c = (int64_t)lane * NV; c < a.width[k]; c += 64 * NV) {
            float xv[NV];
            Vec16<T>::load(xr + c, xv);
#pragma unroll
            for (int o = 0; o < 8; ++o) {
              if (o < a.nout) {
                const float* wr = a.w + o * a.cols + off + c;
#pragma unroll
                for (int q = 0; q < NV / 4; ++q) {
                  const float4 wv = *reinterpret_cast<const float4*>(wr + 4 * q);
                  acc[o] = fmaf(xv[4 * q], wv.x, acc[o]);
                  acc[o] = fmaf(xv[4 * q + 1], wv.y, acc[o]);
                  acc[o] = fmaf(xv[4 * q + 2], wv.z, acc[o]);
                  acc[o] = fmaf(xv[4 * q + 3], wv.w, acc[o]);
                }
              }
            }
          }
        } else {
          for (int64_t c = lane; c < a.width[k]; c += 64) {
            const float xv = Elem<T>::ld(xr + c);
#pragma unroll
            for (int o = 0; o < 8; ++o)
              if (o < a.nout) acc[o] = fmaf(xv, a.w[o * a.cols + off + c], acc[o]);
          }
        }
        off += a.width[k];
      }
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      if (o < a.nout) {
        const float t = wave_sum(acc[o]);
        if (lane == 0) a.y[r * a.nout + o] = t + (a.b ? a.b[o] : 0.f);
      }
    }
  }
}

// dx[r, :] = g[r] * w ; dw[c] += sum_r g[r] * x[r, c] ; db += sum_r g[r]   (column tiling as colsum)
template <typename T>
__global__ __launch_bounds__(256) void rowdot_bwd_kernel(const float* __restrict__ g, const T* __restrict__ x,
                                                         const float* __restrict__ w, T* __restrict__ dx,
                                                         float* __restrict__ dw, float* __restrict__ db, int64_t rows,
                                                         int64_t cols, int row_splits) {
  __shared__ float part[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int64_t col = (int64_t)(blockIdx.x / row_splits) * 64 + cx;
  const int split = blockIdx.x % row_splits;
  float s = 0.f, sb = 0.f;
  if (col < cols) {
    const float wc = w[col];
    for (int64_t r = (int64_t)split * 4 + ry; r < rows; r += (int64_t)row_splits * 4) {
      const float gr = g[r];
      s += gr * Elem<T>::ld(x + r * cols + col);
      sb += gr;
      if (dx) Elem<T>::st(dx + r * cols + col, gr * wc);
    }
  }
  part[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && col < cols) atomicAdd(dw + col, part[0][cx] + part[1][cx] + part[2][cx] + part[3][cx]);
  if (db && col == 0) {
    __syncthreads();
    part[ry][0] = sb;
    __syncthreads();
    if (ry == 0 && cx == 0) atomicAdd(db, part[0][0] + part[1][0] + part[2][0] + part[3][0]);
  }
}

static inline bool al16(const void* p) { return ((uintptr_t)p % 16) == 0; }

template <typename TS, typename TD>
__global__ void cast_kernel(const TS* __restrict__ x, TD* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    Elem<TD>::st(y + i, Elem<TS>::ld(x + i));
}

template <typename T>
__global__ void scale_cols_kernel(const T* __restrict__ x, const float* __restrict__ w, T* __restrict__ y, int64_t rows,
                                  int64_t cols) {
  const int64_t n = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    Elem<T>::st(y + i, Elem<T>::ld(x + i) * w[i % cols]);
}

// dx = dy * w ; dw[c] += sum_r dy * x   (same tiling as colsum)
template <typename T>
__global__ __launch_bounds__(256) void scale_cols_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                             const float* __restrict__ w, T* __restrict__ dx,
                                                             float* __restrict__ dw, int64_t rows, int64_t cols,
                                                             int row_splits) {
  __shared__ float part[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int64_t col = (int64_t)(blockIdx.x / row_splits) * 64 + cx;
  const int split = blockIdx.x % row_splits;
  float s = 0.f;
  if (col < cols) {
    const float wc = w[col];
    for (int64_t r = (int64_t)split * 4 + ry; r < rows; r += (int64_t)row_splits * 4) {
      const float g = Elem<T>::ld(dy + r * cols + col);
      s += g * Elem<T>::ld(x + r * cols + col);
      Elem<T>::st(dx + r * cols + col, g * wc);
    }
  }
  part[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && col < cols) atomicAdd(dw + col, part[0][cx] + part[1][cx] + part[2][cx] + part[3][cx]);
}

// y[r, :] = x[r, :] * scale + (pe ? pe[r % seq_len, :] : 0)   (stand-alone PositionalEmbedding and its backward)
template <typename T>
__global__ void scale_add_rows_kernel(const T* __restrict__ x, const float* __restrict__ pe, T* __restrict__ y,
                                      int64_t rows, int64_t seq_len, int64_t H, float scale) {
  const int64_t n = rows * H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / H, c = i % H;
    Elem<T>::st(y + i, Elem<T>::ld(x + i) * scale + (pe ? pe[(r % seq_len) * H + c] : 0.f));
  }
}

// ---- K1 ----------------------------------------------------------------------------------------
template <typename T>
__global__ void embed_pos_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table,
                                     const float* __restrict__ pe, T* __restrict__ out, int64_t rows, int64_t seq_len,
                                     int64_t H, int64_t vocab, float scale, float drop_p, uint64_t seed, uint64_t offset,
                                     const CaseStepState* state) {
  offset += rng_base_of(state);
  const int64_t n = rows * H;
  const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / H, c = i % H;
    int64_t id = ids[r];
    if (id < 0 || id >= vocab) id = 0;  // out-of-range ids read the padding row instead of faulting
    float v = table[id * H + c] * scale + pe[(r % seq_len) * H + c];
    if (drop_p > 0.f) v = rng_uniform(seed, offset + (uint64_t)i) >= drop_p ? v * ks : 0.f;
    Elem<T>::st(out + i, v);
  }
}

// vector form (H % 8 == 0, 16-byte aligned table / pe / out): a thread owns 8 consecutive features of one row -- one id load,
// two 16-byte table loads, two of the position table, one 16-byte (bf16) or two (f32) stores; 32-bit index arithmetic.  The
// scalar kernel above spends a 64-bit divide and modulo per element and ran at 2.5 TB/s.
template <typename T>
__global__ __launch_bounds__(256) void embed_pos_fwd_vec_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table,
                                                                const float* __restrict__ pe, T* __restrict__ out, unsigned chunks,
                                                                unsigned seq_len, unsigned H8, int64_t vocab, float scale, float drop_p,
                                                                uint64_t seed, uint64_t offset, const CaseStepState* state) {
  offset += rng_base_of(state);
  const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  for (unsigned ci = blockIdx.x * 256u + threadIdx.x; ci < chunks; ci += gridDim.x * 256u) {
    const unsigned r = ci / H8, c = (ci - r * H8) * 8u;
    int64_t id = ids[r];
    if (id < 0 || id >= vocab) id = 0;
    const float* tp = table + id * (int64_t)(H8 * 8u) + c;
    const float* pp = pe + (int64_t)(r % seq_len) * (H8 * 8u) + c;
    const float4 t0 = *reinterpret_cast<const float4*>(tp), t1 = *reinterpret_cast<const float4*>(tp + 4);
    const float4 p0 = *reinterpret_cast<const float4*>(pp), p1 = *reinterpret_cast<const float4*>(pp + 4);
    float v[8] = {fmaf(t0.x, scale, p0.x), fmaf(t0.y, scale, p0.y), fmaf(t0.z, scale, p0.z), fmaf(t0.w, scale, p0.w),
                  fmaf(t1.x, scale, p1.x), fmaf(t1.y, scale, p1.y), fmaf(t1.z, scale, p1.z), fmaf(t1.w, scale, p1.w)};
    if (drop_p > 0.f) dropout8(v, seed, offset + (uint64_t)r * (H8 * 8u) + c, drop_p, ks);
    T* op = out + (int64_t)r * (H8 * 8u) + c;
    if constexpr (sizeof(T) == 2) {
      uint4 w;
      w.x = f32x2_to_bf16x2(v[0], v[1]); w.y = f32x2_to_bf16x2(v[2], v[3]);
      w.z = f32x2_to_bf16x2(v[4], v[5]); w.w = f32x2_to_bf16x2(v[6], v[7]);
      *reinterpret_cast<uint4*>(op) = w;
    } else {
      *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
  }
}

template <typename T>
__global__ void embed_pos_bwd_kernel(const int64_t* __restrict__ ids, const T* __restrict__ d_out,
                                     float* __restrict__ d_table, int64_t rows, int64_t H, int64_t vocab, float scale,
                                     float drop_p, uint64_t seed, uint64_t offset, const CaseStepState* state) {
  offset += rng_base_of(state);
  const int64_t n = rows * H;
  const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / H, c = i % H;
    const int64_t id = ids[r];
    if (id <= 0 || id >= vocab) continue;  // padding_idx = 0 gets no gradient
    float g = Elem<T>::ld(d_out + i);
    if (drop_p > 0.f) g = rng_uniform(seed, offset + (uint64_t)i) >= drop_p ? g * ks : 0.f;
    atomicAdd(d_table + id * H + c, g * scale);
  }
}

// ---- K9 ----------------------------------------------------------------------------------------
template <typename T>
__global__ void masked_mean_fwd_kernel(const T* __restrict__ x, const uint8_t* __restrict__ valid, T* __restrict__ out,
                                       int64_t n, int64_t L, int64_t H) {
  const int64_t total = n * H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = i / H, c = i % H;
    float acc = 0.f, cnt = 0.f;
    for (int64_t l = 0; l < L; ++l)
      if (valid[s * L + l]) {
        acc += Elem<T>::ld(x + (s * L + l) * H + c);
        cnt += 1.f;
      }
    Elem<T>::st(out + i, acc / cnt);
  }
}

template <typename T>
__global__ void masked_mean_bwd_kernel(const T* __restrict__ d_out, const uint8_t* __restrict__ valid,
                                       T* __restrict__ dx, int64_t n, int64_t L, int64_t H) {
  const int64_t total = n * L * H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = i % H, sl = i / H, s = sl / L;
    float cnt = 0.f;
    for (int64_t l = 0; l < L; ++l) cnt += valid[s * L + l] ? 1.f : 0.f;
    Elem<T>::st(dx + i, valid[sl] ? Elem<T>::ld(d_out + s * H + c) / cnt : 0.f);
  }
}

// Vector forms: one 256-thread workgroup per (sequence, block of 64 x 16-byte channel chunks).  The four waves take
// interleaved rows (sequence positions), each lane sums its 16-byte chunk over them, LDS combines the four partial sums.
// (The scalar kernel above walks the L rows serially from one thread per channel: 611 us for 320 x 384 x 512 bf16.)
template <typename T>
__global__ __launch_bounds__(256) void masked_mean_fwd_vec_kernel(const T* __restrict__ x, const uint8_t* __restrict__ valid,
                                                                  T* __restrict__ out, int64_t L, int64_t H) {
  constexpr int E = Vec16<T>::N;
  __shared__ float part[4][64][E + 1];
  __shared__ float cnts[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t s = blockIdx.x, c0 = ((int64_t)blockIdx.y * 64 + lane) * E;
  const bool in = c0 < H;
  float acc[E];
#pragma unroll
  for (int e = 0; e < E; ++e) acc[e] = 0.f;
  float cnt = 0.f;
  for (int64_t l = w; l < L; l += 4) {
    if (!valid[s * L + l]) continue;  // wave-uniform
    cnt += 1.f;
    if (in) {
      float v[E];
      Vec16<T>::load(x + (s * L + l) * H + c0, v);
#pragma unroll
      for (int e = 0; e < E; ++e) acc[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < E; ++e) part[w][lane][e] = acc[e];
  if (lane == 0) cnts[w] = cnt;
  __syncthreads();
  if (w == 0 && in) {
    const float n = cnts[0] + cnts[1] + cnts[2] + cnts[3];
    float v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = (part[0][lane][e] + part[1][lane][e] + part[2][lane][e] + part[3][lane][e]) / n;
    Vec16<T>::store(out + s * H + c0, v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void masked_mean_bwd_vec_kernel(const T* __restrict__ d_out, const uint8_t* __restrict__ valid,
                                                                  T* __restrict__ dx, int64_t L, int64_t H) {
  constexpr int E = Vec16<T>::N;
  __shared__ float red[32];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t s = blockIdx.x, c0 = ((int64_t)blockIdx.y * 64 + lane) * E;
  float cnt = 0.f;
  for (int64_t l = threadIdx.x; l < L; l += 256) cnt += valid[s * L + l] ? 1.f : 0.f;
  cnt = block_sum(cnt, red);
  if (c0 >= H) return;
  float g[E], z[E];
  Vec16<T>::load(d_out + s * H + c0, g);
#pragma unroll
  for (int e = 0; e < E; ++e) { g[e] /= cnt; z[e] = 0.f; }
  for (int64_t l = w; l < L; l += 4) Vec16<T>::store(dx + (s * L + l) * H + c0, valid[s * L + l] ? g : z);
}

// ---- K14 epilogue ------------------------------------------------------------------------------
template <typename T>
__global__ void highway_fwd_kernel(const T* __restrict__ gnl, T* __restrict__ y, int64_t rows, int64_t cols) {
  const int64_t n = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols, c = i % cols;
    const T* p = gnl + r * 3 * cols + c;
    const float t = 1.f / (1.f + expf(-Elem<T>::ld(p)));
    Elem<T>::st(y + i, t * tanhf(Elem<T>::ld(p + cols)) + (1.f - t) * Elem<T>::ld(p + 2 * cols));
  }
}

template <typename T>
__global__ void highway_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ gnl, T* __restrict__ d_gnl,
                                   int64_t rows, int64_t cols) {
  const int64_t n = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols, c = i % cols;
    const T* p = gnl + r * 3 * cols + c;
    T* q = d_gnl + r * 3 * cols + c;
    const float t = 1.f / (1.f + expf(-Elem<T>::ld(p)));
    const float f = tanhf(Elem<T>::ld(p + cols)), lin = Elem<T>::ld(p + 2 * cols), g = Elem<T>::ld(dy + i);
    Elem<T>::st(q, g * (f - lin) * t * (1.f - t));
    Elem<T>::st(q + cols, g * t * (1.f - f * f));
    Elem<T>::st(q + 2 * cols, g * (1.f - t));
  }
}

// ---- K8 feature assembly -----------------------------------------------------------------------
template <typename T>
__global__ void concat5_fwd_kernel(const T* __restrict__ e, const T* __restrict__ a1, const T* __restrict__ a2,
                                   const uint8_t* __restrict__ valid, T* __restrict__ out, int64_t rows, int64_t H) {
  const int64_t n = rows * H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / H, c = i % H;
    const bool ok = valid[r];
    const float ev = ok ? Elem<T>::ld(e + i) : 0.f, x1 = ok ? Elem<T>::ld(a1 + i) : 0.f, x2 = ok ? Elem<T>::ld(a2 + i) : 0.f;
    T* o = out + r * 5 * H + c;
    Elem<T>::st(o, ev);
    Elem<T>::st(o + H, x1);
    Elem<T>::st(o + 2 * H, x2);
    Elem<T>::st(o + 3 * H, ev * x1);
    Elem<T>::st(o + 4 * H, ev * x2);
  }
}

template <typename T>
__global__ void concat5_bwd_kernel(const T* __restrict__ d_out, const T* __restrict__ e, const T* __restrict__ a1,
                                   const T* __restrict__ a2, const uint8_t* __restrict__ valid, T* __restrict__ de,
                                   T* __restrict__ da1, T* __restrict__ da2, int64_t rows, int64_t H) {
  const int64_t n = rows * H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / H, c = i % H;
    float ge = 0.f, g1 = 0.f, g2 = 0.f;
    if (valid[r]) {
      const T* g = d_out + r * 5 * H + c;
      const float ev = Elem<T>::ld(e + i), x1 = Elem<T>::ld(a1 + i), x2 = Elem<T>::ld(a2 + i);
      const float g3 = Elem<T>::ld(g + 3 * H), g4 = Elem<T>::ld(g + 4 * H);
      ge = Elem<T>::ld(g) + g3 * x1 + g4 * x2;
      g1 = Elem<T>::ld(g + H) + g3 * ev;
      g2 = Elem<T>::ld(g + 2 * H) + g4 * ev;
    }
    Elem<T>::st(de + i, ge);
    Elem<T>::st(da1 + i, g1);
    Elem<T>::st(da2 + i, g2);
  }
}

// 16-byte forms of the two kernels above (H a multiple of the vector width, aligned tensors): the scalar forms move 2 bytes per lane and
// instruction and ran at 2.5 TB/s on the [122 880, 5 x 512] concatenation
template <typename T>
__global__ void concat5_fwd_vec_kernel(const T* __restrict__ e, const T* __restrict__ a1, const T* __restrict__ a2,
                                       const uint8_t* __restrict__ valid, T* __restrict__ out, int64_t rows, int64_t H) {
  constexpr int E = Vec16<T>::N;
  const int64_t hv = H / E, n = rows * hv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / hv, c = (i - r * hv) * E;
    float ev[E], x1[E], x2[E], p1[E], p2[E];
    Vec16<T>::load(e + r * H + c, ev);
    Vec16<T>::load(a1 + r * H + c, x1);
    Vec16<T>::load(a2 + r * H + c, x2);
    const bool ok = valid[r];
#pragma unroll
    for (int k = 0; k < E; ++k) {
      if (!ok) ev[k] = x1[k] = x2[k] = 0.f;
      p1[k] = ev[k] * x1[k];
      p2[k] = ev[k] * x2[k];
    }
    T* o = out + r * 5 * H + c;
    Vec16<T>::store(o, ev);
    Vec16<T>::store(o + H, x1);
    Vec16<T>::store(o + 2 * H, x2);
    Vec16<T>::store(o + 3 * H, p1);
    Vec16<T>::store(o + 4 * H, p2);
  }
}

template <typename T>
__global__ void concat5_bwd_vec_kernel(const T* __restrict__ d_out, const T* __restrict__ e, const T* __restrict__ a1,
                                       const T* __restrict__ a2, const uint8_t* __restrict__ valid, T* __restrict__ de,
                                       T* __restrict__ da1, T* __restrict__ da2, int64_t rows, int64_t H) {
  constexpr int E = Vec16<T>::N;
  const int64_t hv = H / E, n = rows * hv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / hv, c = (i - r * hv) * E;
    float ge[E], g1[E], g2[E];
    if (valid[r]) {
      const T* g = d_out + r * 5 * H + c;
      float ev[E], x1[E], x2[E], g3[E], g4[E];
      Vec16<T>::load(e + r * H + c, ev);
      Vec16<T>::load(a1 + r * H + c, x1);
      Vec16<T>::load(a2 + r * H + c, x2);
      Vec16<T>::load(g, ge);
      Vec16<T>::load(g + H, g1);
      Vec16<T>::load(g + 2 * H, g2);
      Vec16<T>::load(g + 3 * H, g3);
      Vec16<T>::load(g + 4 * H, g4);
#pragma unroll
      for (int k = 0; k < E; ++k) {
        ge[k] = ge[k] + g3[k] * x1[k] + g4[k] * x2[k];
        g1[k] = g1[k] + g3[k] * ev[k];
        g2[k] = g2[k] + g4[k] * ev[k];
      }
    } else {
#pragma unroll
      for (int k = 0; k < E; ++k) ge[k] = g1[k] = g2[k] = 0.f;
    }
    Vec16<T>::store(de + r * H + c, ge);
    Vec16<T>::store(da1 + r * H + c, g1);
    Vec16<T>::store(da2 + r * H + c, g2);
  }
}

template <typename T>
__global__ void max_over_p_fwd_kernel(const T* __restrict__ x, T* __restrict__ out, int32_t* __restrict__ arg, int64_t B,
                                      int64_t P, int64_t inner) {
  const int64_t n = B * inner;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / inner, k = i % inner;
    float best = Elem<T>::ld(x + (b * P) * inner + k);
    int32_t bi = 0;
    for (int64_t p = 1; p < P; ++p) {
      const float v = Elem<T>::ld(x + (b * P + p) * inner + k);
      if (v > best) {  // first maximum wins, as torch.max
        best = v;
        bi = (int32_t)p;
      }
    }
    Elem<T>::st(out + i, best);
    arg[i] = bi;
  }
}

template <typename T>
__global__ void max_over_p_bwd_kernel(const T* __restrict__ d_out, const int32_t* __restrict__ arg, T* __restrict__ dx,
                                      int64_t B, int64_t P, int64_t inner) {
  const int64_t n = B * P * inner;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = i % inner, bp = i / inner, b = bp / P, p = bp % P;
    Elem<T>::st(dx + i, arg[b * inner + k] == (int32_t)p ? Elem<T>::ld(d_out + b * inner + k) : 0.f);
  }
}

}  // namespace

#define EW_DISPATCH(what, n_elems, KERNEL, ...)                                                              \
  do {                                                                                                       \
    const int grid = grid_for((n_elems), EW_THREADS, 4);                                                     \
    hipStream_t s_ = (hipStream_t)stream;                                                                    \
    if (dtype == CASE_F32) { typedef float T; hipLaunchKernelGGL(KERNEL<T>, dim3(grid), dim3(EW_THREADS), 0, s_, __VA_ARGS__); } \
    else if (dtype == CASE_BF16) { typedef bf16_t T; hipLaunchKernelGGL(KERNEL<T>, dim3(grid), dim3(EW_THREADS), 0, s_, __VA_ARGS__); } \
    else return case_set_error(CASE_E_UNSUPPORTED, what ": dtype %d", dtype);                              \
    return case_check_launch(what);                                                                          \
  } while (0)

extern "C" int case_add(const void* a, const void* b, void* out, int64_t n, int32_t dtype, case_stream_t stream) {
  CASE_REQUIRE(a && b && out && n > 0, "case_add: bad argument");
  const int ev = dtype == CASE_BF16 ? 8 : 4;
  if (n % ev == 0 && al16(a) && al16(b) && al16(out))
    EW_DISPATCH("case_add", n / ev, add_vec_kernel, (const T*)a, (const T*)b, (T*)out, n / ev);
  EW_DISPATCH("case_add", n, add_kernel, (const T*)a, (const T*)b, (T*)out, n);
}

extern "C" int case_add_n(const void* const* srcs, int32_t count, void* out, int64_t n, int32_t dtype, case_stream_t stream) {
  CASE_REQUIRE(srcs && out && n > 0 && count >= 2 && count <= 8, "case_add_n: bad argument (2 <= count <= 8)");
  const int ev = dtype == CASE_BF16 ? 8 : 4;
  CASE_REQUIRE(n % ev == 0 && al16(out), "case_add_n: element count must be a multiple of %d and the tensors 16-byte aligned", ev);
  AddNArgs a = {};
  a.n = count;
  for (int j = 0; j < count; ++j) {
    CASE_REQUIRE(srcs[j] && al16(srcs[j]), "case_add_n: null or misaligned source %d", j);
    a.src[j] = srcs[j];
  }
  EW_DISPATCH("case_add_n", n / ev, add_n_vec_kernel, a, (T*)out, n / ev);
}

extern "C" int case_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, uint64_t offset, const CaseStepState* state,
                            int32_t dtype, case_stream_t stream) {
  CASE_REQUIRE(x && y && n > 0 && p >= 0.f && p < 1.f, "case_dropout: bad argument");
  const int ev = dtype == CASE_BF16 ? 8 : 4;
  if (n % ev == 0 && al16(x) && al16(y))
    EW_DISPATCH("case_dropout", n / ev, dropout_vec_kernel, (const T*)x, (T*)y, n / ev, p, seed, offset, state);
  EW_DISPATCH("case_dropout", n, dropout_kernel, (const T*)x, (T*)y, n, p, seed, offset, state);
}

extern "C" int case_mask_rows(const void* x, const uint8_t* row_valid, void* y, int64_t rows, int64_t cols,
                              int32_t dtype, case_stream_t stream) {
  CASE_REQUIRE(x && row_valid && y && rows > 0 && cols > 0, "case_mask_rows: bad argument");
  const int ev = dtype == CASE_BF16 ? 8 : 4;
  if (cols % ev == 0 && al16(x) && al16(y))
    EW_DISPATCH("case_mask_rows", rows * cols / ev, mask_rows_vec_kernel, (const T*)x, row_valid, (T*)y, rows, cols);
  EW_DISPATCH("case_mask_rows", rows * cols, mask_rows_kernel, (const T*)x, row_valid, (T*)y, rows, cols);
}

static int colsum_splits(int64_t rows, int64_t cols) {
  const int64_t col_blocks = (cols + 63) / 64;
  int64_t want = 2048 / col_blocks;  // aim at ~2048 workgroups
  const int64_t cap = (rows + 31) / 32;
  if (want > cap) want = cap;
  return (int)(want < 1 ? 1 : want);
}

extern "C" int case_colsum(const void* x, float* out, int64_t rows, int64_t cols, int32_t dtype, case_stream_t stream) {
  CASE_REQUIRE(x && out && rows > 0 && cols > 0, "case_colsum: bad argument");
  {
    const int ev = dtype == CASE_BF16 ? 8 : 4;
    if (cols % ev == 0 && al16(x)) {
      const int64_t col_blocks = (cols + 32 * ev - 1) / (32 * ev);
      int64_t want = 2048 / col_blocks, cap = (rows + 63) / 64;
      if (want > cap) want = cap;
      if (want < 1) want = 1;
      const int vsplits = (int)want;
      hipStream_t vs = (hipStream_t)stream;
      if (dtype == CASE_F32) hipLaunchKernelGGL(colsum_vec_kernel<float>, dim3((unsigned)(col_blocks * vsplits)), dim3(256), 0, vs, (const float*)x, out, rows, cols, vsplits);
      else hipLaunchKernelGGL(colsum_vec_kernel<bf16_t>, dim3((unsigned)(col_blocks * vsplits)), dim3(256), 0, vs, (const bf16_t*)x, out, rows, cols, vsplits);
      return case_check_launch("case_colsum");
    }
  }
  const int splits = colsum_splits(rows, cols);
  const int grid = (int)((cols + 63) / 64) * splits;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CASE_F32) hipLaunchKernelGGL(colsum_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, out, rows, cols, splits);
  else hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, out, rows, cols, splits);
  return case_check_launch("case_colsum");
}

extern "C" int case_cast(const void* x, void* y, int64_t n, int32_t src, int32_t dst, case_stream_t stream) {
  CASE_REQUIRE(x && y && n > 0, "case_cast: bad argument");
  const int grid = grid_for(n, EW_THREADS, 4);
  hipStream_t s = (hipStream_t)stream;
  if (src == CASE_F32 && dst == CASE_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(grid), dim3(EW_THREADS), 0, s, (const float*)x, (bf16_t*)y, n);
  else if (src == CASE_BF16 && dst == CASE_F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(grid), dim3(EW_THREADS), 0, s, (const bf16_t*)x, (float*)y, n);
  else if (src == CASE_F32 && dst == CASE_F32) hipLaunchKernelGGL((cast_kernel<float, float>), dim3(grid), dim3(EW_THREADS), 0, s, (const float*)x, (float*)y, n);
  else if (src == CASE_BF16 && dst == CASE_BF16) hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), dim3(grid), dim3(EW_THREADS), 0, s, (const bf16_t*)x, (bf16_t*)y, n);
  else return case_set_error(CASE_E_UNSUPPORTED, "case_cast: dtype");
  return case_check_launch("case_cast");
}

extern "C" int case_scale_cols(const void* x, const float* w, void* y, int64_t rows, int64_t cols, int32_t dtype,
                               case_stream_t stream) {
  CASE_REQUIRE(x && w && y && rows > 0 && cols > 0, "case_scale_cols: bad argument");
  EW_DISPATCH("case_scale_cols", rows * cols, scale_cols_kernel, (const T*)x, w, (T*)y, rows, cols);
}

extern "C" int case_scale_cols_bwd(const void* dy, const void* x, const float* w, void* dx, float* dw, int64_t rows,
                                   int64_t cols, int32_t dtype, case_stream_t stream) {
  CASE_REQUIRE(dy && x && w && dx && dw && rows > 0 && cols > 0, "case_scale_cols_bwd: bad argument");
  // every row must be visited exactly once for dx: one split per 4-row lane group is required
  const int splits = colsum_splits(rows, cols);
  const int grid = (int)((cols + 63) / 64) * splits;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CASE_F32) hipLaunchKernelGGL(scale_cols_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)dy, (const float*)x, w, (float*)dx, dw, rows, cols, splits);
  else hipLaunchKernelGGL(scale_cols_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, w, (bf16_t*)dx, dw, rows, cols, splits);
  return case_check_launch("case_scale_cols_bwd");
}

extern "C" int case_scale_add_rows(const void* x, const float* pe, void* y, int64_t rows, int64_t seq_len, int64_t H,
                                   float scale, int32_t dtype, case_stream_t stream) {
  CASE_REQUIRE(x && y && rows > 0 && seq_len > 0 && H > 0, "case_scale_add_rows: bad argument");
  EW_DISPATCH("case_scale_add_rows", rows * H, scale_add_rows_kernel, (const T*)x, pe, (T*)y, rows, seq_len, H, scale);
}

extern "C" int case_embed_pos_fwd(const int64_t* ids, const float* table, const float* pe, void* out, int64_t rows,
                                  int64_t seq_len, int64_t H, int64_t vocab, float scale, float drop_p, uint64_t seed,
                                  uint64_t offset, const CaseStepState* state, int32_t dtype, case_stream_t stream) {
  CASE_REQUIRE(ids && table && pe && out && rows > 0 && seq_len > 0 && H > 0 && vocab > 0, "case_embed_pos_fwd: bad argument");
  if ((dtype == CASE_BF16 || dtype == CASE_F32) && H % 8 == 0 && al16(table) && al16(pe) && al16(out) && rows * (H / 8) < (1ll << 31) &&
      seq_len < (1ll << 31)) {
    const unsigned chunks = (unsigned)(rows * (H / 8));
    const dim3 grid((unsigned)grid_for(chunks, 256, 1, 256 * 16));
    hipStream_t s_ = (hipStream_t)stream;
    if (dtype == CASE_F32)
      hipLaunchKernelGGL(embed_pos_fwd_vec_kernel<float>, grid, dim3(256), 0, s_, ids, table, pe, (float*)out, chunks, (unsigned)seq_len,
                         (unsigned)(H / 8), vocab, scale, drop_p, seed, offset, state);
    else
      hipLaunchKernelGGL(embed_pos_fwd_vec_kernel<bf16_t>, grid, dim3(256), 0, s_, ids, table, pe, (bf16_t*)out, chunks, (unsigned)seq_len,
                         (unsigned)(H / 8), vocab, scale, drop_p, seed, offset, state);
    return case_check_launch("case_embed_pos_fwd");
  }
  EW_DISPATCH("case_embed_pos_fwd", rows * H, embed_pos_fwd_kernel, ids, table, pe, (T*)out, rows, seq_len, H, vocab, scale,
              drop_p, seed, offset, state);
}

extern "C" int case_embed_pos_bwd(const int64_t* ids, const void* d_out, float* d_table, int64_t rows, int64_t H,
                                  int64_t vocab, float scale, float drop_p, uint64_t seed, uint64_t offset, const CaseStepState* state,
                                  int32_t dtype, case_stream_t stream) {
  CASE_REQUIRE(ids && d_out && d_table && rows > 0 && H > 0 && vocab > 0, "case_embed_pos_bwd: bad argument");
  EW_DISPATCH("case_embed_pos_bwd", rows * H, embed_pos_bwd_kernel, ids, (const T*)d_out, d_table, rows, H, vocab, scale,
              drop_p, seed, offset, state);
}

extern "C" int case_masked_mean_fwd(const void* x, const uint8_t* valid, void* out, int64_t n, int64_t L, int64_t H,
                                    int32_t dtype, case_stream_t stream) {
  CASE_REQUIRE(x && valid && out && n > 0 && L > 0 && H > 0, "case_masked_mean_fwd: bad argument");
  {
    const int ev = dtype == CASE_BF16 ? 8 : 4;
    if ((dtype == CASE_BF16 || dtype == CASE_F32) && H % ev == 0 && al16(x) && al16(out) && n < (1ll << 31)) {
      const dim3 grid((unsigned)n, (unsigned)((H / ev + 63) / 64));
      hipStream_t s_ = (hipStream_t)stream;
      if (dtype == CASE_F32) hipLaunchKernelGGL(masked_mean_fwd_vec_kernel<float>, grid, dim3(256), 0, s_, (const float*)x, valid, (float*)out, L, H);
      else hipLaunchKernelGGL(masked_mean_fwd_vec_kernel<bf16_t>, grid, dim3(256), 0, s_, (const bf16_t*)x, valid, (bf16_t*)out, L, H);
      return case_check_launch("case_masked_mean_fwd");
    }
  }
  EW_DISPATCH("case_masked_mean_fwd", n * H, masked_mean_fwd_kernel, (const T*)x, valid, (T*)out, n, L, H);
}

extern "C" int case_masked_mean_bwd(const void* d_out, const uint8_t* valid, void* dx, int64_t n, int64_t L, int64_t H,
                                    int32_t dtype, case_stream_t stream) {
  CASE_REQUIRE(d_out && valid && dx && n > 0 && L > 0 && H > 0, "case_masked_mean_bwd: bad argument");
  {
    const int ev = dtype == CASE_BF16 ? 8 : 4;
    if ((dtype == CASE_BF16 || dtype == CASE_F32) && H % ev == 0 && al16(d_out) && al16(dx) && n < (1ll << 31)) {
      const dim3 grid((unsigned)n, (unsigned)((H / ev + 63) / 64));
      hipStream_t s_ = (hipStream_t)stream;
      if (dtype == CASE_F32) hipLaunchKernelGGL(masked_mean_bwd_vec_kernel<float>, grid, dim3(256), 0, s_, (const float*)d_out, valid, (float*)dx, L, H);
      else hipLaunchKernelGGL(masked_mean_bwd_vec_kernel<bf16_t>, grid, dim3(256), 0, s_, (const bf16_t*)d_out, valid, (bf16_t*)dx, L, H);
      return case_check_launch("case_masked_mean_bwd");
    }
  }
  EW_DISPATCH("case_masked_mean_bwd", n * L * H, masked_mean_bwd_kernel, (const T*)d_out, valid, (T*)dx, n, L, H);
}

extern "C" int case_highway_gate_fwd(const void* gnl, void* y, int64_t rows, int64_t cols, int32_t dtype,
                                     case_stream_t stream) {
  CASE_REQUIRE(gnl && y && rows > 0 && cols > 0, "case_highway_gate_fwd: bad argument");
  EW_DISPATCH("case_highway_gate_fwd", rows * cols, highway_fwd_kernel, (const T*)gnl, (T*)y, rows, cols);
}

extern "C" int case_highway_gate_bwd(const void* dy, const void* gnl, void* d_gnl, int64_t rows, int64_t cols,
                                     int32_t dtype, case_stream_t stream) {
  CASE_REQUIRE(dy && gnl && d_gnl && rows > 0 && cols > 0, "case_highway_gate_bwd: bad argument");
  EW_DISPATCH("case_highway_gate_bwd", rows * cols, highway_bwd_kernel, (const T*)dy, (const T*)gnl, (T*)d_gnl, rows, cols);
}

extern "C" int case_concat5_fwd(const void* e, const void* a1, const void* a2, const uint8_t* row_valid, void* out,
                                int64_t rows, int64_t H, int32_t dtype, case_stream_t stream) {
  CASE_REQUIRE(e && a1 && a2 && row_valid && out && rows > 0 && H > 0, "case_concat5_fwd: bad argument");
  if (H % (dtype == CASE_BF16 ? 8 : 4) == 0 && al16(e) && al16(a1) && al16(a2) && al16(out))
    EW_DISPATCH("case_concat5_fwd", rows * H / (dtype == CASE_BF16 ? 8 : 4), concat5_fwd_vec_kernel, (const T*)e, (const T*)a1, (const T*)a2, row_valid,
                (T*)out, rows, H);
  EW_DISPATCH("case_concat5_fwd", rows * H, concat5_fwd_kernel, (const T*)e, (const T*)a1, (const T*)a2, row_valid, (T*)out,
              rows, H);
}

extern "C" int case_concat5_bwd(const void* d_out, const void* e, const void* a1, const void* a2,
                                const uint8_t* row_valid, void* de, void* da1, void* da2, int64_t rows, int64_t H,
                                int32_t dtype, case_stream_t stream) {
  CASE_REQUIRE(d_out && e && a1 && a2 && row_valid && de && da1 && da2 && rows > 0 && H > 0, "case_concat5_bwd: bad argument");
  if (H % (dtype == CASE_BF16 ? 8 : 4) == 0 && al16(d_out) && al16(e) && al16(a1) && al16(a2) && al16(de) && al16(da1) && al16(da2))
    EW_DISPATCH("case_concat5_bwd", rows * H / (dtype == CASE_BF16 ? 8 : 4), concat5_bwd_vec_kernel, (const T*)d_out, (const T*)e, (const T*)a1,
                (const T*)a2, row_valid, (T*)de, (T*)da1, (T*)da2, rows, H);
  EW_DISPATCH("case_concat5_bwd", rows * H, concat5_bwd_kernel, (const T*)d_out, (const T*)e, (const T*)a1, (const T*)a2,
              row_valid, (T*)de, (T*)da1, (T*)da2, rows, H);
}

extern "C" int case_max_over_p_fwd(const void* x, void* out, int32_t* argmax, int64_t B, int64_t P, int64_t inner,
                                   int32_t dtype, case_stream_t stream) {
  CASE_REQUIRE(x && out && argmax && B > 0 && P > 0 && inner > 0, "case_max_over_p_fwd: bad argument");
  EW_DISPATCH("case_max_over_p_fwd", B * inner, max_over_p_fwd_kernel, (const T*)x, (T*)out, argmax, B, P, inner);
}

extern "C" int case_max_over_p_bwd(const void* d_out, const int32_t* argmax, void* dx, int64_t B, int64_t P,
                                   int64_t inner, int32_t dtype, case_stream_t stream) {
  CASE_REQUIRE(d_out && argmax && dx && B > 0 && P > 0 && inner > 0, "case_max_over_p_bwd: bad argument");
  EW_DISPATCH("case_max_over_p_bwd", B * P * inner, max_over_p_bwd_kernel, (const T*)d_out, argmax, (T*)dx, B, P, inner);
}

extern "C" int case_rowdot_fwd(const void* x, const float* w, const float* b, float* y, int64_t rows, int64_t cols,
                               int32_t dtype, case_stream_t stream) {
  CASE_REQUIRE(x && w && y && rows > 0 && cols > 0, "case_rowdot_fwd: bad argument");
  const int grid = grid_for(rows, 4, 2, 256 * 8);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CASE_F32) hipLaunchKernelGGL(rowdot_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, w, b, y, rows, cols);
  else if (dtype == CASE_BF16) hipLaunchKernelGGL(rowdot_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, w, b, y, rows, cols);
  else return case_set_error(CASE_E_UNSUPPORTED, "case_rowdot_fwd: dtype %d", dtype);
  return case_check_launch("case_rowdot_fwd");
}

extern "C" int case_linear_skinny(const void* const* xs, const int64_t* widths, int32_t nseg, const float* w, const float* b, float* y,
                                  int64_t rows, int32_t nout, int32_t dtype, case_stream_t stream) {
  CASE_REQUIRE(xs && widths && w && y && rows > 0 && nseg >= 1 && nseg <= 4 && nout >= 1 && nout <= 8, "case_linear_skinny: 1-4 inputs, 1-8 outputs");
  SkinnyArgs a;
  a.cols = 0;
  for (int k = 0; k < 4; ++k) {
    a.x[k] = k < nseg ? xs[k] : nullptr;
    a.width[k] = k < nseg ? widths[k] : 0;
    CASE_REQUIRE(k >= nseg || (xs[k] && widths[k] > 0), "case_linear_skinny: null or empty input %d", k);
    a.cols += a.width[k];
  }
  a.w = w; a.b = b; a.y = y;
  a.rows = rows; a.nseg = nseg; a.nout = nout;
  const int grid = grid_for(rows, 4, 1, 256 * 8);
  hipStream_t s = (hipStream_t)stream;
  const int nv = dtype == CASE_F32 ? 4 : 8;
  bool vec = ((uintptr_t)w % 16) == 0 && a.cols % 4 == 0;
  for (int k = 0; k < nseg; ++k) vec = vec && widths[k] % nv == 0 && ((uintptr_t)xs[k] % 16) == 0;
  if (dtype == CASE_F32) {
    if (vec) hipLaunchKernelGGL((linear_skinny_kernel<float, true>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((linear_skinny_kernel<float, false>), dim3(grid), dim3(256), 0, s, a);
  } else if (dtype == CASE_BF16) {
    if (vec) hipLaunchKernelGGL((linear_skinny_kernel<bf16_t, true>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((linear_skinny_kernel<bf16_t, false>), dim3(grid), dim3(256), 0, s, a);
  } else return case_set_error(CASE_E_UNSUPPORTED, "case_linear_skinny: dtype %d", dtype);
  return case_check_launch("case_linear_skinny");
}

extern "C" int case_rowdot_bwd(const float* g, const void* x, const float* w, void* dx, float* dw, float* db, int64_t rows,
                               int64_t cols, int32_t dtype, case_stream_t stream) {
  CASE_REQUIRE(g && x && w && dw && rows > 0 && cols > 0, "case_rowdot_bwd: bad argument");
  const int splits = colsum_splits(rows, cols);
  const int grid = (int)((cols + 63) / 64) * splits;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CASE_F32) hipLaunchKernelGGL(rowdot_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, g, (const float*)x, w, (float*)dx, dw, db, rows, cols, splits);
  else if (dtype == CASE_BF16) hipLaunchKernelGGL(rowdot_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, g, (const bf16_t*)x, w, (bf16_t*)dx, dw, db, rows, cols, splits);
  else return case_set_error(CASE_E_UNSUPPORTED, "case_rowdot_bwd: dtype %d", dtype);
  return case_check_launch("case_rowdot_bwd");
}
