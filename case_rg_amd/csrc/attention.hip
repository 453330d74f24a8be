// K4/K5/K6 fused multi-head attention for gfx950 (bf16 in, f32 accumulate): no [L, L] score tensor ever reaches HBM.
//
// All four kernels share one skeleton.  One side of the attention matrix is STATIONARY: a wave keeps 32 of its
// rows (queries for fwd / dQ, keys for dK / dV) as MFMA B-operand fragments in registers, ONE ROW PER LANE, so every
// per-row quantity (running max / sum, log-sum-exp, delta) is lane-local.  The other side STREAMS through LDS in
// tiles of TS rows; products are oriented so that the 32x32 accumulator of the first product (streamed row index in
// the registers, stationary row index on the lanes) is, after bf16 packing, directly the B operand of the second
// product -- no lane exchange and no LDS round trip (v_mfma_f32_32x32x16_bf16, accumulator registers 8s..8s+7 form
// the fragment of k-step s).  Output accumulators are transposed ([d][stationary row]): 16 f32 registers per
// 32-wide slice of the head dimension.
//
//   fwd : S^T = K Q^T (A = K rows from LDS, B = Q^T regs) -> online softmax over the streamed keys (lane-local)
//         O^T += V^T P^T   (A = V^T via ds_read_b64_tr_b16 of the V tile, B = packed P^T)
//   dQ  : S^T = K Q^T, dP^T = V dO^T, dS^T = P^T (dP^T - delta) ; dQ^T += K^T dS^T
//   dK  : S = Q K^T (A = Q rows, B = K regs), dP = dO V^T (B = V regs), dS ; dK^T += Q^T dS
//   dV  : S = Q K^T, P ; dV^T += dO^T P
// Recomputing S in each backward kernel costs 8 instead of 5 products per tile, but keeps the register budget of the
// head_dim-320 blocks (5H/8, 73 % of CaSE's FLOPs) inside one wave per SIMD: 80 (stationary fragments) [+80] + 160
// (output accumulators) + tiles.
//
// LDS images (bf16): row-read operand [TS][D*2 + 16 B] (row stride = 36 dwords mod 64 -> conflict-free ds_read_b128);
// transposed-read operand [TS][stride = 48 dwords mod 64] (four k-rows of a ds_read_b64_tr_b16 block land on
// disjoint bank windows).  Tiles are double-buffered; the next tile's global loads are issued before the MFMA phase
// and written to the other buffer after it (one barrier per tile).
//
// Dropout uses the same counter RNG and the same element index ((n*h + head)*Lq + q)*Lk + k as the unfused
// softmax kernel, so fused and unfused paths draw identical masks.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

namespace {

struct FaArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v;  // offset-adjusted bases
  int64_t ldq, ldk, ldv, sq, sk, sv;                   // row strides and per-sequence strides (elements)
  bf16_t* o; int64_t ldo, so;                          // fwd output [N, Lq, heads*D]
  float* lse;                                          // [N, heads, Lq]  (fwd: out; bwd: in)
  const uint8_t* key_valid;                            // [N, Lk] or null
  // backward only
  const bf16_t* dout; int64_t lddo, sdo;               // dO [N, Lq, heads*D]
  const float* delta;                                  // [N, heads, Lq] = rowsum(dO * O)
  float* dq; float* dk; float* dv;                     // f32 gradient slices written in the source layout
  int64_t lddq, sdq, lddk, sdk, lddv, sdv;
  int Lq, Lk, heads, causal, nblk, tiles;
  float scale, drop_p;
  uint64_t seed, offset;
};

__device__ __forceinline__ int xcd_remap(int pid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = pid & 7, idx = pid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int D> struct Geo {
  static constexpr int ROW = D * 2 + 16;  // row-read image stride (bytes)
  // transposed-read image stride: smallest stride >= D*2 with (stride/4) % 64 == 48
  static constexpr int TRS = ((D * 2 + 63) / 256) * 256 + 192 >= D * 2 ? ((D * 2 + 63) / 256) * 256 + 192 : ((D * 2 + 63) / 256) * 256 + 448;
};

// ---- tile staging: [TS rows][D] bf16 from global (row stride ld) into an LDS image with row stride STRIDE ----------
// Per-thread state is one 32-bit byte offset per 16-byte chunk (set once); the tile advances through the uniform base
// pointer, so the prefetch costs no 64-bit address registers inside the main loop.
template <int D, int TS>
struct Stage {
  static constexpr int CH = D / 8;              // 16-byte chunks per row
  static constexpr int NV = TS * CH / 256;      // chunks per thread
  static_assert(TS * CH % 256 == 0, "tile must be a whole number of 256-thread passes");
  u32x4 r[NV];
  unsigned off[NV];
  unsigned ok;
  __device__ __forceinline__ void init(int64_t ld) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = threadIdx.x + i * 256;
      off[i] = (unsigned)((c / CH) * ld * 2 + (c % CH) * 16);
    }
  }
  // tile_base = first row of the tile (uniform pointer); rows_left = valid rows from tile_base on
  __device__ __forceinline__ void load(const bf16_t* __restrict__ tile_base, int rows_left) {
    ok = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int row = (threadIdx.x + i * 256) / CH;
      const bool in = row < rows_left;
      r[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(tile_base) + (in ? off[i] : 0u));
      ok |= (in ? 1u : 0u) << i;
    }
  }
  template <int STRIDE>
  __device__ __forceinline__ void store(char* lds) const {
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = threadIdx.x + i * 256;
      *reinterpret_cast<u32x4*>(lds + (c / CH) * STRIDE + (c % CH) * 16) = ((ok >> i) & 1u) ? r[i] : z;
    }
  }
};

// A operand, row read: rows = streamed rows (32 per tile t32), k = head-dim step s (16 values)
template <int STRIDE>
__device__ __forceinline__ bf16x8 frag_rows(const char* lds, int t32, int s) {
  const int l = threadIdx.x & 63;
  return *reinterpret_cast<const bf16x8*>(lds + (t32 + (l & 31)) * STRIDE + (2 * s + (l >> 5)) * 16);
}

// A operand, transposed read: rows = head-dim slice dt (32 values), k = streamed rows of k-step s2 (16 rows) in the
// order the packed accumulator uses: element j of lane half h is streamed row 16 s2 + 8 (j>>2) + 4 h + (j&3).
template <int STRIDE>
__device__ __forceinline__ bf16x8 frag_tr(const char* lds, int row0, int dt) {
  const int l = threadIdx.x & 63;
  const int q = (l & 15) >> 2, p = l & 3;
  const int col = dt * 32 + 16 * ((l >> 4) & 1) + 4 * p;
  const int off = (row0 + 4 * (l >> 5) + q) * STRIDE + col * 2;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + off));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + off + 8 * STRIDE));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

// accumulator registers 8*s2 .. 8*s2+7 -> B-operand fragment of k-step s2
__device__ __forceinline__ bf16x8 pack_acc(const f32x16& a, int s2) {
  u32x4_t w;
#pragma unroll
  for (int j = 0; j < 4; ++j) w[j] = f32x2_to_bf16x2(a[8 * s2 + 2 * j], a[8 * s2 + 2 * j + 1]);
  return *reinterpret_cast<bf16x8*>(&w);
}

// stationary fragments: lane (c = l&31, h = l>>5) holds row (row0 + c), head-dim values 16 s + 8 h .. + 7
template <int D>
__device__ __forceinline__ void load_stationary(bf16x8 (&f)[D / 16], const bf16_t* base, int64_t ld, int row0, int rows) {
  const int l = threadIdx.x & 63;
  int row = row0 + (l & 31);
  row = row < rows ? row : rows - 1;  // clamp: out-of-range lanes compute garbage that is never stored
  const bf16_t* p = base + (int64_t)row * ld + 8 * (l >> 5);
#pragma unroll
  for (int s = 0; s < D / 16; ++s) f[s] = *reinterpret_cast<const bf16x8*>(p + 16 * s);
}

// Validity of the TS streamed rows of a tile as a wave-uniform 64-bit mask (bit r = row0 + r is a real, unpadded row):
// one coalesced byte load per wave instead of one dependent global load per accumulator element.  The byte is fetched
// early (`fetch`, next to the tile prefetch) and turned into the mask late (`ballot`, after the MFMA phase).
struct RowMask {
  unsigned char byte;
  __device__ __forceinline__ void fetch(const uint8_t* __restrict__ valid, int row0, int rows) {
    const int r = row0 + (threadIdx.x & 63);
    byte = (r < rows) ? (valid ? valid[r] : (unsigned char)1) : (unsigned char)0;
  }
  __device__ __forceinline__ unsigned long long ballot() const { return __ballot(byte != 0); }
};

constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.69314718055994531f;
// The softmax runs in the base-2 domain: scores are scaled by scale * log2(e) once, every exponential is a bare v_exp_f32.
constexpr float RESCALE_THR = 8.f * LOG2E;

// Uniforms of four consecutive counters (the four keys one accumulator register group of a lane covers): two hashes when
// the first counter is even (element pairs share a hash, common.h), four otherwise.
__device__ __forceinline__ void rng4(uint64_t seed, uint64_t i0, float (&u)[4]) {
  if ((i0 & 1) == 0) {
    rng_uniform2(seed, i0, u[0], u[1]);
    rng_uniform2(seed, i0 + 2, u[2], u[3]);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) u[k] = rng_uniform(seed, i0 + k);
  }
}

// streamed row index of accumulator register e in a 32-row tile
__device__ __forceinline__ int acc_row(int e, int half) { return (e & 3) + 8 * (e >> 2) + 4 * half; }

// =====================================================================================================
// forward
// =====================================================================================================
template <int D, int TS>
__global__ __launch_bounds__(256, (D <= 128 ? 2 : 1)) void fa_fwd_kernel(const FaArgs a) {
  constexpr int KROW = Geo<D>::ROW, VROW = Geo<D>::TRS;
  constexpr int KBYTES = TS * KROW, VBYTES = TS * VROW, BUF = KBYTES + VBYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int pid = xcd_remap(blockIdx.x, a.nblk);
  const int qt = pid % a.tiles, head = (pid / a.tiles) % a.heads, n = pid / (a.tiles * a.heads);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
  const int q0 = qt * 128 + wave * 32, qi = q0 + (lane & 31);

  const bf16_t* Q = a.q + (int64_t)n * a.sq + head * D;
  const bf16_t* K = a.k + (int64_t)n * a.sk + head * D;
  const bf16_t* V = a.v + (int64_t)n * a.sv + head * D;
  const uint8_t* kv = a.key_valid ? a.key_valid + (int64_t)n * a.Lk : nullptr;

  bf16x8 qf[D / 16];
  load_stationary<D>(qf, Q, a.ldq, q0, a.Lq);

  f32x16 o[D / 32];
#pragma unroll
  for (int t = 0; t < D / 32; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[t][e] = 0.f;
  float m = -INFINITY, lsum = 0.f;

  int ntiles = (a.Lk + TS - 1) / TS;
  if (a.causal) {  // keys beyond the last query of this workgroup never contribute
    const int last = min(a.Lq, qt * 128 + 128) - 1;
    ntiles = min(ntiles, last / TS + 1);
  }
  Stage<D, TS> sk, sv;
  RowMask rm;
  sk.init(a.ldk);
  sv.init(a.ldv);
  sk.load(K, a.Lk);
  sv.load(V, a.Lk);
  rm.fetch(kv, 0, a.Lk);
  sk.template store<KROW>(smem);
  sv.template store<VROW>(smem + KBYTES);
  unsigned long long mask = rm.ballot();
  __syncthreads();
  const float keep_scale = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
  const float scale2 = a.scale * LOG2E;
  const uint64_t rng_row = a.offset + (uint64_t)(((int64_t)n * a.heads + head) * a.Lq + qi) * (uint64_t)a.Lk;

  for (int t = 0; t < ntiles; ++t) {
    const char* kb = smem + (t & 1) * BUF;
    const char* vb = kb + KBYTES;
    if (t + 1 < ntiles) {
      sk.load(K + (int64_t)(t + 1) * TS * a.ldk, a.Lk - (t + 1) * TS);
      sv.load(V + (int64_t)(t + 1) * TS * a.ldv, a.Lk - (t + 1) * TS);
      rm.fetch(kv, (t + 1) * TS, a.Lk);
    }
    const unsigned long long mrow = mask >> (4 * half);  // bit (e&3) + 8 (e>>2) + 32 kt of this lane's half
#pragma unroll
    for (int kt = 0; kt < TS / 32; ++kt) {
      // ---- S^T tile: 32 keys (registers) x 32 queries (lanes)
      f32x16 st;
#pragma unroll
      for (int e = 0; e < 16; ++e) st[e] = 0.f;
#pragma unroll
      for (int s = 0; s < D / 16; ++s) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<KROW>(kb, kt * 32, s), qf[s], st, 0, 0, 0);
        if (D > 128 && (s & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // bound the fragment prefetch depth (VGPR budget)
      }
      // ---- mask + online softmax (per lane = per query)
      const int key0 = t * TS + kt * 32;
      float mx = -INFINITY;
      // interior tile: every key valid and (causal) not beyond the wave's first query -> no per-element mask work
      const bool all_ok = ((mask >> (32 * kt)) & 0xffffffffull) == 0xffffffffull && (!a.causal || key0 + 31 <= q0);
      if (all_ok) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          st[e] *= scale2;
          mx = fmaxf(mx, st[e]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = key0 + acc_row(e, half);
          const bool ok = ((mrow >> ((e & 3) + 8 * (e >> 2) + 32 * kt)) & 1ull) && (!a.causal || key <= qi);
          st[e] = ok ? st[e] * scale2 : -INFINITY;
          mx = fmaxf(mx, st[e]);
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      // lazy rescale: the running reference max moves only when some query's tile max exceeds it by more than THR;
      // probabilities then stay below e^THR (fine for the bf16 P operand, l and O accumulate in f32), and the
      // O-wide rescale (whose accumulators live in AGPRs) runs on a few early tiles only.  The decision precedes the
      // exponentiation of this tile and follows the previous tile's P V, so every term is scaled exactly once.
      if (__any(mx > m + RESCALE_THR)) {
        const float m_new = fmaxf(m, mx);
        const float alpha = (m_new == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m - m_new);
        lsum *= alpha;
        m = m_new;
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt) {
#pragma unroll
          for (int e = 0; e < 16; ++e) o[dt][e] *= alpha;
          __builtin_amdgcn_sched_barrier(0);  // one 16-register slice at a time through the arch VGPRs
        }
      }
      float ps = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float p = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(st[e] - m);
        ps += p;
        st[e] = p;
      }
      lsum += ps;
      if (a.drop_p > 0.f) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          float u[4];
          rng4(a.seed, rng_row + (uint64_t)(key0 + 8 * gq + 4 * half), u);
#pragma unroll
          for (int k = 0; k < 4; ++k) st[4 * gq + k] = u[k] >= a.drop_p ? st[4 * gq + k] * keep_scale : 0.f;
        }
      }
      // ---- O^T += V^T P^T
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pf = pack_acc(st, s2);
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt) {
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<VROW>(vb, kt * 32 + 16 * s2, dt), pf, o[dt], 0, 0, 0);
          if (D > 128 && (dt & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (t + 1 < ntiles) {
      char* nb = smem + ((t + 1) & 1) * BUF;
      sk.template store<KROW>(nb);
      sv.template store<VROW>(nb + KBYTES);
      mask = rm.ballot();
    }
    __syncthreads();
  }

  // ---- finalise: O = O^T / l, LSE = ln 2 * m + log(l)  (m is in the base-2 domain)
  lsum += __shfl_xor(lsum, 32, 64);
  const float inv = lsum > 0.f ? 1.f / lsum : 0.f;
  if (qi < a.Lq) {
    bf16_t* orow = a.o + (int64_t)n * a.so + (int64_t)qi * a.ldo + head * D;
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint32_t w0 = f32x2_to_bf16x2(o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv);
        const uint32_t w1 = f32x2_to_bf16x2(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
        *reinterpret_cast<uint2*>(orow + dt * 32 + 8 * g + 4 * half) = make_uint2(w0, w1);
      }
    if (half == 0) a.lse[((int64_t)n * a.heads + head) * a.Lq + qi] = lsum > 0.f ? m * LN2 + __logf(lsum) : -INFINITY;  // natural log
  }
}

// =====================================================================================================
// backward, head_dim <= 128 (two waves per SIMD): delta, dQ (query-stationary), dK+dV (key-stationary)
// =====================================================================================================
// delta[n, head, q] = sum_d dO[n, q, head, d] * O[n, q, head, d]   (one wave per 64 (row, head) pairs)
template <int D>
__global__ __launch_bounds__(256) void fa_delta_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ out,
                                                       float* __restrict__ delta, int64_t rows, int heads, int Lq) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (n*Lq + q)*heads + head
  if (i >= rows * heads) return;
  const int64_t row = i / heads;
  const int head = (int)(i % heads);
  const bf16_t* a = dout + row * heads * D + head * D;
  const bf16_t* b = out + row * heads * D + head * D;
  float acc = 0.f;
#pragma unroll
  for (int c = 0; c < D; c += 8) {
    const uint4 x = *reinterpret_cast<const uint4*>(a + c), y = *reinterpret_cast<const uint4*>(b + c);
    const uint32_t xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      acc += __uint_as_float(xs[k] << 16) * __uint_as_float(ys[k] << 16) +
             __uint_as_float(xs[k] & 0xffff0000u) * __uint_as_float(ys[k] & 0xffff0000u);
  }
  const int64_t n = row / Lq, q = row % Lq;
  delta[(n * heads + head) * Lq + q] = acc;
}

// transposed accumulators [d][stationary row] -> bf16 rows of the gradient slice
template <int D>
__device__ __forceinline__ void store_transposed(const f32x16 (&acc)[D / 32], bf16_t* row_ptr, int half, float mul) {
#pragma unroll
  for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const uint32_t w0 = f32x2_to_bf16x2(acc[dt][4 * g] * mul, acc[dt][4 * g + 1] * mul);
      const uint32_t w1 = f32x2_to_bf16x2(acc[dt][4 * g + 2] * mul, acc[dt][4 * g + 3] * mul);
      *reinterpret_cast<uint2*>(row_ptr + dt * 32 + 8 * g + 4 * half) = make_uint2(w0, w1);
    }
}

struct BwdOut {  // bf16 gradient slices, addressed like q / k / v
  bf16_t* dq; bf16_t* dk; bf16_t* dv;
};

// ---- dQ: stationary queries (lane = query); streams K (row + transposed images) and V (row image) -----------------
template <int D, int TS>
__global__ __launch_bounds__(256, 2) void fa_bwd_dq_kernel(const FaArgs a, const BwdOut g) {
  constexpr int RS = Geo<D>::ROW, TR = Geo<D>::TRS;
  constexpr int KR = 0, KT = TS * RS, VR = KT + TS * TR;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int pid = xcd_remap(blockIdx.x, a.nblk);
  const int qt = pid % a.tiles, head = (pid / a.tiles) % a.heads, n = pid / (a.tiles * a.heads);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
  const int q0 = qt * 128 + wave * 32, qi = q0 + (lane & 31);
  const bf16_t* Q = a.q + (int64_t)n * a.sq + head * D;
  const bf16_t* K = a.k + (int64_t)n * a.sk + head * D;
  const bf16_t* V = a.v + (int64_t)n * a.sv + head * D;
  const bf16_t* DO = a.dout + (int64_t)n * a.sdo + head * D;
  const uint8_t* kv = a.key_valid ? a.key_valid + (int64_t)n * a.Lk : nullptr;

  bf16x8 qf[D / 16], dof[D / 16];
  load_stationary<D>(qf, Q, a.ldq, q0, a.Lq);
  load_stationary<D>(dof, DO, a.lddo, q0, a.Lq);
  const int64_t stat = ((int64_t)n * a.heads + head) * a.Lq + (qi < a.Lq ? qi : a.Lq - 1);
  const float lse2_q = a.lse[stat] * LOG2E, delta_q = a.delta[stat];  // base-2 domain: p = 2^(scale2 s - lse2)
  const float scale2 = a.scale * LOG2E;
  f32x16 acc[D / 32];
#pragma unroll
  for (int t = 0; t < D / 32; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  int ntiles = (a.Lk + TS - 1) / TS;
  if (a.causal) ntiles = min(ntiles, (min(a.Lq, qt * 128 + 128) - 1) / TS + 1);
  const float keep_scale = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
  const uint64_t rng_row = a.offset + (uint64_t)(((int64_t)n * a.heads + head) * a.Lq + qi) * (uint64_t)a.Lk;
  Stage<D, TS> sk, sv;
  RowMask rm;
  sk.init(a.ldk);
  sv.init(a.ldv);

  for (int t = 0; t < ntiles; ++t) {
    sk.load(K + (int64_t)t * TS * a.ldk, a.Lk - t * TS);
    sv.load(V + (int64_t)t * TS * a.ldv, a.Lk - t * TS);
    rm.fetch(kv, t * TS, a.Lk);
    __syncthreads();  // previous tile fully consumed
    sk.template store<RS>(smem + KR);
    sk.template store<TR>(smem + KT);
    sv.template store<RS>(smem + VR);
    const unsigned long long mrow_all = rm.ballot(), mrow = mrow_all >> (4 * half);
    __syncthreads();
#pragma unroll
    for (int kt = 0; kt < TS / 32; ++kt) {
      f32x16 st, dp;
#pragma unroll
      for (int e = 0; e < 16; ++e) st[e] = dp[e] = 0.f;
#pragma unroll
      for (int s = 0; s < D / 16; ++s) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<RS>(smem + KR, kt * 32, s), qf[s], st, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<RS>(smem + VR, kt * 32, s), dof[s], dp, 0, 0, 0);
      }
      const int key0 = t * TS + kt * 32;
      if (a.drop_p > 0.f) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          float u[4];
          rng4(a.seed, rng_row + (uint64_t)(key0 + 8 * gq + 4 * half), u);
#pragma unroll
          for (int k = 0; k < 4; ++k) dp[4 * gq + k] = u[k] >= a.drop_p ? dp[4 * gq + k] * keep_scale : 0.f;
        }
      }
      const bool all_ok = ((mrow_all >> (32 * kt)) & 0xffffffffull) == 0xffffffffull && (!a.causal || key0 + 31 <= q0);
      if (all_ok) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float p = __builtin_amdgcn_exp2f(fmaf(st[e], scale2, -lse2_q));
          st[e] = p * (dp[e] - delta_q) * a.scale;  // dS^T
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = key0 + acc_row(e, half);
          const bool ok = ((mrow >> ((e & 3) + 8 * (e >> 2) + 32 * kt)) & 1ull) && (!a.causal || key <= qi);
          const float p = ok ? __builtin_amdgcn_exp2f(fmaf(st[e], scale2, -lse2_q)) : 0.f;
          st[e] = p * (dp[e] - delta_q) * a.scale;  // dS^T
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 df = pack_acc(st, s2);
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt)
          acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<TR>(smem + KT, kt * 32 + 16 * s2, dt), df, acc[dt], 0, 0, 0);
      }
    }
  }
  if (qi < a.Lq) store_transposed<D>(acc, g.dq + (int64_t)n * a.sq + (int64_t)qi * a.ldq + head * D, half, 1.f);
}

// ---- dK + dV: stationary keys (lane = key); streams Q and dO (row + transposed images each) ------------------------
template <int D, int TS>
__global__ __launch_bounds__(256, 2) void fa_bwd_dkv_kernel(const FaArgs a, const BwdOut g) {
  constexpr int RS = Geo<D>::ROW, TR = Geo<D>::TRS;
  constexpr int QR = 0, QT = TS * RS, OR_ = QT + TS * TR, OT = OR_ + TS * RS, ST = OT + TS * TR;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lse_s = reinterpret_cast<float*>(smem + ST);
  float* del_s = lse_s + TS;
  const int pid = xcd_remap(blockIdx.x, a.nblk);
  const int ktile = pid % a.tiles, head = (pid / a.tiles) % a.heads, n = pid / (a.tiles * a.heads);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
  const int k0 = ktile * 128 + wave * 32, ki = k0 + (lane & 31);
  const bf16_t* Q = a.q + (int64_t)n * a.sq + head * D;
  const bf16_t* K = a.k + (int64_t)n * a.sk + head * D;
  const bf16_t* V = a.v + (int64_t)n * a.sv + head * D;
  const bf16_t* DO = a.dout + (int64_t)n * a.sdo + head * D;
  const bool key_ok = ki < a.Lk && (!a.key_valid || a.key_valid[(int64_t)n * a.Lk + ki]);

  bf16x8 kf[D / 16], vf[D / 16];
  load_stationary<D>(kf, K, a.ldk, k0, a.Lk);
  load_stationary<D>(vf, V, a.ldv, k0, a.Lk);
  f32x16 dk[D / 32], dv[D / 32];
#pragma unroll
  for (int t = 0; t < D / 32; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) dk[t][e] = dv[t][e] = 0.f;

  const int ntiles = (a.Lq + TS - 1) / TS;
  const int tbegin = a.causal ? (ktile * 128) / TS : 0;  // queries before the first key of this workgroup see none of its keys
  const float keep_scale = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
  const uint64_t rng_base = a.offset + (uint64_t)(((int64_t)n * a.heads + head) * a.Lq) * (uint64_t)a.Lk + (uint64_t)ki;
  const float scale2 = a.scale * LOG2E;
  const float* lse_g = a.lse + ((int64_t)n * a.heads + head) * a.Lq;
  const float* del_g = a.delta + ((int64_t)n * a.heads + head) * a.Lq;
  Stage<D, TS> sq, so;
  sq.init(a.ldq);
  so.init(a.lddo);

  for (int t = tbegin; t < ntiles; ++t) {
    sq.load(Q + (int64_t)t * TS * a.ldq, a.Lq - t * TS);
    so.load(DO + (int64_t)t * TS * a.lddo, a.Lq - t * TS);
    float stat = 0.f;
    if (threadIdx.x < 2 * TS) {
      const int r = t * TS + (threadIdx.x % TS);
      stat = threadIdx.x < TS ? (r < a.Lq ? lse_g[r] * LOG2E : INFINITY) : (r < a.Lq ? del_g[r] : 0.f);  // lse = +inf -> p = 0
    }
    __syncthreads();
    sq.template store<RS>(smem + QR);
    sq.template store<TR>(smem + QT);
    so.template store<RS>(smem + OR_);
    so.template store<TR>(smem + OT);
    if (threadIdx.x < 2 * TS) lse_s[threadIdx.x] = stat;
    __syncthreads();
#pragma unroll
    for (int qt = 0; qt < TS / 32; ++qt) {
      f32x16 st, dp;
#pragma unroll
      for (int e = 0; e < 16; ++e) st[e] = dp[e] = 0.f;
#pragma unroll
      for (int s = 0; s < D / 16; ++s) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<RS>(smem + QR, qt * 32, s), kf[s], st, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<RS>(smem + OR_, qt * 32, s), vf[s], dp, 0, 0, 0);
      }
      const int qrow0 = qt * 32;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = qrow0 + acc_row(e, half);
        const int query = t * TS + r;
        const bool ok = key_ok && (!a.causal || ki <= query);
        const float p = ok ? __builtin_amdgcn_exp2f(fmaf(st[e], scale2, -lse_s[r])) : 0.f;
        float keep = 1.f;
        if (a.drop_p > 0.f) keep = rng_uniform(a.seed, rng_base + (uint64_t)query * (uint64_t)a.Lk) >= a.drop_p ? keep_scale : 0.f;
        dp[e] = p * (dp[e] * keep - del_s[r]) * a.scale;  // dS
        st[e] = p * keep;                                  // dropped P
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pf = pack_acc(st, s2), df = pack_acc(dp, s2);
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<TR>(smem + OT, qt * 32 + 16 * s2, dt), pf, dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<TR>(smem + QT, qt * 32 + 16 * s2, dt), df, dk[dt], 0, 0, 0);
        }
      }
    }
  }
  if (ki < a.Lk) {
    store_transposed<D>(dk, g.dk + (int64_t)n * a.sk + (int64_t)ki * a.ldk + head * D, half, 1.f);
    store_transposed<D>(dv, g.dv + (int64_t)n * a.sv + (int64_t)ki * a.ldv + head * D, half, 1.f);
  }
}

template <int D, int TS>
int launch_bwd(FaArgs a, const BwdOut& g, const bf16_t* out, hipStream_t s) {
  const int64_t rows = (int64_t)(a.sdo / a.lddo) * 0 + 0;  // unused
  (void)rows;
  constexpr int RS = Geo<D>::ROW, TR = Geo<D>::TRS;
  const size_t lds_dq = (size_t)TS * (2 * RS + TR);
  const size_t lds_dkv = (size_t)TS * (2 * RS + 2 * TR) + 2 * TS * 4;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_bwd_dq_kernel<D, TS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dq);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_bwd_dkv_kernel<D, TS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dkv);
    attr = true;
  }
  // dQ: one workgroup per 128 queries; dK/dV: one per 128 keys
  a.tiles = (a.Lq + 127) / 128;
  a.nblk = a.tiles * a.heads * (int)(a.sdq);  // sdq carries N here (set by the caller)
  hipLaunchKernelGGL((fa_bwd_dq_kernel<D, TS>), dim3(a.nblk), dim3(256), lds_dq, s, a, g);
  a.tiles = (a.Lk + 127) / 128;
  a.nblk = a.tiles * a.heads * (int)(a.sdq);
  hipLaunchKernelGGL((fa_bwd_dkv_kernel<D, TS>), dim3(a.nblk), dim3(256), lds_dkv, s, a, g);
  return case_check_launch("case_attention_bwd");
}

template <int D, int TS>
int launch_fwd(const FaArgs& a, hipStream_t s) {
  const size_t lds = 2 * (size_t)(TS * Geo<D>::ROW + TS * Geo<D>::TRS);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_fwd_kernel<D, TS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipLaunchKernelGGL((fa_fwd_kernel<D, TS>), dim3(a.nblk), dim3(256), lds, s, a);
  return case_check_launch("case_attention_fwd");
}

}  // namespace

extern "C" int case_attention_supported(int64_t head_dim) { return head_dim == 64 || head_dim == 320; }

extern "C" int case_attention_fwd(const CaseAttnDesc* d, const void* q, const void* k, const void* v, const uint8_t* key_valid,
                                  void* out, float* lse, case_stream_t stream) {
  CASE_REQUIRE(d && q && k && v && out && lse, "case_attention_fwd: null argument");
  CASE_REQUIRE(d->N > 0 && d->heads > 0 && d->Lq > 0 && d->Lk > 0, "case_attention_fwd: empty problem");
  CASE_REQUIRE(case_attention_supported(d->head_dim), "case_attention_fwd: head_dim %lld not built (64, 320)", (long long)d->head_dim);
  CASE_REQUIRE(d->ldq % 8 == 0 && d->ldk % 8 == 0 && d->ldv % 8 == 0 && d->sq % 8 == 0 && d->sk % 8 == 0 && d->sv % 8 == 0 &&
                   (uintptr_t)q % 16 == 0 && (uintptr_t)k % 16 == 0 && (uintptr_t)v % 16 == 0 && (uintptr_t)out % 8 == 0 && d->ldo % 4 == 0,
               "case_attention_fwd: operands must be 16-byte aligned with strides that are multiples of 8 elements");
  CASE_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f, "case_attention_fwd: drop_p out of range");
  FaArgs a = {};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v;
  a.ldq = d->ldq; a.ldk = d->ldk; a.ldv = d->ldv; a.sq = d->sq; a.sk = d->sk; a.sv = d->sv;
  a.o = (bf16_t*)out; a.ldo = d->ldo; a.so = d->so; a.lse = lse; a.key_valid = key_valid;
  a.Lq = (int)d->Lq; a.Lk = (int)d->Lk; a.heads = (int)d->heads; a.causal = d->causal;
  a.tiles = (a.Lq + 127) / 128;
  const int64_t nblk = (int64_t)a.tiles * d->heads * d->N;
  CASE_REQUIRE(nblk < (1ll << 31), "case_attention_fwd: grid too large");
  a.nblk = (int)nblk;
  a.scale = d->scale; a.drop_p = d->drop_p; a.seed = d->seed; a.offset = d->offset;
  hipStream_t s = (hipStream_t)stream;
  if (d->head_dim == 64) return launch_fwd<64, 64>(a, s);
  return launch_fwd<320, 32>(a, s);
}

extern "C" int case_attention_bwd_supported(int64_t head_dim) { return head_dim == 64; }

extern "C" int case_attention_bwd(const CaseAttnDesc* d, const void* q, const void* k, const void* v, const uint8_t* key_valid,
                                  const void* out, const float* lse, const void* dout, float* delta, void* dq, void* dk, void* dv,
                                  case_stream_t stream) {
  CASE_REQUIRE(d && q && k && v && out && lse && dout && delta && dq && dk && dv, "case_attention_bwd: null argument");
  CASE_REQUIRE(d->N > 0 && d->heads > 0 && d->Lq > 0 && d->Lk > 0, "case_attention_bwd: empty problem");
  CASE_REQUIRE(case_attention_bwd_supported(d->head_dim), "case_attention_bwd: head_dim %lld not built (64)", (long long)d->head_dim);
  CASE_REQUIRE(d->ldq % 8 == 0 && d->ldk % 8 == 0 && d->ldv % 8 == 0 && d->sq % 8 == 0 && d->sk % 8 == 0 && d->sv % 8 == 0 &&
                   d->ldo % 8 == 0 && (uintptr_t)q % 16 == 0 && (uintptr_t)k % 16 == 0 && (uintptr_t)v % 16 == 0 &&
                   (uintptr_t)out % 16 == 0 && (uintptr_t)dout % 16 == 0 && (uintptr_t)dq % 8 == 0 && (uintptr_t)dk % 8 == 0 &&
                   (uintptr_t)dv % 8 == 0,
               "case_attention_bwd: operands must be 16-byte aligned with strides that are multiples of 8 elements");
  CASE_REQUIRE(d->ldo == d->heads * d->head_dim && d->so == d->Lq * d->ldo, "case_attention_bwd: out / dout must be contiguous [N, Lq, heads*head_dim]");
  FaArgs a = {};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v;
  a.ldq = d->ldq; a.ldk = d->ldk; a.ldv = d->ldv; a.sq = d->sq; a.sk = d->sk; a.sv = d->sv;
  a.lse = const_cast<float*>(lse); a.key_valid = key_valid;
  a.dout = (const bf16_t*)dout; a.lddo = d->ldo; a.sdo = d->so; a.delta = delta;
  a.Lq = (int)d->Lq; a.Lk = (int)d->Lk; a.heads = (int)d->heads; a.causal = d->causal;
  a.scale = d->scale; a.drop_p = d->drop_p; a.seed = d->seed; a.offset = d->offset;
  a.sdq = d->N;  // launch_bwd reads N from here
  const int64_t blocks = ((int64_t)((d->Lq > d->Lk ? d->Lq : d->Lk) + 127) / 128) * d->heads * d->N;
  CASE_REQUIRE(blocks < (1ll << 31), "case_attention_bwd: grid too large");
  hipStream_t s = (hipStream_t)stream;
  const int64_t rows = d->N * d->Lq;
  hipLaunchKernelGGL((fa_delta_kernel<64>), dim3((unsigned)((rows * d->heads + 255) / 256)), dim3(256), 0, s, (const bf16_t*)dout,
                     (const bf16_t*)out, delta, rows, (int)d->heads, (int)d->Lq);
  BwdOut g = {(bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv};
  return launch_bwd<64, 64>(a, g, (const bf16_t*)out, s);
}
