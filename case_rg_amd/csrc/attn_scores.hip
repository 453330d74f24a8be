// K17: attention probabilities of the wide-head blocks with the softmax IN the GEMM (gfx950, bf16 in, f32 accumulate).
//
// The 5H TransformerBlocks (common/TransformerBlock.py:26: nn.MultiheadAttention at head_dim 5H/8 = 320) train on the path
// GEMM -> softmax -> GEMM with the probabilities saved for backward (their fused recompute backward is slower and less exact,
// DESIGN.md section 5).  That path wrote the scores S = alpha Q K^T as f32 [N, heads, Lq, Lk] (1.5 GB per block at cfg 2), read
// them back in the softmax kernel, and did the same again in backward with dP = dO V^T.  Here one workgroup owns 128 query rows
// times ALL Lk <= 384 keys of one (sequence, head), so a row of scores is complete inside the workgroup's accumulators:
//
//   forward : S tile -> key mask -> row softmax (f32) -> P (bf16) and the dropped-out copy Pd (bf16): no score tensor in HBM
//   backward: dP tile = dO V^T -> g = mask(dP)/(1-p) -> dS = P (g - rowsum(g P))  with P read once: no dP tensor, no softmax pass
//
// Same arithmetic points as case_gemm + case_softmax_fwd / _bwd: f32 scores, P rounded to bf16 once, Pd = bf16(p / (1 - p_drop))
// from the unrounded p, the same counter RNG and element index (common.h: attention dropout); dP stays f32 here (the GEMM path
// rounds it to bf16).
//
//   * 8 waves = 2 (query halves of 64) x 4 (key quarters of 96); v_mfma_f32_16x16x32_bf16 with the KEYS on the MFMA rows, so a
//     lane holds 4 consecutive keys of one query: accumulators acc[4 query blocks][6 key blocks], 96 registers;
//   * operands by LDS-DMA (buffer_load_dwordx4 ... lds) in K steps of 64 into two 64 KiB stages, images [rows][128 B] with the
//     16-byte chunks XOR-swizzled on the source side (chunk c of row r at slot c ^ ((r >> 1) & 7): conflict-free ds_read_b128
//     fragments, the scheme of gemm8w.inc); rows beyond Lq / Lk fall outside the buffer descriptors and read as zeros;
//   * row statistics: lane-local over 24 values, two cross-lane steps, then ONE exchange of (max, sum) pairs between the four key
//     quarters through LDS (each quarter exponentiates against its own maximum; the common factor 2^(m_w - m) / l follows);
//   * stores / loads of the [Lq, Lk] matrices are 16-byte vectors: one v_permlane16_swap pair turns two neighbouring key blocks
//     (4 + 4 keys per lane) into 8 consecutive keys per lane, 64 contiguous bytes per query row and block pair.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;

namespace attn_sc {

// Cache policies, measured at cfg 2 (tools/scores_bench.py, forward / backward ms): default everywhere 0.715 / 0.575; streaming (nt)
// loads of the query-side operand, which no other workgroup reads again, 0.699 / 0.565; nt result stores 0.757 / 0.617.
#ifndef SC_NT_A
#define SC_NT_A true
#endif
#ifndef SC_NT_ST
#define SC_NT_ST 0
#endif

constexpr int TM = 128, TN = 384, BKE = 64, NTHR = 512;
constexpr int A_BYTES = TM * 128, B_BYTES = TN * 128, STAGE = A_BYTES + B_BYTES;
constexpr int STATS_OFF = 2 * STAGE, STATS_BYTES = TM * 4 * 8, LDS_BYTES = STATS_OFF + STATS_BYTES;
constexpr float LOG2E = 1.4426950408889634f;

struct Args {
  const bf16_t* a;           // rows = queries: Q (forward) / dO (backward); element (n, q, head, j) at a[n sa + q lda + head d + j]
  const bf16_t* b;           // rows = keys:    K (forward) / V  (backward)
  int64_t lda, ldb, sa, sb;
  const uint8_t* key_valid;  // forward: [N, Lk] or null
  const bf16_t* p_in;        // backward: P [N, heads, Lq, Lk]
  bf16_t* out0;              // forward: P; backward: dS
  bf16_t* out1;              // forward: Pd (null without dropout)
  int N, heads, Lq, Lk, d, tiles, nblk;
  float alpha, drop_p;
  uint64_t seed, offset;
  const CaseStepState* state;  // nullable: offset += state->rng_base (ABI 600)
};

__device__ __forceinline__ i32x4 make_rsrc(const void* p, uint32_t bytes) {
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(size_t)p);
  r[1] = __builtin_amdgcn_readfirstlane((int)(((size_t)p) >> 32) & 0xffff);
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t as_rsrc(const void* p, uint32_t bytes) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((int)(size_t)p), hi = __builtin_amdgcn_readfirstlane((int)(((size_t)p) >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, (int)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
template <bool NT>  // NT: streaming cache policy for bytes no other workgroup reads again (the query-side operand)
__device__ __forceinline__ void dma16(i32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
  if constexpr (NT)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory", "m0");
  else
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory", "m0");
}
#pragma clang diagnostic pop

__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// Two neighbouring 16-key blocks in accumulator layout -- lane row g holds keys 4 g .. 4 g + 3 of each, packed x = block 0, y = block 1 --
// <-> 8 consecutive keys per lane: even lane rows give their y for the odd row's x.  Afterwards lane row g holds keys 8 g' .. 8 g' + 7 of
// the pair's 32, g' = bit-swapped g (0, 2, 1, 3).  The exchange is its own inverse.
__device__ __forceinline__ void pair_swap(uint32_t (&x)[2], uint32_t (&y)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const auto s = __builtin_amdgcn_permlane16_swap(x[i], y[i], false, false);
    x[i] = s[0];
    y[i] = s[1];
  }
}

#ifdef SC_STAMPS
// diagnostic build only (tools/scores_stamps.py): s_memtime at the phase boundaries of each workgroup's SECOND item, wave SC_STAMP_WAVE
__device__ uint64_t g_sc_stamps[256 * 16];
#ifndef SC_STAMP_WAVE
#define SC_STAMP_WAVE 0
#endif
#define SC_STAMP(i) { if (wave == SC_STAMP_WAVE && idx == slot + nwx) { const uint64_t t_ = __builtin_amdgcn_s_memtime(); \
    if (l == 0) g_sc_stamps[(int)blockIdx.x * 16 + (i)] = t_; } }
#else
#define SC_STAMP(i)
#endif

template <bool BWD>
__global__ __launch_bounds__(NTHR) void scores_kernel(const Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, l = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3, lr = l & 15, lg = l >> 4;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  float* stats = reinterpret_cast<float*>(smem + STATS_OFF);

  // Persistent: XCD x (blockIdx.x & 7) owns a contiguous range of work items -- (sequence, head, query tile), query tile fastest -- and
  // its workgroups walk it with stride (workgroups of the XCD): at any time the XCD works on neighbouring items, so the three query
  // tiles of a (sequence, head) read the same keys from one L2.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nwx = ((int)gridDim.x - xcd + 7) >> 3;
  const int per = g.nblk >> 3, rem = g.nblk & 7;
  const int first = xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per, count = per + (xcd < rem ? 1 : 0);
  struct Item { int n, head, m0, rows; i32x4 rsa, rsb; };
  auto item_of = [&](int idx) {
    Item t;
    const int pid = first + idx;
    const int qt = pid % g.tiles;
    t.head = (pid / g.tiles) % g.heads;
    t.n = pid / (g.tiles * g.heads);
    t.m0 = qt * TM;
    t.rows = min(TM, g.Lq - t.m0);
    const bf16_t* A0 = g.a + (int64_t)t.n * g.sa + (int64_t)t.m0 * g.lda + t.head * g.d;
    const bf16_t* B0 = g.b + (int64_t)t.n * g.sb + t.head * g.d;
    t.rsa = make_rsrc(A0, (uint32_t)(((int64_t)(t.rows - 1) * g.lda + g.d) * 2));
    t.rsb = make_rsrc(B0, (uint32_t)(((int64_t)(g.Lk - 1) * g.ldb + g.d) * 2));
    return t;
  };
  if (slot >= count) return;

  // DMA: one wave-instruction = 8 rows x 128 B; lane l -> row 8 gi + (l >> 3), slot l & 7 <- chunk (l & 7) ^ ((row >> 1) & 7)
  unsigned va[2], vb[6];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 8 * (wave * 2 + i) + (l >> 3);
    va[i] = (unsigned)(row * g.lda * 2 + (((l & 7) ^ ((row >> 1) & 7)) << 4));
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int row = 8 * (wave * 6 + i) + (l >> 3);
    vb[i] = (unsigned)(row * g.ldb * 2 + (((l & 7) ^ ((row >> 1) & 7)) << 4));
  }
#define SC_ISSUE(T, ST, KT)                                                                                          \
  {                                                                                                                  \
    const unsigned sb_ = lds0 + (ST) * STAGE, so_ = (unsigned)(KT) * 128u;                                           \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) dma16<SC_NT_A>((T).rsa, va[i], so_, sb_ + (wave * 2 + i) * 1024);            \
    _Pragma("unroll") for (int i = 0; i < 6; ++i) dma16<false>((T).rsb, vb[i], so_, sb_ + A_BYTES + (wave * 6 + i) * 1024);  \
  }
  const int nk = g.d / BKE;
  const int fb0 = lr * 128 + ((lg ^ (lr >> 1)) << 4), fb1 = lr * 128 + (((lg + 4) ^ (lr >> 1)) << 4);
  const float keep_scale = g.drop_p > 0.f ? 1.f / (1.f - g.drop_p) : 1.f;
  const uint32_t thr = rng_threshold(g.drop_p);
  const int gq = ((lg & 1) << 1) | (lg >> 1);  // after pair_swap this lane holds keys 8 gq .. 8 gq + 7 of a block pair
  // vector-memory operations of one epilogue that are YOUNGER than the next item's first DMAs (issued before the epilogue): the first
  // wait of the next K loop must not drain them
  const bool many_stores = !BWD && g.out1 != nullptr;

  // The DMA stream of this workgroup is continuous over its items: stream step q (item q / nk, K step q % nk) lands in stage q & 1 and
  // is issued two steps ahead, as soon as every wave has read step q - 2's fragments out of that stage (two K steps of 64 in flight:
  // one step ahead -- one 64 KiB request per 0.7 us of MFMA work -- left every K step waiting 2-3 us for its data).
  Item cur = item_of(slot);
  SC_ISSUE(cur, 0, 0)
  SC_ISSUE(cur, 1, 1)
  int gs = 0;
  for (int idx = slot; idx < count; idx += nwx) {
  const int n = cur.n, head = cur.head, m0 = cur.m0, rows = cur.rows;
  const bool first_item = idx == slot, has_nxt = idx + nwx < count;
  Item nxt = cur;
  if (has_nxt) nxt = item_of(idx + nwx);
  SC_STAMP(0)

  // validity of this lane's key columns 96 wc + 16 j + 4 lg + e as one byte each (loaded now, used in the epilogue)
  uint32_t kvw[6];
  if constexpr (!BWD) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int key = 96 * wc + 16 * j + 4 * lg;
      kvw[j] = key < g.Lk ? (g.key_valid ? *reinterpret_cast<const uint32_t*>(g.key_valid + (int64_t)n * g.Lk + key) : 0x01010101u) : 0u;
    }
  }

  f32x4 acc[4][6];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int kt = 0; kt < nk; ++kt, ++gs) {
    // This wave's share of stream step gs has landed.  Vector-memory operations retire in issue order; younger than step gs's DMAs
    // are step gs + 1's (8 per wave, if that step exists) and, at an item's first step, the previous item's result stores (12 or 24
    // per lane, issued whatever the bounds): leave those in flight.  (A smaller count than the true one only waits longer.)
    if (kt == 0 && !first_item) {
      if (many_stores) __builtin_amdgcn_s_waitcnt(0x8f70);  // vmcnt(32)
      else __builtin_amdgcn_s_waitcnt(0x4f74);              // vmcnt(20)
    } else if (kt + 1 < nk || has_nxt) {
      __builtin_amdgcn_s_waitcnt(0x0f78);                   // vmcnt(8)
    } else {
      __builtin_amdgcn_s_waitcnt(0x0f70);                   // vmcnt(0)
    }
    __syncthreads();  // everybody's share has
    SC_STAMP(1 + kt)
    const int st = gs & 1;
    const char* As = smem + st * STAGE + wr * (64 * 128);
    const char* Bs = smem + st * STAGE + A_BYTES + wc * (96 * 128);
    bf16x8 af[2][4], bfr[2][6];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      const int fb = kh ? fb1 : fb0;
#pragma unroll
      for (int i = 0; i < 4; ++i) af[kh][i] = *reinterpret_cast<const bf16x8*>(As + i * 2048 + fb);
#pragma unroll
      for (int j = 0; j < 6; ++j) bfr[kh][j] = *reinterpret_cast<const bf16x8*>(Bs + j * 2048 + fb);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the fragments are in registers
    __syncthreads();                     // everybody's are: the stage is free for stream step gs + 2
    if (kt + 2 < nk) SC_ISSUE(cur, st, kt + 2)
    else if (has_nxt) SC_ISSUE(nxt, st, kt + 2 - nk)
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[kh][j], af[kh][i], acc[i][j], 0, 0, 0);
  }
  SC_STAMP(8)
  // ---- epilogue.  acc[i][j][e] = score of query m0 + 64 wr + 16 i + lr, key 96 wc + 16 j + 4 lg + e -------------------------------------
  const int64_t grow0 = ((int64_t)n * g.heads + head) * g.Lq + m0;  // global row index of the tile's first query
  const uint32_t tile_bytes = (uint32_t)rows * (uint32_t)g.Lk * 2u;
  uint32_t rkey[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) rkey[i] = g.drop_p > 0.f ? rng_row_key(g.seed, g.offset + rng_base_of(g.state) + (uint64_t)(grow0 + 64 * wr + 16 * i + lr)) : 0u;
  // per-lane byte offset of (query block i, block pair pr) inside the tile's [rows, Lk] slice; columns beyond Lk -> out of range
  auto mat_off = [&](int i, int pr) -> int {
    const int key0 = 96 * wc + 32 * pr + 8 * gq;
    return key0 < g.Lk ? ((64 * wr + 16 * i + lr) * g.Lk + key0) * 2 : 0x7fffffff;
  };

  if constexpr (!BWD) {
    const float a2 = g.alpha * LOG2E;
    float mloc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float m = -INFINITY;
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float s = ((kvw[j] >> (8 * e)) & 0xffu) ? acc[i][j][e] * a2 : -INFINITY;
          acc[i][j][e] = s;
          m = fmaxf(m, s);
        }
      m = fmaxf(m, __shfl_xor(m, 16, 64));
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      const float mref = m == -INFINITY ? 0.f : m;
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p = __builtin_amdgcn_exp2f(acc[i][j][e] - mref);
          acc[i][j][e] = p;
          sum += p;
        }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      mloc[i] = m;
      if (lg == 0) *reinterpret_cast<float2*>(stats + ((64 * wr + 16 * i + lr) * 4 + wc) * 2) = make_float2(m, sum);
    }
    __syncthreads();
    SC_STAMP(10)
    const __amdgpu_buffer_rsrc_t rp = as_rsrc(g.out0 + grow0 * g.Lk, tile_bytes);
    const __amdgpu_buffer_rsrc_t rd = as_rsrc(g.out1 ? g.out1 + grow0 * g.Lk : nullptr, g.out1 ? tile_bytes : 0u);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float* sp = stats + (64 * wr + 16 * i + lr) * 8;
      const f32x4 s01 = *reinterpret_cast<const f32x4*>(sp), s23 = *reinterpret_cast<const f32x4*>(sp + 4);
      const float M = fmaxf(fmaxf(s01[0], s01[2]), fmaxf(s23[0], s23[2]));
      float f = 0.f;
      if (M > -INFINITY) {
        const float L = s01[1] * __builtin_amdgcn_exp2f(s01[0] - M) + s01[3] * __builtin_amdgcn_exp2f(s01[2] - M) +
                        s23[1] * __builtin_amdgcn_exp2f(s23[0] - M) + s23[3] * __builtin_amdgcn_exp2f(s23[2] - M);
        f = mloc[i] > -INFINITY ? __builtin_amdgcn_exp2f(mloc[i] - M) / L : 0.f;
      }
#pragma unroll
      for (int pr = 0; pr < 3; ++pr) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = acc[i][2 * pr][e] * f; v[4 + e] = acc[i][2 * pr + 1][e] * f; }
        uint32_t x[2] = {f32x2_to_bf16x2(v[0], v[1]), f32x2_to_bf16x2(v[2], v[3])};
        uint32_t y[2] = {f32x2_to_bf16x2(v[4], v[5]), f32x2_to_bf16x2(v[6], v[7])};
        pair_swap(x, y);
        const int vo = mat_off(i, pr);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{x[0], x[1], y[0], y[1]}, rp, vo, 0, SC_NT_ST);
        if (g.out1) {
          // columns 96 wc + 16 (2 pr + b) + 4 lg + e: pairs (col >> 1) and (col >> 1) + 1 of each block
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const uint32_t c0 = (uint32_t)(96 * wc + 16 * (2 * pr + b) + 4 * lg);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const uint32_t h = rng_pair_bits(rkey[i], (c0 >> 1) + t);
              v[4 * b + 2 * t] = (h & 0xffffu) >= thr ? v[4 * b + 2 * t] * keep_scale : 0.f;
              v[4 * b + 2 * t + 1] = (h >> 16) >= thr ? v[4 * b + 2 * t + 1] * keep_scale : 0.f;
            }
          }
          uint32_t dx[2] = {f32x2_to_bf16x2(v[0], v[1]), f32x2_to_bf16x2(v[2], v[3])};
          uint32_t dy[2] = {f32x2_to_bf16x2(v[4], v[5]), f32x2_to_bf16x2(v[6], v[7])};
          pair_swap(dx, dy);
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{dx[0], dx[1], dy[0], dy[1]}, rd, vo, 0, SC_NT_ST);
        }
      }
    }
  } else {
    const __amdgpu_buffer_rsrc_t rp = as_rsrc(g.p_in + grow0 * g.Lk, tile_bytes);
    const __amdgpu_buffer_rsrc_t rs = as_rsrc(g.out0 + grow0 * g.Lk, tile_bytes);
    u32x4 pw[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int pr = 0; pr < 3; ++pr) pw[i][pr] = __builtin_amdgcn_raw_buffer_load_b128(rp, mat_off(i, pr), 0, 0);
    // g = keep ? dP / (1 - p) : 0 while the probabilities are in flight
    if (g.drop_p > 0.f) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const uint32_t c0 = (uint32_t)(96 * wc + 16 * j + 4 * lg);
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const uint32_t h = rng_pair_bits(rkey[i], (c0 >> 1) + t);
            acc[i][j][2 * t] = (h & 0xffffu) >= thr ? acc[i][j][2 * t] * keep_scale : 0.f;
            acc[i][j][2 * t + 1] = (h >> 16) >= thr ? acc[i][j][2 * t + 1] * keep_scale : 0.f;
          }
        }
    }
    // probabilities back into accumulator layout (packed: pw[i][pr] = {block 2 pr: keys +0..3, block 2 pr + 1: keys +0..3}), row dots
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float dot = 0.f;
#pragma unroll
      for (int pr = 0; pr < 3; ++pr) {
        uint32_t x[2] = {pw[i][pr][0], pw[i][pr][1]}, y[2] = {pw[i][pr][2], pw[i][pr][3]};
        pair_swap(x, y);
        pw[i][pr] = u32x4{x[0], x[1], y[0], y[1]};
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const uint32_t w0 = pw[i][pr][2 * b], w1 = pw[i][pr][2 * b + 1];
          dot += acc[i][2 * pr + b][0] * bf_lo(w0) + acc[i][2 * pr + b][1] * bf_hi(w0) + acc[i][2 * pr + b][2] * bf_lo(w1) +
                 acc[i][2 * pr + b][3] * bf_hi(w1);
        }
      }
      dot += __shfl_xor(dot, 16, 64);
      dot += __shfl_xor(dot, 32, 64);
      if (lg == 0) stats[(64 * wr + 16 * i + lr) * 4 + wc] = dot;
    }
    __syncthreads();
    SC_STAMP(10)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(stats + (64 * wr + 16 * i + lr) * 4);
      const float dot = (d4[0] + d4[1]) + (d4[2] + d4[3]);
#pragma unroll
      for (int pr = 0; pr < 3; ++pr) {
        float v[8];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const uint32_t w0 = pw[i][pr][2 * b], w1 = pw[i][pr][2 * b + 1];
          v[4 * b + 0] = bf_lo(w0) * (acc[i][2 * pr + b][0] - dot);
          v[4 * b + 1] = bf_hi(w0) * (acc[i][2 * pr + b][1] - dot);
          v[4 * b + 2] = bf_lo(w1) * (acc[i][2 * pr + b][2] - dot);
          v[4 * b + 3] = bf_hi(w1) * (acc[i][2 * pr + b][3] - dot);
        }
        uint32_t x[2] = {f32x2_to_bf16x2(v[0], v[1]), f32x2_to_bf16x2(v[2], v[3])};
        uint32_t y[2] = {f32x2_to_bf16x2(v[4], v[5]), f32x2_to_bf16x2(v[6], v[7])};
        pair_swap(x, y);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{x[0], x[1], y[0], y[1]}, rs, mat_off(i, pr), 0, SC_NT_ST);
      }
    }
  }
  SC_STAMP(11)
  cur = nxt;  // (the statistics words are next written behind the K loop's barriers)
  }
#undef SC_ISSUE
}

template <bool BWD>
int launch(const Args& a, hipStream_t s) {
  static bool attr = false;  // (once per instantiation: the attribute call is not free on the launch path)
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&scores_kernel<BWD>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) !=
        hipSuccess)
      return case_set_error(CASE_E_LAUNCH, "case_attention_scores: cannot raise the dynamic LDS limit");
    attr = true;
  }
  const int cus = case_persistent_cus();  // whole XCD rounds; one workgroup per CU (132 KiB of LDS each)
  hipLaunchKernelGGL((scores_kernel<BWD>), dim3(a.nblk < cus ? a.nblk : cus), dim3(NTHR), LDS_BYTES, s, a);
  return case_check_launch(BWD ? "case_attention_scores_bwd" : "case_attention_scores_fwd");
}

// =====================================================================================================================================
// The four products AROUND the probabilities (O = Pd V, dQ = alpha dS K, dV = Pd^T dO, dK = alpha dS^T Q) on the same skeleton: a
// workgroup owns 128 output rows x all 320 head-dim columns of one (sequence, head), the [L, L] operand is read once, the 320-wide
// operand (V / K / dO / Q: memory rows = the contraction index) comes in k-major pieces of [8 k-rows][64 columns] whose chunks are
// swizzled on the source side (chunk c of k-row j of k-block kb at slot c ^ 2 f, f = ((j >> 1) & 1) + 2 (kb & 1): the scheme of
// gemm8w.inc) and is read by ds_read_b64_tr_b16 pairs.  AK: the [L, L] operand is used transposed (dV, dK: output rows = keys,
// contraction over the queries), so it is k-major as well.  Waves 2 (row halves) x 4 (80 columns = 5 blocks each), C^T accumulator
// orientation (a lane holds 4 consecutive columns of one row), 16-byte stores through pair_swap (8-byte for the odd fifth block).
// =====================================================================================================================================
constexpr int GN = 320, G_BUNITS = 3, G_STAGE = A_BYTES + G_BUNITS * 16384, G_LDS = 2 * G_STAGE;

struct GArgs {
  const bf16_t* a;   // the [L, L] matrices, [N, heads, La, lda]: AK = false rows = output rows, AK = true rows = contraction index
  const bf16_t* b;   // [.., Kc rows, ldb] with the head's 320 columns at b + n sb1 + head sb2
  bf16_t* c;         // output rows at c + n sc1 + head sc2 + row ldc
  int64_t lda, ldb, ldc, sa1, sa2, sb1, sb2, sc1, sc2;
  int M, Kc, heads, tiles, nblk;
  float alpha;
};

typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ bf16x8 frag_km(const char* p) {  // 8 consecutive k of one column: k-rows j and j + 4 of a piece
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 512));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

template <bool AK>
__global__ __launch_bounds__(NTHR) void rc_gemm_kernel(const GArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, l = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3, lr = l & 15, lg = l >> 4;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nwx = ((int)gridDim.x - xcd + 7) >> 3;
  const int per = g.nblk >> 3, rem = g.nblk & 7;
  const int first = xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per, count = per + (xcd < rem ? 1 : 0);
  struct Item { int n, head, m0, rows; i32x4 rsa, rsb; };
  auto item_of = [&](int idx) {
    Item t;
    const int pid = first + idx;
    const int mt = pid % g.tiles;
    t.head = (pid / g.tiles) % g.heads;
    t.n = pid / (g.tiles * g.heads);
    t.m0 = mt * TM;
    t.rows = min(TM, g.M - t.m0);
    const bf16_t* Ab = g.a + (int64_t)t.n * g.sa1 + (int64_t)t.head * g.sa2;
    if constexpr (AK) {  // columns m0 .. of every k-row
      t.rsa = make_rsrc(Ab + t.m0, (uint32_t)(((int64_t)(g.Kc - 1) * g.lda + t.rows) * 2));
    } else {
      t.rsa = make_rsrc(Ab + (int64_t)t.m0 * g.lda, (uint32_t)(((int64_t)(t.rows - 1) * g.lda + g.Kc) * 2));
    }
    t.rsb = make_rsrc(g.b + (int64_t)t.n * g.sb1 + (int64_t)t.head * g.sb2, (uint32_t)(((int64_t)(g.Kc - 1) * g.ldb + GN) * 2));
    return t;
  };
  if (slot >= count) return;

  // DMA lane offsets.  k-contiguous A: as in scores_kernel (two 8-row groups per wave).  k-major operands: the wave owns k-block `wave`
  // of the K step; lane l -> k-row j = l >> 3, slot l & 7 <- chunk (l & 7) ^ 2 f; the 64-column pieces of a k-block differ by 128 bytes.
  const int kj = l >> 3, kf = ((kj >> 1) & 1) + 2 * (wave & 1);
  unsigned va[2];
  if constexpr (AK) {
    va[0] = (unsigned)((wave * 8 + kj) * g.lda * 2 + (((l & 7) ^ (2 * kf)) << 4));
    va[1] = va[0] + 128u;
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = 8 * (wave * 2 + i) + (l >> 3);
      va[i] = (unsigned)(row * g.lda * 2 + (((l & 7) ^ ((row >> 1) & 7)) << 4));
    }
  }
  const unsigned vbk = (unsigned)((wave * 8 + kj) * g.ldb * 2 + (((l & 7) ^ (2 * kf)) << 4));
  const unsigned a_step = AK ? (unsigned)(64 * g.lda * 2) : 128u, b_step = (unsigned)(64 * g.ldb * 2);
#define RG_ISSUE(T, ST, KT)                                                                                              \
  {                                                                                                                      \
    const unsigned sb_ = lds0 + (ST) * G_STAGE, sa_ = (unsigned)(KT) * a_step, so_ = (unsigned)(KT) * b_step;            \
    if constexpr (AK) {                                                                                                  \
      dma16<false>((T).rsa, va[0], sa_, sb_ + wave * 1024);                                                              \
      dma16<false>((T).rsa, va[1], sa_, sb_ + 8192 + wave * 1024);                                                       \
    } else {                                                                                                             \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) dma16<SC_NT_A>((T).rsa, va[i], sa_, sb_ + (wave * 2 + i) * 1024);    \
    }                                                                                                                    \
    _Pragma("unroll") for (int p = 0; p < 5; ++p)  /* pieces: unit p >> 1, column block p & 1 (columns 64 p ..) */         \
        dma16<false>((T).rsb, vbk, so_ + p * 128u, sb_ + A_BYTES + (p >> 1) * 16384 + (p & 1) * 8192 + wave * 1024);     \
  }
  // fragment lane bases.  k-major block x (16 columns) of a piece pair: see gemm8w.inc Op<true>::init
  const int tg = lg, tq = lr >> 2, tt = lr & 3, tf = (tq >> 1) + 2 * (tg & 1);
  auto km_base = [&](int x) { return tg * 1024 + tq * 128 + ((((x ^ tf) * 2) + (tt >> 1)) << 4) + (tt & 1) * 8; };
  int boff[5];  // the wave's column blocks 5 wc + jj: unit (c >> 3), column block ((c >> 2) & 1), sub-block c & 3
#pragma unroll
  for (int jj = 0; jj < 5; ++jj) {
    const int c = 5 * wc + jj;
    boff[jj] = A_BYTES + (c >> 3) * 16384 + ((c >> 2) & 1) * 8192 + km_base(c & 3);
  }
  int aoff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) aoff[i] = AK ? wr * 8192 + km_base(i) : wr * (64 * 128) + i * 2048 + lr * 128;
  const int fb0 = (lg ^ (lr >> 1)) << 4, fb1 = ((lg + 4) ^ (lr >> 1)) << 4;  // k-contiguous A: chunk slots of the two k halves
  const int nk = g.Kc / BKE;
  const int gq = ((lg & 1) << 1) | (lg >> 1);

  Item cur = item_of(slot);
  RG_ISSUE(cur, 0, 0)
  RG_ISSUE(cur, 1, 1)
  int gs = 0;
  for (int idx = slot; idx < count; idx += nwx) {
    const bool first_item = idx == slot, has_nxt = idx + nwx < count;
    Item nxt = cur;
    if (has_nxt) nxt = item_of(idx + nwx);
    f32x4 acc[4][5];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kt = 0; kt < nk; ++kt, ++gs) {
      // as in scores_kernel: 7 DMAs per wave and K step; 12 result stores per lane and item
      if (kt == 0 && !first_item) __builtin_amdgcn_s_waitcnt(0x4f73);      // vmcnt(19)
      else if (kt + 1 < nk || has_nxt) __builtin_amdgcn_s_waitcnt(0x0f77); // vmcnt(7)
      else __builtin_amdgcn_s_waitcnt(0x0f70);                             // vmcnt(0)
      __syncthreads();
      const char* S = smem + (gs & 1) * G_STAGE;
      bf16x8 af[2][4], bfr[2][5];
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if constexpr (AK) af[kh][i] = frag_km(S + kh * 4096 + aoff[i]);
          else af[kh][i] = *reinterpret_cast<const bf16x8*>(S + aoff[i] + (kh ? fb1 : fb0));
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) bfr[kh][j] = frag_km(S + kh * 4096 + boff[j]);
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
      __syncthreads();
      if (kt + 2 < nk) RG_ISSUE(cur, gs & 1, kt + 2)
      else if (has_nxt) RG_ISSUE(nxt, gs & 1, kt + 2 - nk)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[kh][j], af[kh][i], acc[i][j], 0, 0, 0);
    }

    // ---- epilogue: acc[i][j][e] = C[m0 + 64 wr + 16 i + lr][80 wc + 16 j + 4 lg + e] -------------------------------------------------------
    bf16_t* Ct = g.c + (int64_t)cur.n * g.sc1 + (int64_t)cur.head * g.sc2 + (int64_t)cur.m0 * g.ldc;
    const __amdgpu_buffer_rsrc_t rc = as_rsrc(Ct, (uint32_t)(((int64_t)(cur.rows - 1) * g.ldc + GN) * 2));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rowb = (64 * wr + 16 * i + lr) * (int)g.ldc;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        uint32_t x[2] = {f32x2_to_bf16x2(acc[i][2 * pr][0] * g.alpha, acc[i][2 * pr][1] * g.alpha),
                         f32x2_to_bf16x2(acc[i][2 * pr][2] * g.alpha, acc[i][2 * pr][3] * g.alpha)};
        uint32_t y[2] = {f32x2_to_bf16x2(acc[i][2 * pr + 1][0] * g.alpha, acc[i][2 * pr + 1][1] * g.alpha),
                         f32x2_to_bf16x2(acc[i][2 * pr + 1][2] * g.alpha, acc[i][2 * pr + 1][3] * g.alpha)};
        pair_swap(x, y);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{x[0], x[1], y[0], y[1]}, rc, (rowb + 80 * wc + 32 * pr + 8 * gq) * 2, 0, 0);
      }
      typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
      const u32x2 z = {f32x2_to_bf16x2(acc[i][4][0] * g.alpha, acc[i][4][1] * g.alpha),
                       f32x2_to_bf16x2(acc[i][4][2] * g.alpha, acc[i][4][3] * g.alpha)};
      __builtin_amdgcn_raw_buffer_store_b64(z, rc, (rowb + 80 * wc + 64 + 4 * lg) * 2, 0, 0);
    }
    cur = nxt;
  }
#undef RG_ISSUE
}

template <bool AK>
int launch_rc(const GArgs& a, hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&rc_gemm_kernel<AK>), hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS) != hipSuccess)
      return case_set_error(CASE_E_LAUNCH, "case_attention_product: cannot raise the dynamic LDS limit");
    attr = true;
  }
  const int cus = case_persistent_cus();
  hipLaunchKernelGGL((rc_gemm_kernel<AK>), dim3(a.nblk < cus ? a.nblk : cus), dim3(NTHR), G_LDS, s, a);
  return case_check_launch("case_attention_product");
}

#ifdef SC_STAMPS
int read_stamps(uint64_t* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sc_stamps), sizeof(g_sc_stamps)) == hipSuccess ? 0 : -1; }
#endif

bool shape_ok(const CaseAttnDesc* d) {
  return d->head_dim >= 2 * BKE && d->head_dim % BKE == 0 && d->Lk > 0 && d->Lk <= TN && d->Lk % 8 == 0 && d->Lq > 0 && !d->causal;
}

}  // namespace attn_sc

#ifdef SC_STAMPS
extern "C" int case_attention_scores_stamps(uint64_t* out) { return attn_sc::read_stamps(out); }
#endif

extern "C" int case_attention_scores_supported(const CaseAttnDesc* d) { return d && attn_sc::shape_ok(d) ? 1 : 0; }

#define CASE_SCORES_CHECKS(NAME, A, LDA, SA, B, LDB, SB)                                                                              \
  CASE_REQUIRE(attn_sc::shape_ok(d), NAME ": needs head_dim %% 64 == 0 and >= 128, Lk <= 384, Lk %% 8 == 0, no causal mask");                     \
  CASE_REQUIRE(d->N > 0 && d->heads > 0 && d->N * d->heads * ((d->Lq + 127) / 128) < (1ll << 31), NAME ": bad batch geometry");       \
  CASE_REQUIRE((LDA) % 8 == 0 && (LDB) % 8 == 0 && (SA) % 8 == 0 && (SB) % 8 == 0 && (uintptr_t)(A) % 16 == 0 &&                      \
                   (uintptr_t)(B) % 16 == 0,                                                                                           \
               NAME ": operands must be 16-byte aligned with strides that are multiples of 8 elements");                              \
  CASE_REQUIRE(((d->Lq - 1) * (LDA) + d->head_dim) * 2 < (1ll << 31) && ((d->Lk - 1) * (LDB) + d->head_dim) * 2 < (1ll << 31) &&      \
                   (int64_t)128 * d->Lk * 2 < (1ll << 31),                                                                             \
               NAME ": a sequence does not fit 32-bit buffer offsets");                                                                \
  CASE_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f, NAME ": drop_p out of range")

static void fill_scores(attn_sc::Args& a, const CaseAttnDesc* d) {
  a.N = (int)d->N; a.heads = (int)d->heads; a.Lq = (int)d->Lq; a.Lk = (int)d->Lk; a.d = (int)d->head_dim;
  a.tiles = (int)((d->Lq + 127) / 128);
  a.nblk = a.tiles * a.heads * a.N;
  a.alpha = d->scale; a.drop_p = d->drop_p; a.seed = d->seed; a.offset = d->offset; a.state = d->state;
}

extern "C" int case_attention_scores_fwd(const CaseAttnDesc* d, const void* q, const void* k, const uint8_t* key_valid, void* p,
                                         void* p_dropped, case_stream_t stream) {
  CASE_REQUIRE(d && q && k && p, "case_attention_scores_fwd: null argument");
  CASE_SCORES_CHECKS("case_attention_scores_fwd", q, d->ldq, d->sq, k, d->ldk, d->sk);
  CASE_REQUIRE((d->drop_p > 0.f) == (p_dropped != nullptr), "case_attention_scores_fwd: p_dropped goes with drop_p > 0");
  CASE_REQUIRE((uintptr_t)p % 16 == 0 && (uintptr_t)p_dropped % 16 == 0, "case_attention_scores_fwd: outputs must be 16-byte aligned");
  attn_sc::Args a = {};
  fill_scores(a, d);
  a.a = (const bf16_t*)q; a.b = (const bf16_t*)k; a.lda = d->ldq; a.ldb = d->ldk; a.sa = d->sq; a.sb = d->sk;
  a.key_valid = key_valid; a.out0 = (bf16_t*)p; a.out1 = (bf16_t*)p_dropped;
  return attn_sc::launch<false>(a, (hipStream_t)stream);
}

extern "C" int case_attention_scores_bwd(const CaseAttnDesc* d, const void* dout, const void* v, const void* p, void* ds,
                                         case_stream_t stream) {
  CASE_REQUIRE(d && dout && v && p && ds, "case_attention_scores_bwd: null argument");
  CASE_SCORES_CHECKS("case_attention_scores_bwd", dout, d->ldo, d->so, v, d->ldv, d->sv);
  CASE_REQUIRE((uintptr_t)p % 16 == 0 && (uintptr_t)ds % 16 == 0, "case_attention_scores_bwd: P / dS must be 16-byte aligned");
  attn_sc::Args a = {};
  fill_scores(a, d);
  a.a = (const bf16_t*)dout; a.b = (const bf16_t*)v; a.lda = d->ldo; a.ldb = d->ldv; a.sa = d->so; a.sb = d->sv;
  a.p_in = (const bf16_t*)p; a.out0 = (bf16_t*)ds;
  return attn_sc::launch<true>(a, (hipStream_t)stream);
}

// ---- the products around the probabilities -------------------------------------------------------------------------------------------------
static bool product_ok(const CaseAttnProductDesc* d) {
  return d->head_dim == attn_sc::GN && d->Kc >= 2 * attn_sc::BKE && d->Kc % attn_sc::BKE == 0 && d->M > 0 && d->N > 0 && d->heads > 0;
}
extern "C" int case_attention_product_supported(const CaseAttnProductDesc* d) { return d && product_ok(d) ? 1 : 0; }

extern "C" int case_attention_product(const CaseAttnProductDesc* d, const void* a, const void* b, void* c, case_stream_t stream) {
  CASE_REQUIRE(d && a && b && c, "case_attention_product: null argument");
  CASE_REQUIRE(product_ok(d), "case_attention_product: needs head_dim 320 and a contraction length that is a multiple of 64, >= 128");
  CASE_REQUIRE(d->N * d->heads * ((d->M + 127) / 128) < (1ll << 31), "case_attention_product: bad batch geometry");
  CASE_REQUIRE(d->lda % 8 == 0 && d->ldb % 8 == 0 && d->ldc % 8 == 0 && d->sa_seq % 8 == 0 && d->sa_head % 8 == 0 && d->sb_seq % 8 == 0 &&
                   d->sb_head % 8 == 0 && d->sc_seq % 8 == 0 && d->sc_head % 8 == 0 && (uintptr_t)a % 16 == 0 && (uintptr_t)b % 16 == 0 &&
                   (uintptr_t)c % 16 == 0,
               "case_attention_product: operands must be 16-byte aligned with strides that are multiples of 8 elements");
  CASE_REQUIRE(d->lda >= (d->a_transposed ? d->M : d->Kc), "case_attention_product: lda shorter than a row of a");
  const int64_t a_rows = d->a_transposed ? d->Kc : d->M;
  CASE_REQUIRE(a_rows * d->lda * 2 < (1ll << 31) && d->Kc * d->ldb * 2 < (1ll << 31) && (d->M + 127) * d->ldc * 2 < (1ll << 31),
               "case_attention_product: a sequence does not fit 32-bit buffer offsets");
  attn_sc::GArgs g = {};
  g.a = (const bf16_t*)a; g.b = (const bf16_t*)b; g.c = (bf16_t*)c;
  g.lda = d->lda; g.ldb = d->ldb; g.ldc = d->ldc;
  g.sa1 = d->sa_seq; g.sa2 = d->sa_head; g.sb1 = d->sb_seq; g.sb2 = d->sb_head; g.sc1 = d->sc_seq; g.sc2 = d->sc_head;
  g.M = (int)d->M; g.Kc = (int)d->Kc; g.heads = (int)d->heads;
  g.tiles = (int)((d->M + 127) / 128);
  g.nblk = g.tiles * g.heads * (int)d->N;
  g.alpha = d->alpha;
  return d->a_transposed ? attn_sc::launch_rc<true>(g, (hipStream_t)stream) : attn_sc::launch_rc<false>(g, (hipStream_t)stream);
}
