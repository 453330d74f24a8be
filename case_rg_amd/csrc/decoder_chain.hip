// K20  the row-local stages of a decoder layer's GREEDY STEP as one launch (inference; reference: common/TransformerDecoder.py:76-89 run on
// one new position per sequence, CaSE/Model.py:94-123):
//
//     x = LN1(x); x += SelfAttn(x); x = LN2(x); x += CrossAttn(x, memory); x = LN3(x); x += W2 gelu(W1 x)
//
// Between the two attention cores everything is local to a row of 512 features, and at one row per sequence (M = batch, 256 at cfg 4)
// every op is a 10-microsecond launch whose time is one memory round trip: 13 launches per layer and step.  Here the chain between two
// attention cores is ONE kernel -- stages, all optional, in this order:
//
//     S1  y = x Wp^T + bp + resid            (out-projection of the attention core that ran before, + the residual)
//     S2  y = LN_a(y)
//     S3  y = gelu(y W1^T + b1) W2^T + b2 + y (the feed-forward pair; its residual is S2's output)          -> o_out
//     S4  y = LN_b(y)                                                                                       -> n_out (last LN's output)
//     S5  q | q, k, v = y Wqkv^T + bqkv      (the in-projection of the attention core that runs next)       -> q_out, kv_out
//
// used as  [S4 S5]           first layer:  LN1 -> QKV (k, v straight into position t of the layer's cache),
//          [S1 S2 S5(q)]     out-proj + residual -> LN2 -> cross-attention query,
//          [S1 S2 S3 S4 S5]  out-proj + residual -> LN3 -> FFN -> the NEXT layer's LN1 -> QKV   (last layer: [S1 S2 S3], o_out).
// A layer-step is then 2 attention launches + 2 of these (was 13).
//
// A workgroup of eight waves owns 16 rows (one MFMA column block); the rows live in LDS as bf16 (three 16 KiB buffers, 16-byte chunks
// XOR-swizzled by the row: conflict-free fragment reads).  v_mfma_f32_16x16x32_bf16 with the FEATURES on the MFMA rows: a wave owns 64 of
// a stage's 512 output features, a lane four consecutive features of one row per 16-feature block, so bias / residual / GELU /
// LayerNorm / packing run in registers.  Weights are the ordinary row-major bf16 operand copies [N, 512], streamed from L2 straight
// into MFMA A fragments (16 bytes per lane), four K steps ahead.  Every intermediate is rounded to bf16 exactly where the single
// launches round it (GEMM output, LayerNorm output, GELU output), LayerNorm is the same two-pass form: results agree with the unfused
// step to the summation order of the K loop.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace dec_chain {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

constexpr int E = 512, ROWS = 16, NW = 8, NTHR = NW * 64, BUF = ROWS * E * 2;
constexpr int STAT_OFF = 3 * BUF, LDS_B = STAT_OFF + ROWS * NW * 4;

struct Args {
  const bf16_t* x_in; const bf16_t* resid;
  const bf16_t* wp; const float* bp;
  const float* ga; const float* ba;
  const bf16_t* w1; const float* b1; const bf16_t* w2; const float* b2;
  const float* gb; const float* bb;
  const bf16_t* wqkv; const float* bqkv;
  bf16_t* n_out; bf16_t* q_out; bf16_t* kv_out; bf16_t* o_out;
  int64_t M, kv_stride;
  int qkv_parts;  // 1: query only, 3: q, k, v
  float eps_a, eps_b;
};

__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
// byte offset of 16-byte chunk c (0..63) of row t (0..15) inside a row buffer
__device__ __forceinline__ int xoff(int t, int c) { return t * 1024 + ((c ^ t) << 4); }

// rows [row0, row0 + 16) of a [M, 512] bf16 tensor -> a row buffer (zeros beyond M)
__device__ __forceinline__ void load_rows(const bf16_t* src, char* buf, int64_t row0, int64_t M, int tid) {
  const int t = tid >> 5, c0 = (tid & 31) * 2;
  u32x4 v0 = {0u, 0u, 0u, 0u}, v1 = v0;
  if (row0 + t < M) {
    const u32x4* p = reinterpret_cast<const u32x4*>(src + (row0 + t) * E) + c0;
    v0 = p[0];
    v1 = p[1];
  }
  *reinterpret_cast<u32x4*>(buf + xoff(t, c0)) = v0;
  *reinterpret_cast<u32x4*>(buf + xoff(t, c0 + 1)) = v1;
}

// acc[j][e] += sum_k W[n0 + 16 j + 4 g + e][k] * X[t][k]   (lane: t = l & 15, g = l >> 4; W points at the stage's first feature row)
__device__ __forceinline__ void gemm_rows(const bf16_t* __restrict__ W, const char* xbuf, int wave, int l, f32x4 (&acc)[4]) {
  const int t = l & 15, g = l >> 4;
  const bf16_t* wp = W + (int64_t)(64 * wave + t) * E + g * 8;
  u32x4 wf[2][4][4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int j = 0; j < 4; ++j) wf[0][s][j] = *reinterpret_cast<const u32x4*>(wp + j * 16 * E + s * 32);
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    if (kb + 1 < 4) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[(kb + 1) & 1][s][j] = *reinterpret_cast<const u32x4*>(wp + j * 16 * E + ((kb + 1) * 4 + s) * 32);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xbuf + xoff(t, 4 * (kb * 4 + s) + g));
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&wf[kb & 1][s][j]), xf, acc[j], 0, 0, 0);
    }
  }
}

// the lane's 16 values (4 blocks x 4 features of row t) <-> a row buffer
__device__ __forceinline__ void read_own(const char* buf, int wave, int l, float (&v)[4][4]) {
  const int t = l & 15, g = l >> 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int f0 = 64 * wave + 16 * j + 4 * g;
    const u32x2 w = *reinterpret_cast<const u32x2*>(buf + xoff(t, f0 >> 3) + ((f0 >> 2) & 1) * 8);
    v[j][0] = bf_lo(w[0]); v[j][1] = bf_hi(w[0]); v[j][2] = bf_lo(w[1]); v[j][3] = bf_hi(w[1]);
  }
}
__device__ __forceinline__ void round_own(float (&v)[4][4], u32x2 (&pk)[4]) {  // to bf16 and back: the value the next op sees
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    pk[j][0] = f32x2_to_bf16x2(v[j][0], v[j][1]);
    pk[j][1] = f32x2_to_bf16x2(v[j][2], v[j][3]);
    v[j][0] = bf_lo(pk[j][0]); v[j][1] = bf_hi(pk[j][0]); v[j][2] = bf_lo(pk[j][1]); v[j][3] = bf_hi(pk[j][1]);
  }
}
__device__ __forceinline__ void write_own(char* buf, int wave, int l, const u32x2 (&pk)[4]) {
  const int t = l & 15, g = l >> 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int f0 = 64 * wave + 16 * j + 4 * g;
    *reinterpret_cast<u32x2*>(buf + xoff(t, f0 >> 3) + ((f0 >> 2) & 1) * 8) = pk[j];
  }
}
// 8-byte stores of the lane's features of row row0 + t into a [*, ld] bf16 tensor (column offset col0)
__device__ __forceinline__ void store_own(bf16_t* dst, int64_t ld, int64_t row0, int64_t M, int col0, int wave, int l, const u32x2 (&pk)[4]) {
  const int t = l & 15, g = l >> 4;
  if (row0 + t >= M) return;
#pragma unroll
  for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x2*>(dst + (row0 + t) * ld + col0 + 64 * wave + 16 * j + 4 * g) = pk[j];
}

// sum over the 512 features of each row of a per-lane partial (the lane's 16 features): lane groups by shuffles, waves through LDS
__device__ __forceinline__ float row_total(float part, float* stat, int wave, int l) {
  part += __shfl_xor(part, 16);
  part += __shfl_xor(part, 32);
  const int t = l & 15;
  __syncthreads();  // the previous exchange has been read
  if (l < 16) stat[t * NW + wave] = part;
  __syncthreads();
  const f32x4 a = *reinterpret_cast<const f32x4*>(stat + t * NW), b = *reinterpret_cast<const f32x4*>(stat + t * NW + 4);
  return ((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]));
}

// LayerNorm over the row, two passes like ln_fwd_vec (rowops.hip); v holds bf16-representable inputs, returns the normalised values
__device__ __forceinline__ void layer_norm(float (&v)[4][4], const float* __restrict__ gam, const float* __restrict__ bet, float eps, float* stat,
                                           int wave, int l) {
  const int g = l >> 4;
  f32x4 gm[4], bt[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    gm[j] = *reinterpret_cast<const f32x4*>(gam + 64 * wave + 16 * j + 4 * g);
    bt[j] = *reinterpret_cast<const f32x4*>(bet + 64 * wave + 16 * j + 4 * g);
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) s += v[j][e];
  const float mean = row_total(s, stat, wave, l) * (1.f / E);
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[j][e] -= mean;
      q += v[j][e] * v[j][e];
    }
  const float rstd = rsqrtf(row_total(q, stat, wave, l) * (1.f / E) + eps);
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) v[j][e] = v[j][e] * rstd * gm[j][e] + bt[j][e];
}

__device__ __forceinline__ void add_bias(f32x4 (&acc)[4], const float* __restrict__ bias, int wave, int l, float (&v)[4][4]) {
  const int g = l >> 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(bias + 64 * wave + 16 * j + 4 * g);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[j][e] = acc[j][e] + b[e];
  }
}

__global__ __launch_bounds__(NTHR) void chain_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, l = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* stat = reinterpret_cast<float*>(smem + STAT_OFF);
  char* b0 = smem;            // the current GEMM input
  char* b1 = smem + BUF;      // the residual rows / second operand buffer
  char* b2 = smem + 2 * BUF;  // third buffer
  const int64_t row0 = (int64_t)blockIdx.x * ROWS;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

  load_rows(a.x_in, b0, row0, a.M, tid);
  if (a.resid) load_rows(a.resid, b1, row0, a.M, tid);
  __syncthreads();

  float v[4][4];
  u32x2 pk[4];
  char* cur = b0;  // the buffer that holds the current rows
  if (a.wp) {  // S1
    f32x4 acc[4] = {zero, zero, zero, zero};
    gemm_rows(a.wp, b0, wave, l, acc);
    add_bias(acc, a.bp, wave, l, v);
    if (a.resid) {
      float r[4][4];
      read_own(b1, wave, l, r);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[j][e] += r[j][e];
    }
    round_own(v, pk);
  } else {
    read_own(b0, wave, l, v);
  }
  if (a.ga) {  // S2 -> b2 (every wave is past its reads of b0 / b1 after the exchanges inside)
    layer_norm(v, a.ga, a.ba, a.eps_a, stat, wave, l);
    round_own(v, pk);
    write_own(b2, wave, l, pk);
    cur = b2;
    if (!a.w1 && !a.gb && a.n_out) store_own(a.n_out, E, row0, a.M, 0, wave, l, pk);
    __syncthreads();
  }
  if (a.w1) {  // S3: b2 -> gelu -> b0 -> + residual (the lane's own values of b2, still in v)
    f32x4 acc[4] = {zero, zero, zero, zero};
    gemm_rows(a.w1, cur, wave, l, acc);
    float h[4][4];
    add_bias(acc, a.b1, wave, l, h);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) h[j][e] = gelu_f(h[j][e]);
    u32x2 hp[4];
    round_own(h, hp);
    write_own(b0, wave, l, hp);  // b0's last readers (S1) are behind the barriers of S2
    __syncthreads();
    f32x4 acc2[4] = {zero, zero, zero, zero};
    gemm_rows(a.w2, b0, wave, l, acc2);
    float o[4][4];
    add_bias(acc2, a.b2, wave, l, o);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[j][e] += o[j][e];
    round_own(v, pk);
    if (a.o_out) store_own(a.o_out, E, row0, a.M, 0, wave, l, pk);
  }
  if (a.gb) {  // S4 -> b1
    layer_norm(v, a.gb, a.bb, a.eps_b, stat, wave, l);
    round_own(v, pk);
    write_own(b1, wave, l, pk);
    cur = b1;
    if (a.n_out) store_own(a.n_out, E, row0, a.M, 0, wave, l, pk);
    __syncthreads();
  }
  if (a.wqkv) {  // S5
    for (int part = 0; part < a.qkv_parts; ++part) {
      f32x4 acc[4] = {zero, zero, zero, zero};
      gemm_rows(a.wqkv + (int64_t)part * E * E, cur, wave, l, acc);
      float o[4][4];
      add_bias(acc, a.bqkv + part * E, wave, l, o);
      u32x2 op[4];
      round_own(o, op);
      if (part == 0) store_own(a.q_out, E, row0, a.M, 0, wave, l, op);
      else store_own(a.kv_out, a.kv_stride, row0, a.M, (part - 1) * E, wave, l, op);
    }
  }
}

}  // namespace dec_chain

extern "C" int case_decoder_chain(const CaseDecoderChainDesc* d, const void* x_in, const void* resid, const void* w_proj, const float* b_proj,
                                  const float* ln_a_g, const float* ln_a_b, const void* w1, const float* b1, const void* w2, const float* b2,
                                  const float* ln_b_g, const float* ln_b_b, const void* w_qkv, const float* b_qkv, void* n_out, void* q_out,
                                  void* kv_out, void* o_out, case_stream_t stream) {
  CASE_REQUIRE(d && x_in, "case_decoder_chain: null argument");
  CASE_REQUIRE(d->width == dec_chain::E, "case_decoder_chain: built for d_model = dim_feedforward = 512 (got %d)", (int)d->width);
  CASE_REQUIRE(d->rows > 0 && d->rows < (1ll << 31) - 16, "case_decoder_chain: bad row count");
  CASE_REQUIRE(!w_proj || b_proj, "case_decoder_chain: the projection stage needs its bias");
  CASE_REQUIRE(!resid || w_proj, "case_decoder_chain: a residual without the projection stage");
  CASE_REQUIRE((ln_a_g != nullptr) == (ln_a_b != nullptr) && (ln_b_g != nullptr) == (ln_b_b != nullptr), "case_decoder_chain: LayerNorm needs weight and bias");
  CASE_REQUIRE((w1 != nullptr) == (w2 != nullptr) && (!w1 || (b1 && b2 && ln_a_g)), "case_decoder_chain: the feed-forward stage needs W1, b1, W2, b2 and LN_a in front");
  CASE_REQUIRE(!w_qkv || (b_qkv && q_out && (d->qkv_parts == 1 || (d->qkv_parts == 3 && kv_out && d->kv_row_stride >= 2 * dec_chain::E))),
               "case_decoder_chain: the in-projection stage needs its bias, q_out and (qkv_parts = 3) kv_out with kv_row_stride >= 1024");
  CASE_REQUIRE(w_qkv || o_out || n_out, "case_decoder_chain: nothing to write");
  CASE_REQUIRE(!o_out || w1, "case_decoder_chain: o_out is the feed-forward stage's output");
  CASE_REQUIRE(d->kv_row_stride % 4 == 0, "case_decoder_chain: kv_row_stride must be a multiple of 4 elements");
  for (const void* p : {x_in, resid, w_proj, w1, w2, w_qkv, (const void*)n_out, (const void*)q_out, (const void*)o_out, (const void*)b_proj, (const void*)b1,
                        (const void*)b2, (const void*)b_qkv, (const void*)ln_a_g, (const void*)ln_a_b, (const void*)ln_b_g, (const void*)ln_b_b})
    CASE_REQUIRE((reinterpret_cast<uintptr_t>(p) & 15) == 0, "case_decoder_chain: operands must be 16-byte aligned");
  CASE_REQUIRE((reinterpret_cast<uintptr_t>(kv_out) & 7) == 0, "case_decoder_chain: kv_out must be 8-byte aligned");
  dec_chain::Args a;
  a.x_in = (const bf16_t*)x_in; a.resid = (const bf16_t*)resid;
  a.wp = (const bf16_t*)w_proj; a.bp = b_proj;
  a.ga = ln_a_g; a.ba = ln_a_b;
  a.w1 = (const bf16_t*)w1; a.b1 = b1; a.w2 = (const bf16_t*)w2; a.b2 = b2;
  a.gb = ln_b_g; a.bb = ln_b_b;
  a.wqkv = (const bf16_t*)w_qkv; a.bqkv = b_qkv;
  a.n_out = (bf16_t*)n_out; a.q_out = (bf16_t*)q_out; a.kv_out = (bf16_t*)kv_out; a.o_out = (bf16_t*)o_out;
  a.M = d->rows; a.kv_stride = d->kv_row_stride; a.qkv_parts = d->qkv_parts;
  a.eps_a = d->eps_a; a.eps_b = d->eps_b;
  const unsigned grid = (unsigned)((d->rows + dec_chain::ROWS - 1) / dec_chain::ROWS);
  hipLaunchKernelGGL(dec_chain::chain_kernel, dim3(grid), dim3(dec_chain::NTHR), dec_chain::LDS_B, (hipStream_t)stream, a);
  return case_check_launch("case_decoder_chain");
}
