// K21 (round 5): the greedy step's cross-attention over the passage memory with ABSORBED projections.
//
// The cached form (K13, attn_decode64_kernel) streams, per decoder layer and step, the layer's own K and V projections of the memory:
// 2 x S x 512 x 2 B per item -- 8 streams of 3.9 MB for the four layers of the passage stack at S = 3840 (common/TransformerDecoder.py:81-82
// evaluated at one position, CaSE/Model.py:94-123).  Algebraically
//     q_h . K_h[j] = q_h . (Wk_h mem_j + bk_h) = (Wk_h^T q_h) . mem_j + const_j-free      (the constant drops out of the softmax)
//     sum_j p_hj V_h[j] = Wv_h (sum_j p_hj mem_j) + bv_h
// so every layer can attend the RAW memory rows with a 512-wide query per head (qp_h = Wk_h^T q_h, a [E -> heads x E] linear map of the
// layer input, precomputed weights) and project the 512-wide context per head afterwards: ONE stream of S x 512 x 2 B per layer and item
// instead of two, for 8 x the multiply-adds -- which the matrix cores do not notice (64 MFMAs per 32 keys against ~3 us of HBM time).
// This is multi-query attention with head_dim 512, K = V = memory, eight query rows per item.
//
// One workgroup (4 waves) per (item, key range); 32-key tiles (32 KiB) arrive by LDS-DMA into a four-slot ring, three tiles in flight,
// 16-byte chunks XOR-swizzled on the SOURCE address (chunk c of key row j sits at slot c ^ (j & 15)).  Wave w owns the feature quarter
// [128 w, 128 w + 128): it multiplies its slice of the keys with its slice of the queries (S^T partial, keys on the MFMA rows), the four
// partials meet in LDS, every wave then runs the same online softmax (base 2, heads on the lanes) and adds P^T times ITS feature quarter
// of the same tile (read back transposed: ds_read_b64_tr_b16) into its 128 x 16 slice of O^T.  The k order of the second product is the
// accumulator order of the first (slot (g, j): key 4 g + j, then 16 + 4 g + j - 4), so P never leaves its lane.
#include "common.h"

namespace mqa {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;

constexpr int D = 512, HEADS = 8, ROWB = D * 2, TK = 32, TILE = TK * ROWB, NSLOT = 4, NTHR = 256;
constexpr int PBUF = 4 * 2 * 1024;  // one partial-score buffer: 4 waves x 2 key blocks x 64 lanes x 16 B
constexpr int MAX_KEYS = 8192;      // key range of one workgroup (validity bytes staged in LDS)
constexpr int LDS_RING = NSLOT * TILE, LDS_BYTES = LDS_RING + 2 * PBUF + MAX_KEYS;
constexpr int PART_STRIDE = 16 + HEADS * D;  // floats per (item, split): m[8], l[8], O[8][512]

struct Args {
  const bf16_t* qp;      // [B, 8, 512] absorbed queries, pre-multiplied by log2(e) / sqrt(head_dim)
  const bf16_t* mem;     // [B, S, 512]
  const uint8_t* valid;  // [B, S] or null
  bf16_t* out;           // [B, ldo]: head h at columns h * 512
  float* part;           // nsplit > 1: [B, nsplit, PART_STRIDE]
  int64_t S, ldo;
  int nsplit, keys_per_split;
};

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma16(i32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
// The memory rows are a pure stream (B x S x 1 KiB per launch, 1 GB at cfg 4, each byte read once per launch): requested NON-TEMPORAL, so that
// they do not push the step's reusable data -- ~64 MB of decoder weights, the self-attention and query-memory caches -- out of the
// 256-MiB Infinity Cache.  Measured on the greedy pass (cfg 4, one box, alternating builds): 2.19 / 2.16 -> 2.09 / 2.08 ms per cached step.
// -DCASE_STREAM_DEFAULT_POLICY builds the default cache policy for A/B runs.
#ifndef CASE_STREAM_DEFAULT_POLICY
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory", "m0");
#else
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory", "m0");
#endif
}
#pragma clang diagnostic pop

#define MQA_FENCE() { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define MQA_BARRIER() { MQA_FENCE() __builtin_amdgcn_s_barrier(); MQA_FENCE() }

__global__ __launch_bounds__(NTHR, 1) void mqa_decode_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, l = tid & 63, g = l >> 4, r = l & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / a.nsplit, sp = blockIdx.x - b * a.nsplit;
  const int64_t k0 = (int64_t)sp * a.keys_per_split;
  const int nkeys = (int)((a.S - k0) < a.keys_per_split ? (a.S - k0) : a.keys_per_split);
  const int nt = (nkeys + TK - 1) / TK;
  char* ring = smem;
  float* pbuf = reinterpret_cast<float*>(smem + LDS_RING);
  uint8_t* vld = reinterpret_cast<uint8_t*>(smem + LDS_RING + 2 * PBUF);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  // validity bytes of the key range (zero behind its end) and a zeroed ring: rows of the last tile that lie behind the key range are
  // never requested as in-range bytes, and 0 x (stale NaN pattern) would poison the second product
  for (int i = tid; i < nt * TK; i += NTHR) vld[i] = i < nkeys ? (a.valid ? a.valid[(int64_t)b * a.S + k0 + i] : (uint8_t)1) : (uint8_t)0;
  {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < LDS_RING / 16; i += NTHR) reinterpret_cast<f32x4*>(ring)[i] = z;
  }
  // this wave's slice of the absorbed queries: head r (rows 8..15 of the MFMA block are padding), features 128 w + 32 ks + 8 g ..
  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    qf[ks] = r < HEADS ? *reinterpret_cast<const bf16x8*>(a.qp + ((int64_t)b * HEADS + r) * D + 128 * w + 32 * ks + 8 * g) : z;
  }
  // the query loads are retired HERE, where the compiler can see it: left to its own bookkeeping it waits vmcnt(0) at their first use --
  // inside the tile loop, behind the asm DMAs it does not count, draining the ring every iteration
  __builtin_amdgcn_s_waitcnt(0x0f70);
  __syncthreads();

  // buffer resource over this workgroup's key range: reads behind it return nothing
  i32x4 rs;
  {
    const char* base = reinterpret_cast<const char*>(a.mem + ((int64_t)b * a.S + k0) * D);
    rs[0] = __builtin_amdgcn_readfirstlane((int)(size_t)base);
    rs[1] = __builtin_amdgcn_readfirstlane((int)(((size_t)base) >> 32) & 0xffff);
    rs[2] = __builtin_amdgcn_readfirstlane(nkeys * ROWB);
    rs[3] = 0x00020000;
  }
  // tile t -> slot t & 3: wave w requests key rows 8 w .. 8 w + 7 (one KiB each); lane c takes chunk c ^ (row & 15)
  // (the tile offset rides in the per-lane offset: the range check of a raw buffer covers the lane offset, not the scalar one)
#define ISSUE(T)                                                                                     \
  {                                                                                                  \
    const unsigned toff = (unsigned)(T) * TILE, dst = lds0 + ((T) & (NSLOT - 1)) * TILE + w * 8 * ROWB; \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                  \
      const unsigned voff = toff + (unsigned)((8 * w + i) * ROWB) + (unsigned)((l ^ ((8 * w + i) & 15)) << 4); \
      dma16(rs, voff, 0u, dst + i * ROWB);                                                           \
    }                                                                                                \
  }
  for (int t = 0; t < 3 && t < nt; ++t) ISSUE(t)

  float m = -INFINITY, lsum = 0.f;
  f32x4 o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int t = 0; t < nt; ++t) {
    // tile t has landed once at most the tiles behind it (8 requests each) are outstanding
    const int rem = nt - 1 - t;
    if (rem >= 2) __builtin_amdgcn_s_waitcnt(0x4f70);       // vmcnt(16)
    else if (rem == 1) __builtin_amdgcn_s_waitcnt(0x0f78);  // vmcnt(8)
    else __builtin_amdgcn_s_waitcnt(0x0f70);                // vmcnt(0)
    MQA_BARRIER()  // A: tile t is visible to every wave; slot (t + 3) & 3 == (t - 1) & 3 has been read by every wave
    if (t + 3 < nt) ISSUE(t + 3)
    const char* slot = ring + (t & (NSLOT - 1)) * TILE;

    // S^T partial over this wave's features: rows = keys (two blocks of 16), columns = heads
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int ch = ((16 * w + 4 * ks + g) ^ r) << 4;
      const bf16x8 k0f = *reinterpret_cast<const bf16x8*>(slot + r * ROWB + ch);
      const bf16x8 k1f = *reinterpret_cast<const bf16x8*>(slot + (16 + r) * ROWB + ch);
      s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0f, qf[ks], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1f, qf[ks], s1, 0, 0, 0);
    }
    float* pb = pbuf + (t & 1) * (PBUF / 4);
    reinterpret_cast<f32x4*>(pb)[(w * 2 + 0) * 64 + l] = s0;
    reinterpret_cast<f32x4*>(pb)[(w * 2 + 1) * 64 + l] = s1;
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the partials are written
    MQA_BARRIER()  // B
    f32x4 x0 = reinterpret_cast<const f32x4*>(pb)[0 * 64 + l], x1 = reinterpret_cast<const f32x4*>(pb)[1 * 64 + l];
#pragma unroll
    for (int ww = 1; ww < 4; ++ww) {  // fixed order: the scores do not depend on which wave runs first
      x0 += reinterpret_cast<const f32x4*>(pb)[(ww * 2 + 0) * 64 + l];
      x1 += reinterpret_cast<const f32x4*>(pb)[(ww * 2 + 1) * 64 + l];
    }
    // keys 32 t + 4 g + e and 32 t + 16 + 4 g + e of head r
    const uint32_t v0 = *reinterpret_cast<const uint32_t*>(vld + t * TK + 4 * g), v1 = *reinterpret_cast<const uint32_t*>(vld + t * TK + 16 + 4 * g);
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      x0[e] = ((v0 >> (8 * e)) & 0xffu) ? x0[e] : -INFINITY;
      x1[e] = ((v1 >> (8 * e)) & 0xffu) ? x1[e] : -INFINITY;
      mx = fmaxf(mx, fmaxf(x0[e], x1[e]));
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m, mx);
    const float m_use = m_new == -INFINITY ? 0.f : m_new;  // no valid key so far: every p below is exp2(-inf) = 0
    const float alpha = __builtin_amdgcn_exp2f(m - m_use);
    float ps = 0.f;
    float p0[4], p1[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      p0[e] = __builtin_amdgcn_exp2f(x0[e] - m_use);
      p1[e] = __builtin_amdgcn_exp2f(x1[e] - m_use);
      ps += p0[e] + p1[e];
    }
    ps += __shfl_xor(ps, 16, 64);
    ps += __shfl_xor(ps, 32, 64);
    lsum = lsum * alpha + ps;
    m = m_new;
    bf16x8 pf;
    {
      const uint32_t w0 = f32x2_to_bf16x2(p0[0], p0[1]), w1 = f32x2_to_bf16x2(p0[2], p0[3]);
      const uint32_t w2 = f32x2_to_bf16x2(p1[0], p1[1]), w3 = f32x2_to_bf16x2(p1[2], p1[3]);
      typedef __attribute__((ext_vector_type(4))) unsigned u4;
      const u4 pk = {w0, w1, w2, w3};
      pf = *reinterpret_cast<const bf16x8*>(&pk);
    }
    // O^T (features on the rows, heads on the columns) += mem^T P^T over the 32 keys; k slot (g, j): key 4 g + j | 16 + 4 g + j - 4
    {
      const int q = r >> 2, pp = r & 3, key1 = 4 * g + q;
      typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
#pragma unroll
      for (int fb = 0; fb < 8; ++fb) {
        const char* p1a = slot + key1 * ROWB + (((16 * w + 2 * fb + (pp >> 1)) ^ key1) << 4) + 8 * (pp & 1);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p1a));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p1a + 16 * ROWB));
        bf16x8 af;
        af[0] = lo[0]; af[1] = lo[1]; af[2] = lo[2]; af[3] = lo[3];
        af[4] = hi[0]; af[5] = hi[1]; af[6] = hi[2]; af[7] = hi[3];
        f32x4 acc = o[fb];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] *= alpha;
        o[fb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, pf, acc, 0, 0, 0);
      }
    }
  }
#undef ISSUE

  // lane (g, r): head r, features 128 w + 16 fb + 4 g + e
  if (a.nsplit == 1) {
    const float inv = lsum > 0.f ? 1.f / lsum : 0.f;
    if (r < HEADS) {
      bf16_t* dst = a.out + (int64_t)b * a.ldo + r * D + 128 * w + 4 * g;
#pragma unroll
      for (int fb = 0; fb < 8; ++fb) {
        uint2 pk;
        pk.x = f32x2_to_bf16x2(o[fb][0] * inv, o[fb][1] * inv);
        pk.y = f32x2_to_bf16x2(o[fb][2] * inv, o[fb][3] * inv);
        *reinterpret_cast<uint2*>(dst + 16 * fb) = pk;
      }
    }
  } else if (r < HEADS) {
    float* dst = a.part + ((int64_t)b * a.nsplit + sp) * PART_STRIDE;
    if (w == 0 && g == 0) {
      dst[r] = m;
      dst[8 + r] = lsum;
    }
#pragma unroll
    for (int fb = 0; fb < 8; ++fb) *reinterpret_cast<f32x4*>(dst + 16 + r * D + 128 * w + 16 * fb + 4 * g) = o[fb];
  }
}

// nsplit > 1: out[b, h, :] = sum_s 2^(m_s - M) O_s / sum_s 2^(m_s - M) l_s, in split order
__global__ __launch_bounds__(256) void mqa_combine_kernel(const float* __restrict__ part, bf16_t* __restrict__ out, int nsplit, int64_t ldo) {
  const int b = blockIdx.x >> 3, h = blockIdx.x & 7;
  const float* p0 = part + (int64_t)b * nsplit * PART_STRIDE;
  float M = -INFINITY;
  for (int s = 0; s < nsplit; ++s) M = fmaxf(M, p0[(int64_t)s * PART_STRIDE + h]);
  const float Mu = M == -INFINITY ? 0.f : M;
  float den = 0.f, acc0 = 0.f, acc1 = 0.f;
  const int f = threadIdx.x * 2;
  for (int s = 0; s < nsplit; ++s) {
    const float* ps = p0 + (int64_t)s * PART_STRIDE;
    const float wgt = __builtin_amdgcn_exp2f(ps[h] - Mu);
    den += wgt * ps[8 + h];
    acc0 += wgt * ps[16 + h * D + f];
    acc1 += wgt * ps[16 + h * D + f + 1];
  }
  const float inv = den > 0.f ? 1.f / den : 0.f;
  *reinterpret_cast<uint32_t*>(out + (int64_t)b * ldo + h * D + f) = f32x2_to_bf16x2(acc0 * inv, acc1 * inv);
}

}  // namespace mqa

extern "C" int64_t case_attention_decode_mqa_workspace(int64_t B, int64_t S, int32_t nsplit) {
  if (B <= 0 || S <= 0 || nsplit <= 1) return 0;
  return B * nsplit * (int64_t)mqa::PART_STRIDE * 4;
}

// how many key ranges per item the launch below should use for (B, S): enough workgroups for the chip, ranges of whole 32-key tiles that
// fit the kernel's validity stage
extern "C" int case_attention_decode_mqa_splits(int64_t B, int64_t S) {
  if (B <= 0 || S <= 0) return 1;
  const int cus = case_device_cus();
  int64_t n = B >= cus ? 1 : (cus + B - 1) / B;
  const int64_t tiles = (S + mqa::TK - 1) / mqa::TK;
  if (n > tiles / 4) n = tiles / 4 > 0 ? tiles / 4 : 1;  // at least four tiles per range
  const int64_t need = (S + mqa::MAX_KEYS - 1) / mqa::MAX_KEYS;
  if (n < need) n = need;
  return (int)n;
}

extern "C" int case_attention_decode_mqa(const void* qp, const void* mem, const uint8_t* key_valid, void* out, int64_t B, int64_t S,
                                         int64_t ldo, int32_t nsplit, void* workspace, int64_t workspace_bytes, case_stream_t stream) {
  CASE_REQUIRE(qp && mem && out && B > 0 && S > 0 && nsplit >= 1 && ldo >= mqa::HEADS * mqa::D && B * (int64_t)nsplit < (1ll << 31),
               "case_attention_decode_mqa: bad argument");
  CASE_REQUIRE(((uintptr_t)qp % 16) == 0 && ((uintptr_t)mem % 16) == 0 && ((uintptr_t)out % 8) == 0 && ldo % 4 == 0,
               "case_attention_decode_mqa: qp / mem must be 16-byte aligned, out 8-byte aligned with ldo %% 4 == 0");
  const int64_t tiles = (S + mqa::TK - 1) / mqa::TK;
  const int64_t per = ((tiles + nsplit - 1) / nsplit) * mqa::TK;  // whole tiles per range
  CASE_REQUIRE(per <= mqa::MAX_KEYS, "case_attention_decode_mqa: %lld keys per range, at most %d (use case_attention_decode_mqa_splits)",
               (long long)per, mqa::MAX_KEYS);
  const int32_t used = (int32_t)((S + per - 1) / per);  // ranges that hold at least one key
  CASE_REQUIRE(used == 1 || (workspace && (uintptr_t)workspace % 16 == 0 && workspace_bytes >= case_attention_decode_mqa_workspace(B, S, used)),
               "case_attention_decode_mqa: workspace of %lld bytes, %lld needed", (long long)workspace_bytes,
               (long long)case_attention_decode_mqa_workspace(B, S, used));
  mqa::Args a;
  a.qp = reinterpret_cast<const bf16_t*>(qp);
  a.mem = reinterpret_cast<const bf16_t*>(mem);
  a.valid = key_valid;
  a.out = reinterpret_cast<bf16_t*>(out);
  a.part = reinterpret_cast<float*>(workspace);
  a.S = S;
  a.ldo = ldo;
  a.nsplit = used;
  a.keys_per_split = (int)per;
  hipStream_t s = (hipStream_t)stream;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&mqa::mqa_decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, mqa::LDS_BYTES) !=
        hipSuccess)
      return case_set_error(CASE_E_LAUNCH, "case_attention_decode_mqa: cannot reserve %d bytes of LDS", mqa::LDS_BYTES);
    attr = true;
  }
  hipLaunchKernelGGL(mqa::mqa_decode_kernel, dim3((unsigned)(B * used)), dim3(mqa::NTHR), mqa::LDS_BYTES, s, a);
  if (used > 1)
    hipLaunchKernelGGL(mqa::mqa_combine_kernel, dim3((unsigned)(B * mqa::HEADS)), dim3(256), 0, s, a.part, a.out, used, ldo);
  return case_check_launch("case_attention_decode_mqa");
}
