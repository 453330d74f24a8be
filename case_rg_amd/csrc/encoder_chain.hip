// K16  the row-local half of the encoder layer as ONE kernel (inference form; reference: common/TransformerEncoder.py:66-75 and
// the next layer's :66-67, nn.MultiheadAttention's out_proj / in_proj around them):
//
//     y  = ctx Wo^T + bo + s            (out-projection of the attention output + the residual of the NORMED input, :68)
//     s2 = LN2(y)                       (:69)
//     a  = gelu(s2 W1^T + b1)           (:72, erf form)
//     o  = a W2^T + b2 + s2             (:72-75)
//     s' = LN1'(o)                      (the NEXT layer's norm1, :66)
//     qkv' = s' Wqkv'^T + bqkv'         (the next layer's packed in-projection, :67)
//
// for a tile of 128 tokens per workgroup: of the 20 activation passes over HBM the unfused layer makes (5.0 GB at 64 x 10 x 384
// tokens) the chain keeps 6 -- ctx and s in, s' and qkv' out -- and every intermediate stays on the CU.  The K = 512 GEMMs of
// this layer are HBM-co-bound when launched one by one: what the chain removes is their traffic, not their FLOPs.
//
// Structure (d_model = dim_feedforward = 512, bf16):
//   * workgroup = FOUR waves, one per SIMD, the whole 512-register file each (__launch_bounds__(256));
//   * the activation tile X [128 tokens][512] lives in LDS (128 KiB, 16-byte chunks XOR-swizzled by the token's low four bits:
//     every fragment read is conflict free) and is rewritten IN PLACE by each stage's epilogue;
//   * the weights do not pass through LDS at all: they are pre-packed (case_encoder_chain_pack, once per parameter update) into
//     MFMA fragment order -- one contiguous KiB per (16 features x 32 k) fragment, laid out in exactly the order a wave consumes
//     them -- and each wave streams its share straight from L2 into registers, three K steps ahead of the MFMAs that use them
//     (every CU reads the same 3 MB per layer: L2-resident);
//   * a wave owns 128 of each GEMM's 512 output features, in two 64-feature passes (128 accumulator registers): per K step of
//     32 it issues 4 weight-fragment loads, 8 token-fragment ds_read_b128 and 32 v_mfma_f32_16x16x32_bf16 with the FEATURES on
//     the MFMA rows, so a lane's accumulators hold four consecutive features of ONE token: bias / residual / GELU / LayerNorm /
//     packing to bf16 all happen in registers, and the packed result is the next GEMM's operand image once written back to X;
//   * LayerNorm statistics: per-lane partial sums, two cross-lane adds, one 4 KiB exchange through LDS;
//   * s2 (needed again as the residual behind FFN2) is parked in a per-workgroup global scratch slab in the lanes' own order
//     (512-byte contiguous stores, read back by the same lanes: it never leaves L2 / the Infinity Cache).
// Variants: HEAD (layer 0: x -> LN1 -> s, qkv), FULL (layer i -> i + 1), TAIL (last layer: o is the encoder output).
#include "common.h"

namespace enc_chain {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) int i32x4;

constexpr int E = 512, TOK = 128, NTHR = 256, NWAVE = 4, KSTEPS = E / 32;
constexpr int XBYTES = TOK * E * 2;              // 128 KiB activation tile
constexpr int STAT_BYTES = TOK * NWAVE * 8;      // [token][wave] (sum, sum of squares)
constexpr int LDS_BYTES = XBYTES + STAT_BYTES;
constexpr int SLOTS = 12;                        // per wave: 2 x Wo, 2 x W1, 2 x W2, 6 x Wqkv sub-chunks of 64 features
constexpr int STEP_BYTES = 4 * 1024;             // one K step of one sub-chunk: 4 fragments of 1 KiB
constexpr int SLOT_BYTES = KSTEPS * STEP_BYTES;  // 64 KiB
constexpr int64_t WAVE_BYTES = (int64_t)SLOTS * SLOT_BYTES, PACKED_BYTES = NWAVE * WAVE_BYTES;  // 3 MiB per layer

enum { VAR_FULL = 0, VAR_TAIL = 1, VAR_HEAD = 2 };

struct Args {
  const bf16_t* ctx;     // FULL / TAIL: attention output [M, 512]; HEAD: the raw input x [M, 512]
  const bf16_t* resid;   // FULL / TAIL: s [M, 512] (this layer's normed input)
  const bf16_t* wpk;     // packed weights (PACKED_BYTES)
  const float *bo, *b1, *b2, *bqkv, *g2, *be2, *g1n, *be1n;
  bf16_t* s_out;         // FULL / HEAD: s' [M, 512]; TAIL: the layer output o [M, 512]
  bf16_t* qkv_out;       // FULL / HEAD: [M, 1536]
  bf16_t* scratch;       // FULL / TAIL: gridDim.x slabs of 128 x 512 bf16
  int64_t M;
  float eps2, eps1n;
};

// ---- weight packing -------------------------------------------------------------------------------------------------------------------------
// packed[wave][slot][ks][nb][lane][8]: lane l of fragment (nb, ks) holds W[n0 + 16 nb + (l & 15)][32 ks + 8 (l >> 4) + 0..7], the A operand
// of v_mfma_f32_16x16x32_bf16 with the features on the rows.  slot -> (matrix, first feature n0): 0-1 Wo, 2-3 W1, 4-5 W2 at
// n0 = 64 (2 wave + (slot & 1)); 6 + 2 c + i: Wqkv at 512 c + 64 (2 wave + i).
__global__ __launch_bounds__(256) void pack_kernel(const bf16_t* __restrict__ wo, const bf16_t* __restrict__ w1, const bf16_t* __restrict__ w2,
                                                   const bf16_t* __restrict__ wqkv, u32x4* __restrict__ out) {
  const int64_t frag = blockIdx.x * 4 + (threadIdx.x >> 6);  // fragment index: ((wave * SLOTS + slot) * KSTEPS + ks) * 4 + nb
  const int l = threadIdx.x & 63;
  const int nb = (int)(frag & 3), ks = (int)((frag >> 2) % KSTEPS), slot = (int)((frag / (4 * KSTEPS)) % SLOTS),
            wave = (int)(frag / (4 * KSTEPS * SLOTS));
  const bf16_t* src;
  int n0;
  if (slot < 6) {
    src = slot < 2 ? wo : (slot < 4 ? w1 : w2);
    n0 = 64 * (2 * wave + (slot & 1));
  } else {
    src = wqkv;
    n0 = 512 * ((slot - 6) >> 1) + 64 * (2 * wave + (slot & 1));
  }
  u32x4 v = {0u, 0u, 0u, 0u};
  if (src) v = *reinterpret_cast<const u32x4*>(src + (int64_t)(n0 + 16 * nb + (l & 15)) * E + 32 * ks + 8 * (l >> 4));
  out[frag * 64 + l] = v;
}

// ---- device helpers -------------------------------------------------------------------------------------------------------------------------
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
__device__ __forceinline__ void dma16(i32x4 rsrc, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc) : "memory", "m0");
}
#pragma clang diagnostic pop

__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// physical byte offset of 16-byte chunk c (0..63) of token row r in the X image
__device__ __forceinline__ int x_off(int r, int c) { return r * 1024 + ((c ^ (r & 15)) << 4); }

template <int VARIANT>
__global__ __launch_bounds__(NTHR) void chain_kernel(const Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, l = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lt = l & 15, lg = l >> 4;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  float* stats = reinterpret_cast<float*>(smem + XBYTES);

  constexpr int FIRST_SLOT = VARIANT == VAR_HEAD ? 6 : 0, LAST_SLOT = VARIANT == VAR_TAIL ? 6 : 12;
  constexpr int NSTEP = (LAST_SLOT - FIRST_SLOT) * KSTEPS;  // K steps per tile of this wave's weight stream
  const int ntiles = (int)((g.M + TOK - 1) / TOK);

  // the wave's weight stream: step s (0 .. NSTEP-1, then it wraps for the next tile) = 4 fragments at wbase + s * STEP_BYTES
  const __amdgpu_buffer_rsrc_t wr = as_rsrc(reinterpret_cast<const char*>(g.wpk) + wave * WAVE_BYTES + (int64_t)FIRST_SLOT * SLOT_BYTES,
                                            (uint32_t)(NSTEP * STEP_BYTES));
  const int wv = l * 16;
  u32x4 wring[4][4];  // K steps in flight: ring of 4, prefetch distance 3
  int wstep = 0;      // stream position of the NEXT step to request
  auto request = [&](u32x4 (&dst)[4]) {
    const int so = wstep * STEP_BYTES;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) dst[nb] = __builtin_amdgcn_raw_buffer_load_b128(wr, wv + nb * 1024, so, 0);
    wstep = wstep + 1 == NSTEP ? 0 : wstep + 1;
  };
  request(wring[0]);
  request(wring[1]);
  request(wring[2]);

  // per-lane parts of the global addresses (the uniform parts ride in the scalar offset) and of the X fragment addresses
  const int v_row = (lt * E + 4 * lg) * 2, v_row3 = (lt * 3 * E + 4 * lg) * 2;
  const int xlane = lt * 1024 + ((lg ^ (lt & 3)) << 4);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = (int64_t)tile * TOK;
    const int rows = (int)(g.M - row0 < TOK ? g.M - row0 : TOK);
    const uint32_t tile_bytes_e = (uint32_t)rows * E * 2;

    // ---- stage 0: the input tile -> X (LDS-DMA, one 1 KiB row per wave-instruction, chunks permuted on the SOURCE side) -------------
    {
      const i32x4 rs = make_rsrc(g.ctx + row0 * E, tile_bytes_e);
#pragma unroll 4
      for (int i = 0; i < TOK / NWAVE; ++i) {
        const int r = wave * (TOK / NWAVE) + i;
        dma16(rs, (unsigned)(r * 1024 + ((l ^ (r & 15)) << 4)), lds0 + r * 1024);
      }
      __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's rows have landed
      __syncthreads();
    }

    // LayerNorm of the tile's rows IN LDS (HEAD: x -> s): a wave normalises 32 rows, 8 elements per lane per row
    if constexpr (VARIANT == VAR_HEAD) {
      const __amdgpu_buffer_rsrc_t so = as_rsrc(g.s_out + row0 * E, tile_bytes_e);
      float gam[8], bet[8];
#pragma unroll
      for (int e = 0; e < 8; e += 4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(g.g1n + l * 8 + e), b = *reinterpret_cast<const f32x4*>(g.be1n + l * 8 + e);
#pragma unroll
        for (int i = 0; i < 4; ++i) { gam[e + i] = a[i]; bet[e + i] = b[i]; }
      }
      for (int i = 0; i < TOK / NWAVE; ++i) {
        const int r = wave * (TOK / NWAVE) + i;
        u32x4* p = reinterpret_cast<u32x4*>(smem + x_off(r, l));
        const u32x4 w = *p;
        float x[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) { x[2 * k] = bf_lo(w[k]); x[2 * k + 1] = bf_hi(w[k]); }
        float s1 = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s1 += x[k];
        const float mean = wave_sum(s1) * (1.f / E);
        float s2 = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { x[k] -= mean; s2 += x[k] * x[k]; }
        const float rstd = rsqrtf(wave_sum(s2) * (1.f / E) + g.eps1n);
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          o[k] = f32x2_to_bf16x2(x[2 * k] * rstd * gam[2 * k] + bet[2 * k], x[2 * k + 1] * rstd * gam[2 * k + 1] + bet[2 * k + 1]);
        *p = o;
        __builtin_amdgcn_raw_buffer_store_b128(o, so, r * 1024 + l * 16, 0, 0);  // rows beyond M fall outside the descriptor
      }
      __syncthreads();
    }

    uint32_t hold[2][8][4][2];  // the wave's packed outputs of a stage (2 sub-chunks x 8 token blocks x 4 feature blocks x 4 bf16)
    float ps1[8], ps2[8];       // LayerNorm partial sums per token block (this lane's features)

    const __amdgpu_buffer_rsrc_t rres = as_rsrc(g.resid ? g.resid + row0 * E : nullptr, g.resid ? tile_bytes_e : 0);
    const __amdgpu_buffer_rsrc_t rscr = as_rsrc(g.scratch ? g.scratch + (int64_t)blockIdx.x * TOK * E : nullptr, g.scratch ? XBYTES : 0);
    const __amdgpu_buffer_rsrc_t rso = as_rsrc(g.s_out + row0 * E, tile_bytes_e);
    const __amdgpu_buffer_rsrc_t rqkv = as_rsrc(g.qkv_out ? g.qkv_out + row0 * (3 * E) : nullptr, g.qkv_out ? (uint32_t)rows * 3 * E * 2 : 0);

    // stages: 0 out-proj, 1 FFN1, 2 FFN2, 3..5 QKV chunks
    constexpr int ST0 = VARIANT == VAR_HEAD ? 3 : 0, ST1 = VARIANT == VAR_TAIL ? 3 : 6;
    for (int st = ST0; st < ST1; ++st) {
      if (st == 0 || st == 2) {
#pragma unroll
        for (int tb = 0; tb < 8; ++tb) ps1[tb] = ps2[tb] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // feature origin of this sub-chunk inside its stage's 512-wide output, and the per-lane feature offset
        const int n0 = 64 * (2 * wave + j);
        const float* bias = st == 0 ? g.bo : (st == 1 ? g.b1 : (st == 2 ? g.b2 : g.bqkv + 512 * (st - 3)));
        f32x4 acc[8][4];
        // ---- accumulator init: bias (+ residual) ---------------------------------------------------------------------------
        {
          f32x4 b4[4];
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) b4[nb] = *reinterpret_cast<const f32x4*>(bias + n0 + 16 * nb + 4 * lg);
          if (st == 0 || st == 2) {
            u32x2 rr[8][4];
            if (st == 0) {  // s, row-major [M, 512]: 4 consecutive features of token 16 tb + lt
#pragma unroll
              for (int tb = 0; tb < 8; ++tb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                  rr[tb][nb] = __builtin_amdgcn_raw_buffer_load_b64(rres, v_row, tb * 16 * E * 2 + (n0 + 16 * nb) * 2, 0);
            } else {  // s2 from the scratch slab, lane order
#pragma unroll
              for (int tb = 0; tb < 8; ++tb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                  rr[tb][nb] = __builtin_amdgcn_raw_buffer_load_b64(rscr, l * 8, (((wave * 2 + j) * 8 + tb) * 4 + nb) * 512, 0);
            }
#pragma unroll
            for (int tb = 0; tb < 8; ++tb)
#pragma unroll
              for (int nb = 0; nb < 4; ++nb) {
                acc[tb][nb][0] = b4[nb][0] + bf_lo(rr[tb][nb][0]);
                acc[tb][nb][1] = b4[nb][1] + bf_hi(rr[tb][nb][0]);
                acc[tb][nb][2] = b4[nb][2] + bf_lo(rr[tb][nb][1]);
                acc[tb][nb][3] = b4[nb][3] + bf_hi(rr[tb][nb][1]);
              }
          } else {
#pragma unroll
            for (int tb = 0; tb < 8; ++tb)
#pragma unroll
              for (int nb = 0; nb < 4; ++nb) acc[tb][nb] = b4[nb];
          }
        }
        // ---- K loop: 16 steps of 32 -----------------------------------------------------------------------------------------
        // X fragment of token block tb, K step ks: chunk 4 ks + lg of row 16 tb + lt, i.e. byte
        //   lt * 1024 + tb * 16384 + ((((4 ks + lg) ^ lt)) << 4) = xlane + ((ks ^ (lt >> 2)) << 6) + tb * 16384,   xlane = lt * 1024 + ((lg ^ (lt & 3)) << 4)
        // One fragment buffer: fragment tb of the NEXT step is requested right behind the four MFMAs that consumed fragment tb of
        // this one (28 MFMAs = 450 cycles cover the LDS latency).
        {
          int xl = xlane, xh = lt >> 2;
          asm volatile("" : "+v"(xl), "+v"(xh));  // opaque: keeps the 16 per-step addresses from being hoisted out of the tile loop (and spilled)
          bf16x8 xf[8];
          {
            const char* p0 = smem + xl + ((0 ^ xh) << 6);
#pragma unroll
            for (int tb = 0; tb < 8; ++tb) xf[tb] = *reinterpret_cast<const bf16x8*>(p0 + tb * 16384);
          }
#pragma unroll
          for (int ks = 0; ks < KSTEPS; ++ks) {
#ifndef CHAIN_DBG_NO_WLOAD
            request(wring[(ks + 3) & 3]);
#endif
            const char* pn = smem + xl + ((((ks + 1) & (KSTEPS - 1)) ^ xh) << 6);
#pragma unroll
            for (int tb = 0; tb < 8; ++tb) {
#pragma unroll
              for (int nb = 0; nb < 4; ++nb)
                acc[tb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&wring[ks & 3][nb]), xf[tb], acc[tb][nb], 0, 0, 0);
#ifndef CHAIN_DBG_NO_XREAD
              if (ks + 1 < KSTEPS) xf[tb] = *reinterpret_cast<const bf16x8*>(pn + tb * 16384);
#endif
              __builtin_amdgcn_sched_barrier(0);  // keep the refill behind its MFMAs and each step's requests inside the step
            }
          }
        }
        // ---- per-sub-chunk epilogue ------------------------------------------------------------------------------------------
        if (st >= 3) {  // QKV: bias is in, pack and store [M, 1536]
          const int col = 512 * (st - 3) + n0;
#pragma unroll
          for (int tb = 0; tb < 8; ++tb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
              u32x2 w;
              w[0] = f32x2_to_bf16x2(acc[tb][nb][0], acc[tb][nb][1]);
              w[1] = f32x2_to_bf16x2(acc[tb][nb][2], acc[tb][nb][3]);
              __builtin_amdgcn_raw_buffer_store_b64(w, rqkv, v_row3, tb * 16 * 3 * E * 2 + (col + 16 * nb) * 2, 0);
            }
        } else {
#pragma unroll
          for (int tb = 0; tb < 8; ++tb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
              float v0 = acc[tb][nb][0], v1 = acc[tb][nb][1], v2 = acc[tb][nb][2], v3 = acc[tb][nb][3];
              if (st == 1) { v0 = gelu_f(v0); v1 = gelu_f(v1); v2 = gelu_f(v2); v3 = gelu_f(v3); }
              const uint32_t w0 = f32x2_to_bf16x2(v0, v1), w1 = f32x2_to_bf16x2(v2, v3);
              hold[j][tb][nb][0] = w0;
              hold[j][tb][nb][1] = w1;
              if (st != 1 && !(VARIANT == VAR_TAIL && st == 2)) {  // statistics of the ROUNDED values, as a LayerNorm pass over bf16 y sees them
                const float r0 = bf_lo(w0), r1 = bf_hi(w0), r2 = bf_lo(w1), r3 = bf_hi(w1);
                ps1[tb] += (r0 + r1) + (r2 + r3);
                ps2[tb] += (r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3);
              }
            }
        }
      }
      if (st >= 3) continue;

      // ---- stage epilogue: (LayerNorm), write the 128 x 128 block of this wave back into X ------------------------------------------
      const bool do_ln = (st == 0) || (st == 2 && VARIANT != VAR_TAIL);
      float mean[8], rstd[8];
      if (do_ln) {
#pragma unroll
        for (int tb = 0; tb < 8; ++tb) {
          float a = ps1[tb], b = ps2[tb];
          a += __shfl_xor(a, 16);
          b += __shfl_xor(b, 16);
          a += __shfl_xor(a, 32);
          b += __shfl_xor(b, 32);
          if (lg == 0) *reinterpret_cast<float2*>(stats + ((tb * 16 + lt) * NWAVE + wave) * 2) = make_float2(a, b);
        }
      }
      __syncthreads();  // every wave is behind its K loops: X is free; the partial sums are visible
      if (do_ln) {
        const float eps = st == 0 ? g.eps2 : g.eps1n;
#pragma unroll
        for (int tb = 0; tb < 8; ++tb) {
          const f32x4 p0 = *reinterpret_cast<const f32x4*>(stats + (tb * 16 + lt) * NWAVE * 2),
                      p1 = *reinterpret_cast<const f32x4*>(stats + (tb * 16 + lt) * NWAVE * 2 + 4);
          const float s1 = (p0[0] + p0[2]) + (p1[0] + p1[2]), s2 = (p0[1] + p0[3]) + (p1[1] + p1[3]);
          mean[tb] = s1 * (1.f / E);
          rstd[tb] = rsqrtf(fmaxf(s2 * (1.f / E) - mean[tb] * mean[tb], 0.f) + eps);
        }
      }
      const float* gam = st == 0 ? g.g2 : g.g1n;
      const float* bet = st == 0 ? g.be2 : g.be1n;
      int xq = lt ^ (lg >> 1), xw = lt * 1024 + (lg & 1) * 8;
      asm volatile("" : "+v"(xq), "+v"(xw));
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n0 = 64 * (2 * wave + j);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          f32x4 gm = {1.f, 1.f, 1.f, 1.f}, bt = {0.f, 0.f, 0.f, 0.f};
          if (do_ln) {
            gm = *reinterpret_cast<const f32x4*>(gam + n0 + 16 * nb + 4 * lg);
            bt = *reinterpret_cast<const f32x4*>(bet + n0 + 16 * nb + 4 * lg);
          }
#pragma unroll
          for (int tb = 0; tb < 8; ++tb) {
            uint32_t w0 = hold[j][tb][nb][0], w1 = hold[j][tb][nb][1];
            if (do_ln) {
              const float m = mean[tb], r = rstd[tb];
              w0 = f32x2_to_bf16x2((bf_lo(w0) - m) * r * gm[0] + bt[0], (bf_hi(w0) - m) * r * gm[1] + bt[1]);
              w1 = f32x2_to_bf16x2((bf_lo(w1) - m) * r * gm[2] + bt[2], (bf_hi(w1) - m) * r * gm[3] + bt[3]);
            }
            u32x2 w;
            w[0] = w0;
            w[1] = w1;
            // X image: features n0 + 16 nb + 4 lg .. + 3 of token 16 tb + lt = half (lg & 1) of chunk (n0 + 16 nb) / 8 + (lg >> 1)
            //   = xw + ((8 j + 2 nb) ^ xq) * 16 + tb * 16384 + wave * 256,   xq = lt ^ (lg >> 1),   xw = lt * 1024 + (lg & 1) * 8
            *reinterpret_cast<u32x2*>(smem + xw + (((8 * j + 2 * nb) ^ xq) << 4) + tb * 16384 + wave * 256) = w;
            if (st == 0)  // s2: parked for the FFN2 residual, lane order
              __builtin_amdgcn_raw_buffer_store_b64(w, rscr, l * 8, (((wave * 2 + j) * 8 + tb) * 4 + nb) * 512, 0);
            if (st == 2)  // s' (FULL) or the layer output (TAIL), row-major
              __builtin_amdgcn_raw_buffer_store_b64(w, rso, v_row, tb * 16 * E * 2 + (n0 + 16 * nb) * 2, 0);
          }
        }
      }
      __syncthreads();  // X holds the next stage's operand
    }
    __syncthreads();  // the last K loops are done before the next tile's DMA overwrites X
  }
}

template <int VARIANT>
int launch(const Args& a, int cus, hipStream_t s) {
  const int ntiles = (int)((a.M + TOK - 1) / TOK);
  const dim3 grid(ntiles < cus ? ntiles : cus), block(NTHR);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_kernel<VARIANT>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) !=
      hipSuccess)
    return case_set_error(CASE_E_LAUNCH, "case_encoder_chain: cannot reserve %d bytes of LDS", LDS_BYTES);
  hipLaunchKernelGGL((chain_kernel<VARIANT>), grid, block, LDS_BYTES, s, a);
  return case_check_launch("case_encoder_chain");
}

}  // namespace enc_chain

extern "C" int64_t case_encoder_chain_packed_bytes(void) { return enc_chain::PACKED_BYTES; }
extern "C" int64_t case_encoder_chain_scratch_bytes(void) { return (int64_t)256 * enc_chain::XBYTES; }

extern "C" int case_encoder_chain_pack(const void* wo, const void* w1, const void* w2, const void* wqkv, void* packed, case_stream_t stream) {
  CASE_REQUIRE(packed, "case_encoder_chain_pack: null output");
  CASE_REQUIRE(((reinterpret_cast<uintptr_t>(wo) | reinterpret_cast<uintptr_t>(w1) | reinterpret_cast<uintptr_t>(w2) |
                 reinterpret_cast<uintptr_t>(wqkv) | reinterpret_cast<uintptr_t>(packed)) & 15) == 0,
               "case_encoder_chain_pack: operands must be 16-byte aligned");
  const int64_t frags = (int64_t)enc_chain::NWAVE * enc_chain::SLOTS * enc_chain::KSTEPS * 4;
  hipLaunchKernelGGL(enc_chain::pack_kernel, dim3((unsigned)(frags / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)wo,
                     (const bf16_t*)w1, (const bf16_t*)w2, (const bf16_t*)wqkv, (enc_chain::u32x4*)packed);
  return case_check_launch("case_encoder_chain_pack");
}

extern "C" int case_encoder_chain(const CaseEncoderChainDesc* d, const void* x_in, const void* resid, const void* packed, const float* bo,
                                  const float* b1, const float* b2, const float* bqkv, const float* ln2_g, const float* ln2_b,
                                  const float* ln1n_g, const float* ln1n_b, void* s_out, void* qkv_out, void* scratch, case_stream_t stream) {
  CASE_REQUIRE(d && x_in && packed && s_out, "case_encoder_chain: null argument");
  CASE_REQUIRE(d->width == enc_chain::E, "case_encoder_chain: built for d_model = dim_feedforward = 512 (got %d)", (int)d->width);
  CASE_REQUIRE(d->rows > 0 && d->rows < (1ll << 31) - 128, "case_encoder_chain: bad row count");
  CASE_REQUIRE(d->variant >= 0 && d->variant <= 2, "case_encoder_chain: variant must be 0 (full), 1 (tail) or 2 (head)");
  const bool head = d->variant == enc_chain::VAR_HEAD, tail = d->variant == enc_chain::VAR_TAIL;
  if (!head) CASE_REQUIRE(resid && bo && b1 && b2 && ln2_g && ln2_b && scratch, "case_encoder_chain: the layer stages need resid, biases, LN2 and the scratch slab");
  if (!tail) CASE_REQUIRE(bqkv && ln1n_g && ln1n_b && qkv_out, "case_encoder_chain: the LN + QKV stage needs its parameters and qkv_out");
  for (const void* p : {x_in, resid, packed, (const void*)s_out, (const void*)qkv_out, (const void*)scratch, (const void*)bo, (const void*)b1,
                        (const void*)b2, (const void*)bqkv, (const void*)ln2_g, (const void*)ln2_b, (const void*)ln1n_g, (const void*)ln1n_b})
    CASE_REQUIRE((reinterpret_cast<uintptr_t>(p) & 15) == 0, "case_encoder_chain: operands must be 16-byte aligned");
  enc_chain::Args a;
  a.ctx = (const bf16_t*)x_in;
  a.resid = (const bf16_t*)resid;
  a.wpk = (const bf16_t*)packed;
  a.bo = bo; a.b1 = b1; a.b2 = b2; a.bqkv = bqkv;
  a.g2 = ln2_g; a.be2 = ln2_b; a.g1n = ln1n_g; a.be1n = ln1n_b;
  a.s_out = (bf16_t*)s_out;
  a.qkv_out = (bf16_t*)qkv_out;
  a.scratch = (bf16_t*)scratch;
  a.M = d->rows;
  a.eps2 = d->eps_ln2;
  a.eps1n = d->eps_ln1_next;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  if (cus > 256) cus = 256;  // the scratch slab is sized for 256 workgroups
  switch (d->variant) {
    case enc_chain::VAR_FULL: return enc_chain::launch<enc_chain::VAR_FULL>(a, cus, (hipStream_t)stream);
    case enc_chain::VAR_TAIL: return enc_chain::launch<enc_chain::VAR_TAIL>(a, cus, (hipStream_t)stream);
    default: return enc_chain::launch<enc_chain::VAR_HEAD>(a, cus, (hipStream_t)stream);
  }
}
