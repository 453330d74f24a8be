// K8  the query / passage dual co-attention as TWO kernels (round 6; reference: common/Interaction.py:32-74):
//
//     U[i, j] = w1 . Eq[j] + w2 . Ep[i] + (w3 o Ep[i]) . Eq[j]                 (:32-36, without the [n, Lp, Lq, 3H] tensor)
//     A  = softmax_j U (0 at masked positions, all-masked rows 0)               (:38-46)
//     Bm = softmax_i U
//     A1 = A Eq,  B1 = Bm^T Ep,  A2 = A B1,  B2 = Bm^T A1                       (:48-52)
//     G_q_p = [Ep, A1, A2, Ep o A1, Ep o A2] (0 at padded passage rows),  G_p_q = [Eq, B1, B2, Eq o B1, Eq o B2] (0 at padded query rows)   (:65-72)
//
// per (item, passage) pair: Lq = 64 query rows, Lp <= 512 passage rows (a multiple of 32), H = 512, bf16.  The single-launch path makes 16
// launches of this and moves A1 / B1 / A2 / B2 through HBM twice (product out, concatenation in); here
//   scores_kernel    one workgroup per pair: U on the MFMA pipes with the passage rows straight from global memory in fragment order and
//                    the query side (w3 o Eq[j] + w2, so that the row term rides in the product) resident in LDS, the whole [Lp, 64] score tile
//                    in the accumulators, both softmaxes in the epilogue (rows: cross-lane; columns: lane-local + one LDS exchange between the
//                    four waves) -> A [n, Lp, 64] and Bm^T [n, 64, Lp] in bf16 (what the backward pass reads as saved probabilities);
//   products_kernel  one workgroup per pair: Eq resident in LDS, Ep streamed in 32-row chunks; per chunk A1 = A Eq, B1 += Bm^T Ep,
//                    B2 += Bm^T A1 and the [Ep, A1, Ep o A1] columns of G_q_p written from LDS as whole 16-byte vectors; then B1 / B2 -> G_p_q,
//                    and a second sweep over the chunks for A2 = A B1 and the [A2, Ep o A2] columns.  Every operand of the "k-major" kind
//                    (Ep, Eq, A1, B1 with the contraction index on their ROWS) sits in LDS as it is in memory and is read with the
//                    transposing ds_read_tr16_b64, as in gemm_impl.inc.
// HBM traffic per pair: Ep twice + Eq + the probabilities in, G_q_p and G_p_q out once -- the 5H-wide outputs are the floor.
#include "common.h"

namespace k8 {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

constexpr int H = 512, LQ = 64;
constexpr int KM = H * 2 + 64;   // row stride of a [k][512] image (bytes): rows 16 banks apart for the transposing reads
constexpr int RSQ = H * 2 + 16;  // row stride of the k-contiguous query image of the score kernel

struct Args {
  const bf16_t* eq; const bf16_t* ep; const uint8_t* qv; const uint8_t* pv; const float* w;  // w = [w1 | w2 | w3], 3H floats
  bf16_t* a; bf16_t* bt;      // [n][Lp][64], [n][64][Lp]
  bf16_t* gqp; bf16_t* gpq;   // [n][Lp][5H], [n][64][5H]
  int n, Lp, eq_div;          // pair p reads Eq / qv of query p / eq_div (one query against P passages: eq_div = P)
};

__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// MFMA 32x32x16 fragments.  A-side (rows on the M axis) from a k-contiguous image: lane (r = l & 31, h = l >> 5) holds row m0 + r, k0 + 8 h .. + 7.
__device__ __forceinline__ bf16x8 frag_rm(const char* img, int rs, int m0, int k0, int l) {
  return *reinterpret_cast<const bf16x8*>(img + (m0 + (l & 31)) * rs + (k0 + 8 * (l >> 5)) * 2);
}
// B-side (columns on the N axis) from a [k][n] image: lane (c = l & 31, h = l >> 5) needs column n0 + c, rows k0 + 8 h .. + 7 -- two transposing
// reads: per 16-lane group a 4 (k) x 16 (n) block; lane 4 q + p passes the address of k-row q, columns 4 p .. 4 p + 3, and receives column (l & 15).
__device__ __forceinline__ bf16x8 frag_km(const char* img, int n0, int k0, int l) {
  const int kk = k0 + 8 * (l >> 5), q = (l & 15) >> 2, p = l & 3;
  const int off = (kk + q) * KM + (n0 + 16 * ((l >> 4) & 1) + 4 * p) * 2;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + off));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + off + 4 * KM));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}
// accumulator layout of a 32 x 32 tile: register e of lane l is C[(e & 3) + 8 (e >> 2) + 4 (l >> 5)][l & 31]
__device__ __forceinline__ int acc_row(int e, int l) { return (e & 3) + 8 * (e >> 2) + 4 * (l >> 5); }

// =============================================================================================================================================
// scores_kernel: 256 threads, TW = row tiles (of 32 passage rows) per wave: 3 for Lp <= 384, 4 for Lp <= 512
// =============================================================================================================================================
constexpr int SC_LDS = LQ * RSQ + LQ * 4 + 2 * 4 * LQ * 4;

template <int TW>
__global__ __launch_bounds__(256, 2) void scores_kernel(const Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* bq = smem;                                          // [64][H] bf16: w3 o Eq[j] + w2
  float* cq = reinterpret_cast<float*>(smem + LQ * RSQ);    // [64]: w1 . Eq[j]
  float* red = cq + LQ;                                     // [2][4 waves][64]
  const int t = threadIdx.x, l = t & 63, wave = t >> 6;
  const int pair = blockIdx.x, qi = pair / g.eq_div, Lp = g.Lp, RT = Lp >> 5;
  const bf16_t* eq = g.eq + (int64_t)qi * LQ * H;
  const bf16_t* ep = g.ep + (int64_t)pair * Lp * H;
  const uint8_t* qv = g.qv + (int64_t)qi * LQ;
  const uint8_t* pv = g.pv + (int64_t)pair * Lp;
  {  // stage the query side: thread (j = t >> 2, quarter = t & 3) handles 128 features of row j
    const int j = t >> 2, c0 = (t & 3) * 128;
    float dot = 0.f;
#pragma unroll 4
    for (int c = 0; c < 128; c += 8) {
      const u32x4 x = *reinterpret_cast<const u32x4*>(eq + j * H + c0 + c);
      float v[8] = {bf_lo(x[0]), bf_hi(x[0]), bf_lo(x[1]), bf_hi(x[1]), bf_lo(x[2]), bf_hi(x[2]), bf_lo(x[3]), bf_hi(x[3])};
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        dot = fmaf(v[e], g.w[c0 + c + e], dot);
        o[e] = fmaf(v[e], g.w[2 * H + c0 + c + e], g.w[H + c0 + c + e]);
      }
      u32x4 pk = {f32x2_to_bf16x2(o[0], o[1]), f32x2_to_bf16x2(o[2], o[3]), f32x2_to_bf16x2(o[4], o[5]), f32x2_to_bf16x2(o[6], o[7])};
      *reinterpret_cast<u32x4*>(bq + j * RSQ + (c0 + c) * 2) = pk;
    }
    dot += __shfl_xor(dot, 1);
    dot += __shfl_xor(dot, 2);
    if ((t & 3) == 0) cq[j] = dot;
  }
  __syncthreads();

  f32x16 acc[TW][2];
#pragma unroll
  for (int tt = 0; tt < TW; ++tt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[tt][ct][e] = 0.f;
  // K in groups of 64: lane half h owns k = 64 g + 32 h .. + 31 of its row (64 contiguous bytes: two lanes cover a 128-byte line); the
  // four K steps of a group take 8 of them each -- the same permutation of k on both operands
  for (int gk = 0; gk < H / 64; ++gk) {
    u32x4 a[TW][4];
#pragma unroll
    for (int tt = 0; tt < TW; ++tt) {
      const int tile = wave + 4 * tt;
      const int row = (tile < RT ? tile : 0) * 32 + (l & 31);
      const bf16_t* p = ep + (int64_t)row * H + 64 * gk + 32 * (l >> 5);
#pragma unroll
      for (int s = 0; s < 4; ++s) a[tt][s] = *reinterpret_cast<const u32x4*>(p + 8 * s);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf16x8 b[2];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
        b[ct] = *reinterpret_cast<const bf16x8*>(bq + (32 * ct + (l & 31)) * RSQ + (64 * gk + 32 * (l >> 5) + 8 * s) * 2);
#pragma unroll
      for (int tt = 0; tt < TW; ++tt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
          acc[tt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&a[tt][s]), b[ct], acc[tt][ct], 0, 0, 0);
    }
  }
  // ---- epilogue: + w1 . Eq[j], masks, both softmaxes -------------------------------------------------------------------------------------
  const int c_lane = l & 31;
  float cqv[2];
  bool qok[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    cqv[ct] = cq[32 * ct + c_lane];
    qok[ct] = qv[32 * ct + c_lane] != 0;
  }
  float cmax[2] = {-INFINITY, -INFINITY};
#pragma unroll
  for (int tt = 0; tt < TW; ++tt) {
    const int tile = wave + 4 * tt;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = tile * 32 + acc_row(e, l);
      const bool pok = tile < RT && pv[tile < RT ? row : 0] != 0;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const float u = (pok && qok[ct]) ? acc[tt][ct][e] + cqv[ct] : -INFINITY;
        acc[tt][ct][e] = u;
        cmax[ct] = fmaxf(cmax[ct], u);
      }
    }
  }
  // column statistics: lane-local over the wave's rows, the two lane halves, then the four waves through LDS
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    cmax[ct] = fmaxf(cmax[ct], __shfl_xor(cmax[ct], 32));
    if (l < 32) red[wave * LQ + 32 * ct + c_lane] = cmax[ct];
  }
  __syncthreads();
  float csum[2] = {0.f, 0.f};
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int c = 32 * ct + c_lane;
    cmax[ct] = fmaxf(fmaxf(red[c], red[LQ + c]), fmaxf(red[2 * LQ + c], red[3 * LQ + c]));
  }
  // rows first (A), keeping u; then the column exponentials
  bf16_t* ao = g.a + (int64_t)pair * Lp * LQ;
  bf16_t* bo = g.bt + (int64_t)pair * LQ * Lp;
#pragma unroll
  for (int tt = 0; tt < TW; ++tt) {
    const int tile = wave + 4 * tt;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float m = fmaxf(acc[tt][0][e], acc[tt][1][e]);
#pragma unroll
      for (int o = 1; o < 32; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
      const float e0 = m > -INFINITY ? __expf(acc[tt][0][e] - m) : 0.f, e1 = m > -INFINITY ? __expf(acc[tt][1][e] - m) : 0.f;
      float s = e0 + e1;
#pragma unroll
      for (int o = 1; o < 32; o <<= 1) s += __shfl_xor(s, o);
      const float inv = s > 0.f ? 1.f / s : 0.f;
      if (tile < RT) {
        const int row = tile * 32 + acc_row(e, l);
        ao[row * LQ + c_lane] = f32_to_bf16(e0 * inv);
        ao[row * LQ + 32 + c_lane] = f32_to_bf16(e1 * inv);
      }
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const float x = cmax[ct] > -INFINITY ? __expf(acc[tt][ct][e] - cmax[ct]) : 0.f;  // exp(-inf) = 0 for masked positions
        acc[tt][ct][e] = x;
        csum[ct] += x;
      }
    }
  }
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    csum[ct] += __shfl_xor(csum[ct], 32);
    if (l < 32) red[4 * LQ + wave * LQ + 32 * ct + c_lane] = csum[ct];
  }
  __syncthreads();
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int c = 32 * ct + c_lane;
    const float s = red[4 * LQ + c] + red[5 * LQ + c] + red[6 * LQ + c] + red[7 * LQ + c];
    const float inv = s > 0.f ? 1.f / s : 0.f;
#pragma unroll
    for (int tt = 0; tt < TW; ++tt) {
      const int tile = wave + 4 * tt;
      if (tile < RT) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // four consecutive passage rows of one query column: 8 contiguous bytes of Bm^T
          u32x2 pk = {f32x2_to_bf16x2(acc[tt][ct][4 * q] * inv, acc[tt][ct][4 * q + 1] * inv),
                      f32x2_to_bf16x2(acc[tt][ct][4 * q + 2] * inv, acc[tt][ct][4 * q + 3] * inv)};
          *reinterpret_cast<u32x2*>(bo + (int64_t)c * Lp + tile * 32 + 8 * q + 4 * (l >> 5)) = pk;
        }
      }
    }
  }
}

// =============================================================================================================================================
// products_kernel: 512 threads (8 waves; wave w owns the output columns 64 w .. 64 w + 63 of every product)
// =============================================================================================================================================
constexpr int CH = 32;                                  // passage rows per chunk
constexpr int EQ_OFF = 0, R_OFF = LQ * KM;              // Eq image [64][512] | region R: Ep chunk [32][512] + A1 chunk [32][512], later B1 / B2 [64][512]
constexpr int A1_OFF = R_OFF + CH * KM;
constexpr int AC_RS = LQ * 2 + 16, BT_RS = CH * 2 + 16; // k-contiguous images of A chunk [32][64] and Bm^T chunk [64][32]
constexpr int AC_OFF = R_OFF + 2 * CH * KM, BT_OFF = AC_OFF + CH * AC_RS;
constexpr int PR_LDS = BT_OFF + LQ * BT_RS;

__device__ __forceinline__ void mul8(const u32x4& x, const u32x4& y, u32x4& o) {
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = f32x2_to_bf16x2(bf_lo(x[i]) * bf_lo(y[i]), bf_hi(x[i]) * bf_hi(y[i]));
}
// a 32 x 32 accumulator tile -> bf16 in a [rows][512] image (2-byte writes: lane = column, 16 rows)
__device__ __forceinline__ void tile_to_lds(char* img, const f32x16& v, int m0, int n0, int l) {
#pragma unroll
  for (int e = 0; e < 16; ++e) *reinterpret_cast<bf16_t*>(img + (m0 + acc_row(e, l)) * KM + (n0 + (l & 31)) * 2) = f32_to_bf16(v[e]);
}

__global__ __launch_bounds__(512) void products_kernel(const Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* eqi = smem + EQ_OFF;
  char* epc = smem + R_OFF;
  char* a1c = smem + A1_OFF;
  char* aci = smem + AC_OFF;
  char* bti = smem + BT_OFF;
  const int t = threadIdx.x, l = t & 63, wave = t >> 6, n0 = 64 * wave;
  const int pair = blockIdx.x, qi = pair / g.eq_div, Lp = g.Lp, NC = Lp / CH;
  const bf16_t* eq = g.eq + (int64_t)qi * LQ * H;
  const bf16_t* ep = g.ep + (int64_t)pair * Lp * H;
  const bf16_t* am = g.a + (int64_t)pair * Lp * LQ;
  const bf16_t* btm = g.bt + (int64_t)pair * LQ * Lp;
  const uint8_t* qv = g.qv + (int64_t)qi * LQ;
  const uint8_t* pv = g.pv + (int64_t)pair * Lp;
  bf16_t* gqp = g.gqp + (int64_t)pair * Lp * (5 * H);
  bf16_t* gpq = g.gpq + (int64_t)pair * LQ * (5 * H);
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  for (int i = t; i < LQ * 64; i += 512)  // Eq -> its image (64 rows x 64 chunks of 16 bytes)
    *reinterpret_cast<u32x4*>(eqi + (i >> 6) * KM + (i & 63) * 16) = *reinterpret_cast<const u32x4*>(eq + (i >> 6) * H + (i & 63) * 8);

  f32x16 b1[2][2], b2[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int e = 0; e < 16; ++e) b1[mt][ct][e] = b2[mt][ct][e] = 0.f;

  // ---- sweep 1: A1 = A Eq, B1 += Bm^T Ep, B2 += Bm^T A1, and the [Ep, A1, Ep o A1] columns of G_q_p ---------------------------------------
  for (int c = 0; c < NC; ++c) {
    const int i0 = c * CH;
    for (int i = t; i < CH * 64; i += 512)
      *reinterpret_cast<u32x4*>(epc + (i >> 6) * KM + (i & 63) * 16) = *reinterpret_cast<const u32x4*>(ep + (int64_t)(i0 + (i >> 6)) * H + (i & 63) * 8);
    if (t < 256)  // A chunk [32][64]: 8 chunks of 16 bytes per row
      *reinterpret_cast<u32x4*>(aci + (t >> 3) * AC_RS + (t & 7) * 16) = *reinterpret_cast<const u32x4*>(am + (i0 + (t >> 3)) * LQ + (t & 7) * 8);
    else {        // Bm^T chunk [64][32]: 4 chunks per row
      const int u = t - 256;
      *reinterpret_cast<u32x4*>(bti + (u >> 2) * BT_RS + (u & 3) * 16) = *reinterpret_cast<const u32x4*>(btm + (int64_t)(u >> 2) * Lp + i0 + (u & 3) * 8);
    }
    __syncthreads();
    f32x16 a1[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int e = 0; e < 16; ++e) a1[ct][e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < LQ / 16; ++ks) {
      const bf16x8 af = frag_rm(aci, AC_RS, 0, 16 * ks, l);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) a1[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, frag_km(eqi, n0 + 32 * ct, 16 * ks, l), a1[ct], 0, 0, 0);
    }
    tile_to_lds(a1c, a1[0], 0, n0, l);
    tile_to_lds(a1c, a1[1], 0, n0 + 32, l);
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < CH / 16; ++ks) {
      bf16x8 af[2], be[2], ba[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) af[mt] = frag_rm(bti, BT_RS, 32 * mt, 16 * ks, l);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        be[ct] = frag_km(epc, n0 + 32 * ct, 16 * ks, l);
        ba[ct] = frag_km(a1c, n0 + 32 * ct, 16 * ks, l);
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          b1[mt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt], be[ct], b1[mt][ct], 0, 0, 0);
          b2[mt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt], ba[ct], b2[mt][ct], 0, 0, 0);
        }
    }
    for (int i = t; i < CH * 64; i += 512) {  // assembly: (row, 8 features) per item, whole 16-byte vectors
      const int r = i >> 6, ch = i & 63;
      const bool ok = pv[i0 + r] != 0;
      const u32x4 e = *reinterpret_cast<const u32x4*>(epc + r * KM + ch * 16), a = *reinterpret_cast<const u32x4*>(a1c + r * KM + ch * 16);
      u32x4 ea;
      mul8(e, a, ea);
      bf16_t* o = gqp + (int64_t)(i0 + r) * (5 * H) + ch * 8;
      *reinterpret_cast<u32x4*>(o) = ok ? e : zero4;
      *reinterpret_cast<u32x4*>(o + H) = ok ? a : zero4;
      *reinterpret_cast<u32x4*>(o + 3 * H) = ok ? ea : zero4;
    }
    __syncthreads();
  }
  // ---- G_p_q = [Eq, B1, B2, Eq o B1, Eq o B2]: B2 through region R first, then B1 (which stays there as the operand of A2 = A B1) -----------
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) tile_to_lds(epc, pass == 0 ? b2[mt][ct] : b1[mt][ct], 32 * mt, n0 + 32 * ct, l);
    __syncthreads();
    for (int i = t; i < LQ * 64; i += 512) {
      const int r = i >> 6, ch = i & 63;
      const bool ok = qv[r] != 0;
      const u32x4 e = *reinterpret_cast<const u32x4*>(eqi + r * KM + ch * 16), b = *reinterpret_cast<const u32x4*>(epc + r * KM + ch * 16);
      u32x4 eb;
      mul8(e, b, eb);
      bf16_t* o = gpq + (int64_t)r * (5 * H) + ch * 8;
      if (pass == 0) {
        *reinterpret_cast<u32x4*>(o) = ok ? e : zero4;
        *reinterpret_cast<u32x4*>(o + 2 * H) = ok ? b : zero4;
        *reinterpret_cast<u32x4*>(o + 4 * H) = ok ? eb : zero4;
      } else {
        *reinterpret_cast<u32x4*>(o + H) = ok ? b : zero4;
        *reinterpret_cast<u32x4*>(o + 3 * H) = ok ? eb : zero4;
      }
    }
    __syncthreads();
  }
  // ---- sweep 2: A2 = A B1 and the [A2, Ep o A2] columns of G_q_p (B1 in region R, the A2 chunk staged where Eq stood) ------------------------
  for (int c = 0; c < NC; ++c) {
    const int i0 = c * CH;
    if (t < 256) *reinterpret_cast<u32x4*>(aci + (t >> 3) * AC_RS + (t & 7) * 16) = *reinterpret_cast<const u32x4*>(am + (i0 + (t >> 3)) * LQ + (t & 7) * 8);
    __syncthreads();
    f32x16 a2[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int e = 0; e < 16; ++e) a2[ct][e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < LQ / 16; ++ks) {
      const bf16x8 af = frag_rm(aci, AC_RS, 0, 16 * ks, l);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) a2[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, frag_km(epc, n0 + 32 * ct, 16 * ks, l), a2[ct], 0, 0, 0);
    }
    tile_to_lds(eqi, a2[0], 0, n0, l);
    tile_to_lds(eqi, a2[1], 0, n0 + 32, l);
    __syncthreads();
    for (int i = t; i < CH * 64; i += 512) {
      const int r = i >> 6, ch = i & 63;
      const bool ok = pv[i0 + r] != 0;
      const u32x4 a = *reinterpret_cast<const u32x4*>(eqi + r * KM + ch * 16);
      const u32x4 e = *reinterpret_cast<const u32x4*>(ep + (int64_t)(i0 + r) * H + ch * 8);
      u32x4 ea;
      mul8(e, a, ea);
      bf16_t* o = gqp + (int64_t)(i0 + r) * (5 * H) + ch * 8;
      *reinterpret_cast<u32x4*>(o + 2 * H) = ok ? a : zero4;
      *reinterpret_cast<u32x4*>(o + 4 * H) = ok ? ea : zero4;
    }
    __syncthreads();
  }
}

}  // namespace k8

extern "C" int case_interaction_supported(const CaseInteractionDesc* d) {
  return d && d->H == k8::H && d->Lq == k8::LQ && d->Lp >= 32 && d->Lp <= 512 && d->Lp % 32 == 0 && d->n > 0 && d->n < (1 << 30) && d->eq_div >= 1 &&
         d->dtype == CASE_BF16;
}

extern "C" int case_interaction_fwd(const CaseInteractionDesc* d, const void* eq, const void* ep, const uint8_t* q_valid, const uint8_t* p_valid,
                                    const float* w, void* a, void* bt, void* g_q_p, void* g_p_q, case_stream_t stream) {
  CASE_REQUIRE(d && eq && ep && q_valid && p_valid && w && a && bt && g_q_p && g_p_q, "case_interaction_fwd: null argument");
  if (!case_interaction_supported(d))
    return case_set_error(CASE_E_UNSUPPORTED, "case_interaction_fwd: built for bf16, H = 512, Lq = 64, Lp a multiple of 32 up to 512 (got H %lld, Lq %lld, Lp %lld)",
                          (long long)d->H, (long long)d->Lq, (long long)d->Lp);
  for (const void* p : {eq, ep, (const void*)a, (const void*)bt, (const void*)g_q_p, (const void*)g_p_q})
    CASE_REQUIRE((reinterpret_cast<uintptr_t>(p) & 15) == 0, "case_interaction_fwd: operands must be 16-byte aligned");
  k8::Args g;
  g.eq = (const bf16_t*)eq; g.ep = (const bf16_t*)ep; g.qv = q_valid; g.pv = p_valid; g.w = w;
  g.a = (bf16_t*)a; g.bt = (bf16_t*)bt; g.gqp = (bf16_t*)g_q_p; g.gpq = (bf16_t*)g_p_q;
  g.n = (int)d->n; g.Lp = (int)d->Lp; g.eq_div = (int)d->eq_div;
  hipStream_t s = (hipStream_t)stream;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k8::scores_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, k8::SC_LDS) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k8::scores_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, k8::SC_LDS) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k8::products_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, k8::PR_LDS) != hipSuccess)
      return case_set_error(CASE_E_LAUNCH, "case_interaction_fwd: cannot reserve LDS");
    attr = true;
  }
  if (d->Lp <= 384)
    hipLaunchKernelGGL(k8::scores_kernel<3>, dim3((unsigned)d->n), dim3(256), k8::SC_LDS, s, g);
  else
    hipLaunchKernelGGL(k8::scores_kernel<4>, dim3((unsigned)d->n), dim3(256), k8::SC_LDS, s, g);
  hipLaunchKernelGGL(k8::products_kernel, dim3((unsigned)d->n), dim3(512), k8::PR_LDS, s, g);
  return case_check_launch("case_interaction_fwd");
}
