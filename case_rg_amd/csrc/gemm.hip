// K3: strided-batched MFMA GEMM for gfx950 (MI355X), C = epilogue(alpha * op(A) op(B)).
//
// One kernel family covers every contraction on the CaSE path: nn.Linear forward (NT), its input
// gradient (NN, weight is k-major), its weight gradient (TN, both operands k-major, split-K with f32
// atomics), attention QK^T / PV and their gradients (two-level batch strides address heads in place
// inside the packed QKV projection), the Interaction bmm chain and the dense pointer map.
//
// Tile: 128x128 per 256-thread workgroup (4 waves, 2x2, each wave 64x64 = 2x2 MFMA 32x32 tiles),
// 128 bytes of K per LDS row (64 bf16 / 32 f32), two LDS stages, register-staged global loads issued
// before the MFMA phase of the previous tile and written after it (one barrier per K tile).
//   bf16 : v_mfma_f32_32x32x16_bf16, f32 accumulate.
//   f32  : v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain) -- the parity mode.
// LDS images
//   k-contiguous operand : [128 rows][128 B + 16 B pad]  (row stride 36 dwords -> conflict-free ds_read_b128)
//   k-major operand      : [BK k-rows][128 cols + pad]; bf16 fragments are gathered with the gfx950
//                          transposing read ds_read_b64_tr_b16 (row stride 320 B -> 4 k-rows land on disjoint
//                          bank windows); f32 fragments are plain ds_read_b32 (lane = column).
// Workgroup ids are remapped so each XCD (8, private L2) walks a contiguous range of tiles, N fastest:
// neighbouring workgroups on an XCD share the A row panel and the (small) weight panel in L2.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

namespace {

constexpr int BM = 128, BN = 128, NTHREADS = 256;
constexpr int ROWB = 128;            // bytes of K per tile row
constexpr int RM_STRIDE = ROWB + 16; // k-contiguous image: row stride in bytes
constexpr int OP_BYTES = 20480;      // LDS bytes reserved per operand per stage
constexpr int STAGE_BYTES = 2 * OP_BYTES;

template <typename T> struct KM;  // k-major image geometry
template <> struct KM<bf16_t> { static constexpr int row_stride = 256 + 64; };
template <> struct KM<float> { static constexpr int row_stride = 512 + 16; };

struct Args {
  const void* A; const void* B; void* C;
  const float* bias_col; const float* bias_row; const void* aux; void* aux_out;
  int64_t M, N, K, lda, ldb, ldc, ld_aux;
  int64_t batch2, sa1, sa2, sb1, sb2, sc1, sc2, saux1, saux2;
  int tiles_m, tiles_n, split_k, k_tiles_per_split;
  int nwg;
  float alpha, drop_p;
  uint64_t seed, offset;
  int vec_a, vec_b;  // 16-byte global loads are legal for the operand
  int vec_c;         // 16-byte accesses are legal for C / aux / aux_out / bias_col
};

// ---- global -> register staging -------------------------------------------------------------
// One 16-byte chunk = EPT elements along the operand's contiguous axis.  `lead` indexes the strided
// axis (row for k-contiguous operands, k for k-major ones), `c0` the first element on the contiguous axis.
template <typename T, bool VEC>
__device__ __forceinline__ u32x4 load_chunk(const T* __restrict__ base, int64_t ld, int64_t lead, int64_t lead_max,
                                            int64_t c0, int64_t c_max, bool& ok) {
  constexpr int EPT = 16 / sizeof(T);
  ok = lead < lead_max && c0 < c_max;
  if constexpr (VEC) {
    // branch-free and select-free here: an out-of-range chunk reads element (0,0); it is zeroed when the registers are
    // written to LDS (stage_store), so the eight loads of a tile issue back to back and stay in flight under the MFMAs
    // (a guarded load makes hipcc branch and wait vmcnt(0) per chunk; a select here would pull the wait up to the load)
    return *reinterpret_cast<const u32x4*>(base + (ok ? lead * ld + c0 : 0));
  } else {
    T tmp[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) tmp[e] = (ok && c0 + e < c_max) ? base[lead * ld + c0 + e] : T(0);
    ok = true;
    return *reinterpret_cast<u32x4*>(tmp);
  }
}

template <typename T, bool KMAJOR, bool VEC>
__device__ __forceinline__ unsigned stage_load(u32x4 (&regs)[4], const T* __restrict__ base, int64_t ld, int64_t mn0,
                                               int64_t mn_max, int64_t k0, int64_t k_max) {
  constexpr int EPT = 16 / sizeof(T);
  const int tid = threadIdx.x;
  unsigned okmask = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + i * NTHREADS;
    bool ok;
    if constexpr (!KMAJOR) {
      const int row = c >> 3, kc = c & 7;
      regs[i] = load_chunk<T, VEC>(base, ld, mn0 + row, mn_max, k0 + kc * EPT, k_max, ok);
    } else {
      constexpr int CPR = BM * sizeof(T) / 16;  // chunks per k-row: 16 (bf16) / 32 (f32)
      const int krow = c / CPR, nc = c % CPR;
      regs[i] = load_chunk<T, VEC>(base, ld, k0 + krow, k_max, mn0 + nc * EPT, mn_max, ok);
    }
    okmask |= (ok ? 1u : 0u) << i;
  }
  return okmask;
}

template <typename T, bool KMAJOR, bool FULL = false>
__device__ __forceinline__ void stage_store(const u32x4 (&regs)[4], unsigned okmask, char* lds) {
  const int tid = threadIdx.x;
  const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + i * NTHREADS;
    int off;
    if constexpr (!KMAJOR) {
      off = (c >> 3) * RM_STRIDE + (c & 7) * 16;
    } else {
      constexpr int CPR = BM * sizeof(T) / 16;
      off = (c / CPR) * KM<T>::row_stride + (c % CPR) * 16;
    }
    if constexpr (FULL) *reinterpret_cast<u32x4*>(lds + off) = regs[i];
    else *reinterpret_cast<u32x4*>(lds + off) = ((okmask >> i) & 1u) ? regs[i] : z;
  }
}

// Loads that hipcc does not count: the FULL-tile K loop keeps two tiles of global loads in flight across iterations and
// needs counted waits (s_waitcnt vmcnt(8): "all but the 8 youngest"), but hipcc's own bookkeeping drains to vmcnt(0)
// before the first LDS write of the older tile (it cannot see that the younger tile's loads are not needed yet), which
// turns the distance-2 prefetch back into distance 1.  So the loads are issued from inline asm (SGPR base + 32-bit VGPR
// offset form) and retired by wait_tile<N>(), which names every destination register so no consumer can be scheduled
// above it.
__device__ __forceinline__ void gload16_async(u32x4& dst, unsigned voff, const char* sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_tile(u32x4 (&a)[4], u32x4 (&b)[4]) {
  if constexpr (N == 8)
    asm volatile("s_waitcnt vmcnt(8)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) :: "memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) :: "memory");
}

// Tile loader for the vector path.  Per-thread state is one 32-bit byte offset per 16-byte chunk, fixed for the whole K
// loop; the tile advances through a wave-uniform base pointer (scalar add), so the loads use the SGPR-base + VGPR-offset
// form and the main loop carries no per-chunk address arithmetic.  FULL tiles (no M / N / K edge) skip every bounds
// select: the generic path costs ~9 VALU instructions per MFMA, which competes with MFMA issue at two waves per SIMD.
template <typename T, bool KMAJOR>
struct Loader {
  const char* base;   // uniform: first element of the current tile (compiler-visible global pointer, edge-tile path)
  const char* sbase;  // the same address forced into SGPRs for the inline-asm loads of the FULL path
  unsigned off[4];
  int kofs[4];
  unsigned mnok;
  int64_t step;
  __device__ __forceinline__ void init(const T* __restrict__ op, int64_t ld, int64_t mn0, int64_t mn_max, int64_t k0) {
    constexpr int EPT = 16 / sizeof(T);
    constexpr int BKE = ROWB / sizeof(T);
    const int tid = threadIdx.x;
    mnok = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * NTHREADS;
      if constexpr (!KMAJOR) {
        const int row = c >> 3, kc = c & 7;
        const bool ok = mn0 + row < mn_max;
        off[i] = ok ? (unsigned)((row * ld + kc * EPT) * (int64_t)sizeof(T)) : 0u;
        kofs[i] = kc * EPT;
        mnok |= (ok ? 1u : 0u) << i;
      } else {
        constexpr int CPR = BM * sizeof(T) / 16;
        const int krow = c / CPR, nc = c % CPR;
        const bool ok = mn0 + nc * EPT < mn_max;
        off[i] = ok ? (unsigned)((krow * ld + nc * EPT) * (int64_t)sizeof(T)) : 0u;
        kofs[i] = krow;
        mnok |= (ok ? 1u : 0u) << i;
      }
    }
    base = reinterpret_cast<const char*>(KMAJOR ? op + k0 * ld + mn0 : op + mn0 * ld + k0);
    const uint64_t b = reinterpret_cast<uint64_t>(base);
    // wave-uniform by construction (kernel arguments and blockIdx only); readfirstlane makes that provable so the
    // pointer stays in SGPRs for the saddr load form
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32));  // builtin returns int: go through
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(b & 0xffffffffull));  // unsigned before widening
    sbase = reinterpret_cast<const char*>(((uint64_t)hi << 32) | (uint64_t)lo);
    // hazard: an SGPR written by v_readfirstlane (VALU) needs 5 wait states before a VMEM instruction may read it as its
    // scalar base; hipcc pads this for its own instructions only, not for the inline-asm loads that consume `sbase`
    asm volatile("s_nop 4" : "+s"(sbase));
    step = (KMAJOR ? (int64_t)BKE * ld : (int64_t)BKE) * (int64_t)sizeof(T);
  }
  // loads the tile whose first k index is k0 and advances to the next tile
  template <bool FULL>
  __device__ __forceinline__ unsigned load(u32x4 (&regs)[4], int64_t k0, int64_t k_max) {
    unsigned okmask = 0xFu;
    if constexpr (FULL) {
#pragma unroll
      for (int i = 0; i < 4; ++i) gload16_async(regs[i], off[i], sbase);
      sbase += step;
    } else {
      okmask = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool ok = ((mnok >> i) & 1u) && (k0 + kofs[i] < k_max);
        regs[i] = *reinterpret_cast<const u32x4*>(base + (ok ? off[i] : 0u));
        okmask |= (ok ? 1u : 0u) << i;
      }
    }
    base += step;
    return okmask;
  }
};

// ---- LDS -> MFMA fragments ------------------------------------------------------------------
// bf16, k-step ks in [0,4): lane (r = l&31, h = l>>5) needs k = 16 ks + 8 h + j, j = 0..7, of row/col (32 t + r).
template <bool KMAJOR>
__device__ __forceinline__ bf16x8 frag_bf16(const char* lds, int t32, int ks) {
  const int l = threadIdx.x & 63;
  if constexpr (!KMAJOR) {
    const int off = (t32 + (l & 31)) * RM_STRIDE + (2 * ks + (l >> 5)) * 16;
    return *reinterpret_cast<const bf16x8*>(lds + off);
  } else {
    // transposing read: per 16-lane group a 4(k) x 16(col) block; lane 4q+p passes the address of k-row q,
    // cols 4p..4p+3 and receives column (l & 15) of the 4 k-rows.
    const int k0 = 16 * ks + 8 * (l >> 5);
    const int q = (l & 15) >> 2, p = l & 3;
    const int col = t32 + 16 * ((l >> 4) & 1) + 4 * p;
    const int off = (k0 + q) * KM<bf16_t>::row_stride + col * 2;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + off));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + off + 4 * KM<bf16_t>::row_stride));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
  }
}

// f32, chunk c in [0,4): element j of the returned vector is k = 8 c + 4 h + j.
template <bool KMAJOR>
__device__ __forceinline__ f32x4 frag_f32(const char* lds, int t32, int c) {
  const int l = threadIdx.x & 63;
  if constexpr (!KMAJOR) {
    const int off = (t32 + (l & 31)) * RM_STRIDE + (2 * c + (l >> 5)) * 16;
    return *reinterpret_cast<const f32x4*>(lds + off);
  } else {
    const int k0 = 8 * c + 4 * (l >> 5);
    const char* p = lds + k0 * KM<float>::row_stride + (t32 + (l & 31)) * 4;
    f32x4 r;
    r[0] = *reinterpret_cast<const float*>(p);
    r[1] = *reinterpret_cast<const float*>(p + KM<float>::row_stride);
    r[2] = *reinterpret_cast<const float*>(p + 2 * KM<float>::row_stride);
    r[3] = *reinterpret_cast<const float*>(p + 3 * KM<float>::row_stride);
    return r;
  }
}

template <typename T, bool AK, bool BK_>
__device__ __forceinline__ void mma_tile(f32x16 (&acc)[2][2], const char* la, const char* lb, int wr, int wc) {
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = frag_bf16<AK>(la, wr * 64 + i * 32, ks);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = frag_bf16<BK_>(lb, wc * 64 + j * 32, ks);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f32x4 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = frag_f32<AK>(la, wr * 64 + i * 32, c);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = frag_f32<BK_>(lb, wc * 64 + j * 32, c);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
    }
  }
}

// Software-pipelined K loop.  Two register sets: while tile kt is multiplied out of LDS, tile kt+1 (set X) has already
// been requested one iteration ago and tile kt+2 (set Y) is requested now, so every global load has two MFMA phases
// (x 2 co-resident workgroups) to land before its s_waitcnt -- one phase does not cover HBM / Infinity-Cache latency
// under load.  LDS is double-buffered: one barrier per K tile.
template <typename T, bool AK, bool BK_, bool VEC, bool FULL>
__device__ __forceinline__ void k_loop(f32x16 (&acc)[2][2], char* smem, const T* __restrict__ A, const T* __restrict__ B,
                                       const Args& g, int64_t m0, int64_t n0, int kt_begin, int kt_end, int wr, int wc) {
  constexpr int BKE = ROWB / sizeof(T);
  u32x4 ra0[4], rb0[4], ra1[4], rb1[4];
  unsigned oka0, okb0, oka1 = 0, okb1 = 0;
  Loader<T, AK> lda_;
  Loader<T, BK_> ldb_;
  if constexpr (VEC) {
    lda_.init(A, g.lda, m0, g.M, (int64_t)kt_begin * BKE);
    ldb_.init(B, g.ldb, n0, g.N, (int64_t)kt_begin * BKE);
  }
#define LOAD_A(R, KT) (VEC ? lda_.template load<FULL>(R, (int64_t)(KT) * BKE, g.K) : stage_load<T, AK, false>(R, A, g.lda, m0, g.M, (int64_t)(KT) * BKE, g.K))
#define LOAD_B(R, KT) (VEC ? ldb_.template load<FULL>(R, (int64_t)(KT) * BKE, g.K) : stage_load<T, BK_, false>(R, B, g.ldb, n0, g.N, (int64_t)(KT) * BKE, g.K))
  oka0 = LOAD_A(ra0, kt_begin);
  okb0 = LOAD_B(rb0, kt_begin);
  if constexpr (FULL) wait_tile<0>(ra0, rb0);
  stage_store<T, AK, FULL>(ra0, oka0, smem);
  stage_store<T, BK_, FULL>(rb0, okb0, smem + OP_BYTES);
  if (kt_begin + 1 < kt_end) {
    oka0 = LOAD_A(ra0, kt_begin + 1);
    okb0 = LOAD_B(rb0, kt_begin + 1);
  }
  __syncthreads();
  int cur = 0;
  // one K tile: [request tile KT+2 into the FAR set] -> multiply tile KT out of LDS -> [retire the NEXT set (tile KT+1)
  // and write it to the other LDS buffer] -> barrier.  FARC / NEXTC say (uniformly) whether those tiles exist; WAITN is
  // the counted wait of the FULL path (8 = the FAR tile's eight loads may stay in flight).
#define GEMM_STEP(KT, FARC, NEXTC, WAITN, RA_NEXT, RB_NEXT, OKA_NEXT, OKB_NEXT, RA_FAR, RB_FAR, OKA_FAR, OKB_FAR)  \
  {                                                                                                                \
    if (FARC) {                                                                                                    \
      OKA_FAR = LOAD_A(RA_FAR, (KT) + 2);                                                                          \
      OKB_FAR = LOAD_B(RB_FAR, (KT) + 2);                                                                          \
    }                                                                                                              \
    const char* la = smem + cur * STAGE_BYTES;                                                                     \
    mma_tile<T, AK, BK_>(acc, la, la + OP_BYTES, wr, wc);                                                          \
    if (NEXTC) {                                                                                                   \
      char* nx = smem + (cur ^ 1) * STAGE_BYTES;                                                                   \
      if constexpr (FULL) wait_tile<WAITN>(RA_NEXT, RB_NEXT);                                                      \
      stage_store<T, AK, FULL>(RA_NEXT, OKA_NEXT, nx);                                                             \
      stage_store<T, BK_, FULL>(RB_NEXT, OKB_NEXT, nx + OP_BYTES);                                                 \
    }                                                                                                              \
    __syncthreads();                                                                                               \
    cur ^= 1;                                                                                                      \
  }
  int kt = kt_begin;
  // steady state: both the NEXT and the FAR tile exist for the two unrolled steps
  for (; kt + 3 < kt_end; kt += 2) {
    GEMM_STEP(kt, true, true, 8, ra0, rb0, oka0, okb0, ra1, rb1, oka1, okb1)
    GEMM_STEP(kt + 1, true, true, 8, ra1, rb1, oka1, okb1, ra0, rb0, oka0, okb0)
  }
  // tail: 1..3 tiles left; set 0 holds tile kt+1 (if any)
  const int left = kt_end - kt;
  if (left == 3) {
    GEMM_STEP(kt, true, true, 8, ra0, rb0, oka0, okb0, ra1, rb1, oka1, okb1)
    GEMM_STEP(kt + 1, false, true, 0, ra1, rb1, oka1, okb1, ra0, rb0, oka0, okb0)
    GEMM_STEP(kt + 2, false, false, 0, ra0, rb0, oka0, okb0, ra1, rb1, oka1, okb1)
  } else if (left == 2) {
    GEMM_STEP(kt, false, true, 0, ra0, rb0, oka0, okb0, ra1, rb1, oka1, okb1)
    GEMM_STEP(kt + 1, false, false, 0, ra1, rb1, oka1, okb1, ra0, rb0, oka0, okb0)
  } else if (left == 1) {
    GEMM_STEP(kt, false, false, 0, ra0, rb0, oka0, okb0, ra1, rb1, oka1, okb1)
  }
#undef GEMM_STEP
#undef LOAD_A
#undef LOAD_B
}

constexpr int CT_STRIDE = 132;  // floats per row of the staged C tile (528 B: 16-byte aligned, rows shifted by 4 banks)
static_assert(BM * CT_STRIDE * 4 <= 2 * STAGE_BYTES, "staged C tile must fit in the K-loop buffers");

// accumulator tile -> staged C tile: register e is row (e&3) + 8 (e>>2) (+ 4 per lane half, folded into r0), lane = column
__device__ __forceinline__ void stage_acc(float* ctile, const f32x16& a, int r0, int c0) {
  float* base = ctile + r0 * CT_STRIDE + c0;
#pragma unroll
  for (int e = 0; e < 16; ++e) base[((e & 3) + 8 * (e >> 2)) * CT_STRIDE] = a[e];
}

// Per-element epilogue; order: alpha*acc + bias -> GELU|RELU -> MUL_D* -> DROPOUT -> + RESIDUAL -> store.
template <typename T, typename OutT>
struct Epilogue {
  OutT* C; const T* aux; T* aux_out; const float* bias_row; const float* bias_col;
  int64_t ldc, ld_aux, M, N;
  float alpha, drop_p, drop_scale;
  uint64_t seed, rng_base;
  int epi;
  __device__ __forceinline__ void emit(float accv, int64_t row, int64_t col, float bias_c) const {
    if (row >= M || col >= N) return;
    float v = alpha * accv + bias_c;
    if (epi & CASE_EPI_BIAS_ROW) v += bias_row[row];
    if (epi & CASE_EPI_GELU) {
      if (aux_out) Elem<T>::st(aux_out + row * ld_aux + col, v);
      v = gelu_f(v);
    }
    if (epi & CASE_EPI_RELU) v = fmaxf(v, 0.f);
    if (epi & CASE_EPI_MUL_DGELU) v *= dgelu_f(Elem<T>::ld(aux + row * ld_aux + col));
    if (epi & CASE_EPI_MUL_DRELU) v = Elem<T>::ld(aux + row * ld_aux + col) > 0.f ? v : 0.f;
    if (epi & CASE_EPI_DROPOUT) v = rng_uniform(seed, rng_base + (uint64_t)(row * N + col)) >= drop_p ? v * drop_scale : 0.f;
    if (epi & CASE_EPI_RESIDUAL) v += Elem<T>::ld(aux + row * ld_aux + col);
    OutT* dst = C + row * ldc + col;
    if constexpr (sizeof(OutT) == 4) {
      if (epi & CASE_EPI_ATOMIC) atomicAdd(reinterpret_cast<float*>(dst), v);
      else *reinterpret_cast<float*>(dst) = v;
    } else {
      Elem<OutT>::st(dst, v);
    }
  }
  __device__ __forceinline__ void tile(const f32x16& a, int64_t row0, int64_t col) const {
    const float bc = ((epi & CASE_EPI_BIAS_COL) && col < N) ? bias_col[col] : 0.f;
    emit(a[0], row0 + 0, col, bc);   emit(a[1], row0 + 1, col, bc);   emit(a[2], row0 + 2, col, bc);   emit(a[3], row0 + 3, col, bc);
    emit(a[4], row0 + 8, col, bc);   emit(a[5], row0 + 9, col, bc);   emit(a[6], row0 + 10, col, bc);  emit(a[7], row0 + 11, col, bc);
    emit(a[8], row0 + 16, col, bc);  emit(a[9], row0 + 17, col, bc);  emit(a[10], row0 + 18, col, bc); emit(a[11], row0 + 19, col, bc);
    emit(a[12], row0 + 24, col, bc); emit(a[13], row0 + 25, col, bc); emit(a[14], row0 + 26, col, bc); emit(a[15], row0 + 27, col, bc);
  }
  // second phase of the LDS-staged epilogue: each thread owns 8 chunks of 8 consecutive columns of one row
  __device__ __forceinline__ void store_tile(const float* ctile, int64_t m0, int64_t n0, bool vec) const {
    constexpr int EO = 16 / sizeof(OutT) > 8 ? 8 : 8;  // 8 columns per chunk for every dtype
    (void)EO;
#pragma unroll 2
    for (int i = 0; i < 8; ++i) {
      const int c = threadIdx.x + i * NTHREADS;
      const int r = c >> 4, cc = (c & 15) * 8;
      const int64_t row = m0 + r, col = n0 + cc;
      if (row >= M || col >= N) continue;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(ctile + r * CT_STRIDE + cc);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(ctile + r * CT_STRIDE + cc + 4);
      float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      const bool full = vec && col + 8 <= N;
      if (full) {
        // ---- vector path: bias / aux as 16-byte (bf16) or 2 x 16-byte (f32) loads, C as one or two 16-byte stores
        float ax[8];
        const bool need_aux = epi & (CASE_EPI_RESIDUAL | CASE_EPI_MUL_DGELU | CASE_EPI_MUL_DRELU);
        if (need_aux) load8(aux + row * ld_aux + col, ax);
        float bc[8];
        if (epi & CASE_EPI_BIAS_COL) {
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias_col + col), b1 = *reinterpret_cast<const f32x4*>(bias_col + col + 4);
          bc[0] = b0[0]; bc[1] = b0[1]; bc[2] = b0[2]; bc[3] = b0[3]; bc[4] = b1[0]; bc[5] = b1[1]; bc[6] = b1[2]; bc[7] = b1[3];
        }
        const float br = (epi & CASE_EPI_BIAS_ROW) ? bias_row[row] : 0.f;
        float z[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float x = alpha * v[e] + ((epi & CASE_EPI_BIAS_COL) ? bc[e] : 0.f) + br;
          z[e] = x;
          if (epi & CASE_EPI_GELU) x = gelu_f(x);
          if (epi & CASE_EPI_RELU) x = fmaxf(x, 0.f);
          if (epi & CASE_EPI_MUL_DGELU) x *= dgelu_f(ax[e]);
          if (epi & CASE_EPI_MUL_DRELU) x = ax[e] > 0.f ? x : 0.f;
          if (epi & CASE_EPI_DROPOUT) x = rng_uniform(seed, rng_base + (uint64_t)(row * N + col + e)) >= drop_p ? x * drop_scale : 0.f;
          if (epi & CASE_EPI_RESIDUAL) x += ax[e];
          v[e] = x;
        }
        if ((epi & CASE_EPI_GELU) && aux_out) store8(aux_out + row * ld_aux + col, z);
        store8(C + row * ldc + col, v);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          emit(v[e], row, col + e, ((epi & CASE_EPI_BIAS_COL) && col + e < N) ? bias_col[col + e] : 0.f);
      }
    }
  }
  static __device__ __forceinline__ void load8(const float* p, float (&o)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
  }
  static __device__ __forceinline__ void load8(const bf16_t* p, float (&o)[8]) {
    const u32x4 w = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[2 * i] = __uint_as_float(w[i] << 16);
      o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
  }
  static __device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
    u32x4 w;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (uint32_t)f32_to_bf16(v[2 * i]) | ((uint32_t)f32_to_bf16(v[2 * i + 1]) << 16);
    *reinterpret_cast<u32x4*>(p) = w;
  }
};

template <typename T, typename OutT, bool AK, bool BK_, bool VEC>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_kernel(const Args g, const int epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BKE = ROWB / sizeof(T);  // K elements per tile

  // ---- XCD-aware, bijective remap of the linear workgroup id --------------------------------
  int pid = blockIdx.x;
  {
    const int nwg = g.nwg, q = nwg >> 3, r = nwg & 7, xcd = pid & 7, idx = pid >> 3;
    pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // grouped rasterisation inside one (batch, split) slab: GM row-tiles share their column tiles, so the ~64 workgroups
  // co-resident on an XCD touch ~8 A panels + ~8 B panels (K slices of both stay hot in the 4 MiB L2) instead of one A
  // panel and every B panel
  const int per_slab = g.tiles_m * g.tiles_n;
  const int in_slab = pid % per_slab;
  int rest = pid / per_slab;
  constexpr int GM = 8;
  const int group = in_slab / (GM * g.tiles_n);
  const int first_m = group * GM;
  const int gm = min(GM, g.tiles_m - first_m);
  const int tm = first_m + (in_slab - group * GM * g.tiles_n) % gm;
  const int tn = (in_slab - group * GM * g.tiles_n) / gm;
  const int split = rest % g.split_k;
  const int64_t batch = rest / g.split_k;
  const int64_t b1 = batch / g.batch2, b2 = batch % g.batch2;

  const T* A = reinterpret_cast<const T*>(g.A) + b1 * g.sa1 + b2 * g.sa2;
  const T* B = reinterpret_cast<const T*>(g.B) + b1 * g.sb1 + b2 * g.sb2;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const int kt_begin = split * g.k_tiles_per_split;
  const int kt_total = (int)((g.K + BKE - 1) / BKE);
  int kt_end = kt_begin + g.k_tiles_per_split;
  if (kt_end > kt_total) kt_end = kt_total;

  const int wave = threadIdx.x >> 6, wr = wave >> 1, wc = wave & 1;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (kt_begin < kt_end) {
    const bool full = VEC && (m0 + BM <= g.M) && (n0 + BN <= g.N) && (g.K % BKE == 0) &&
                      (uint64_t)g.lda * (uint64_t)(AK ? BKE : BM) * sizeof(T) < (1ull << 32) &&
                      (uint64_t)g.ldb * (uint64_t)(BK_ ? BKE : BN) * sizeof(T) < (1ull << 32);
    if (full) k_loop<T, AK, BK_, VEC, true>(acc, smem, A, B, g, m0, n0, kt_begin, kt_end, wr, wc);
    else k_loop<T, AK, BK_, VEC, false>(acc, smem, A, B, g, m0, n0, kt_begin, kt_end, wr, wc);
  }

  // ---- epilogue -----------------------------------------------------------------------------------------------------
  const int lane = threadIdx.x & 63;
  Epilogue<T, OutT> ep;
  ep.C = reinterpret_cast<OutT*>(g.C) + b1 * g.sc1 + b2 * g.sc2;
  ep.aux = g.aux ? reinterpret_cast<const T*>(g.aux) + b1 * g.saux1 + b2 * g.saux2 : nullptr;
  ep.aux_out = g.aux_out ? reinterpret_cast<T*>(g.aux_out) + b1 * g.saux1 + b2 * g.saux2 : nullptr;
  ep.bias_row = g.bias_row ? g.bias_row + batch * g.M : nullptr;
  ep.bias_col = g.bias_col;
  ep.ldc = g.ldc; ep.ld_aux = g.ld_aux; ep.M = g.M; ep.N = g.N;
  ep.alpha = g.alpha; ep.drop_p = g.drop_p; ep.seed = g.seed;
  ep.rng_base = g.offset + (uint64_t)(batch * g.M * g.N);
  ep.drop_scale = (epi & CASE_EPI_DROPOUT) ? 1.f / (1.f - g.drop_p) : 1.f;
  ep.epi = (split == 0) ? epi : (epi & ~(CASE_EPI_BIAS_COL | CASE_EPI_BIAS_ROW));
  if (epi & CASE_EPI_ATOMIC) {
    // split-K partial sums: f32 atomics straight from the accumulators
    // (acc[i][j][reg] is C[row = (reg&3) + 8 (reg>>2) + 4 (lane>>5)][col = lane & 31] of its 32x32 tile)
    const int64_t col_base = n0 + wc * 64 + (lane & 31), row_base = m0 + wr * 64 + 4 * (lane >> 5);
    ep.tile(acc[0][0], row_base, col_base);
    ep.tile(acc[0][1], row_base, col_base + 32);
    ep.tile(acc[1][0], row_base + 32, col_base);
    ep.tile(acc[1][1], row_base + 32, col_base + 32);
    return;
  }
  // The 128x128 f32 tile goes through LDS (the K loop's buffers are free after its last barrier) so that every global
  // access of the epilogue -- C, the residual / activation operand, the saved pre-activation -- is a full 16-byte,
  // row-contiguous vector: the accumulator layout itself offers only 64-byte row segments of 2-byte elements, and 64 such
  // stores per lane made the short-K GEMMs (K = 512: 8 K tiles) store-issue bound.
  float* ctile = reinterpret_cast<float*>(smem);
  {
    const int c0 = wc * 64 + (lane & 31), r0 = wr * 64 + 4 * (lane >> 5);
    stage_acc(ctile, acc[0][0], r0, c0);
    stage_acc(ctile, acc[0][1], r0, c0 + 32);
    stage_acc(ctile, acc[1][0], r0 + 32, c0);
    stage_acc(ctile, acc[1][1], r0 + 32, c0 + 32);
  }
  __syncthreads();
  ep.store_tile(ctile, m0, n0, g.vec_c != 0);
}

template <typename T, typename OutT>
int launch(const Args& a, int epi, bool ak, bool bk, hipStream_t s) {
  const dim3 grid(a.nwg), block(NTHREADS);
  const size_t lds = 2 * STAGE_BYTES;
#define GO1(AKV, BKV, VV)                                                                                  \
  do {                                                                                                     \
    static bool attr_set = false;                                                                          \
    if (!attr_set) {                                                                                       \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<T, OutT, AKV, BKV, VV>),        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                     \
      attr_set = true;                                                                                     \
    }                                                                                                      \
    hipLaunchKernelGGL((gemm_kernel<T, OutT, AKV, BKV, VV>), grid, block, lds, s, a, epi);                 \
  } while (0)
#define GO(AKV, BKV)                                                                                       \
  do {                                                                                                     \
    if (a.vec_a && a.vec_b) GO1(AKV, BKV, true);                                                           \
    else GO1(AKV, BKV, false);                                                                             \
  } while (0)
  if (!ak && !bk) GO(false, false);
  else if (!ak && bk) GO(false, true);
  else if (ak && !bk) GO(true, false);
  else GO(true, true);
#undef GO1
#undef GO
  return case_check_launch("case_gemm");
}

}  // namespace

extern "C" int case_gemm(const CaseGemmDesc* d, const void* A, const void* B, void* C, const float* bias_col,
                         const float* bias_row, const void* aux, void* aux_out, case_stream_t stream) {
  CASE_REQUIRE(d && A && B && C, "case_gemm: null argument");
  CASE_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "case_gemm: empty problem M=%lld N=%lld K=%lld", (long long)d->M,
               (long long)d->N, (long long)d->K);
  CASE_REQUIRE(d->batch1 > 0 && d->batch2 > 0, "case_gemm: batch must be positive");
  const int epi = d->epilogue;
  CASE_REQUIRE(!(epi & CASE_EPI_BIAS_COL) || bias_col, "case_gemm: BIAS_COL without bias_col");
  CASE_REQUIRE(!(epi & CASE_EPI_BIAS_ROW) || bias_row, "case_gemm: BIAS_ROW without bias_row");
  CASE_REQUIRE(!(epi & (CASE_EPI_RESIDUAL | CASE_EPI_MUL_DGELU | CASE_EPI_MUL_DRELU)) || aux,
               "case_gemm: epilogue needs aux");
  const int split = d->split_k < 1 ? 1 : d->split_k;
  CASE_REQUIRE(split == 1 || ((epi & CASE_EPI_ATOMIC) && d->out_dtype == CASE_F32),
               "case_gemm: split_k > 1 needs CASE_EPI_ATOMIC and f32 output");
  CASE_REQUIRE(!((epi & CASE_EPI_RESIDUAL) && (epi & (CASE_EPI_MUL_DGELU | CASE_EPI_MUL_DRELU))),
               "case_gemm: RESIDUAL and MUL_D* share the aux operand and cannot be combined");
  CASE_REQUIRE(!(epi & CASE_EPI_DROPOUT) || (d->drop_p > 0.f && d->drop_p < 1.f), "case_gemm: drop_p out of range");
  CASE_REQUIRE(!(epi & CASE_EPI_ATOMIC) ||
                   !(epi & (CASE_EPI_GELU | CASE_EPI_RELU | CASE_EPI_RESIDUAL | CASE_EPI_MUL_DGELU | CASE_EPI_MUL_DRELU |
                            CASE_EPI_DROPOUT)),
               "case_gemm: non-linear epilogue cannot be combined with split-K accumulation");
  const int esz = d->in_dtype == CASE_BF16 ? 2 : 4;
  const int ept = 16 / esz;
  Args a;
  a.A = A; a.B = B; a.C = C; a.bias_col = bias_col; a.bias_row = bias_row; a.aux = aux; a.aux_out = aux_out;
  a.M = d->M; a.N = d->N; a.K = d->K; a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc; a.ld_aux = d->ld_aux;
  a.batch2 = d->batch2; a.sa1 = d->sa1; a.sa2 = d->sa2; a.sb1 = d->sb1; a.sb2 = d->sb2; a.sc1 = d->sc1;
  a.sc2 = d->sc2; a.saux1 = d->saux1; a.saux2 = d->saux2;
  a.tiles_m = (int)((d->M + BM - 1) / BM);
  a.tiles_n = (int)((d->N + BN - 1) / BN);
  const int bke = ROWB / esz;
  const int kt = (int)((d->K + bke - 1) / bke);
  a.split_k = split > kt ? kt : split;
  a.k_tiles_per_split = (kt + a.split_k - 1) / a.split_k;
  a.split_k = (kt + a.k_tiles_per_split - 1) / a.k_tiles_per_split;  // no empty splits
  const int64_t nwg = (int64_t)a.tiles_m * a.tiles_n * a.split_k * d->batch1 * d->batch2;
  CASE_REQUIRE(nwg < (1ll << 31), "case_gemm: grid too large");
  a.nwg = (int)nwg;
  a.alpha = d->alpha;
  a.drop_p = d->drop_p; a.seed = d->seed; a.offset = d->offset;
  // 16-byte loads need the contiguous extent, every leading stride and the base to be 16-byte multiples
  auto aligned = [&](const void* p, int64_t ld, int64_t s1, int64_t s2, int64_t extent) {
    return ((uintptr_t)p % 16 == 0) && (ld % ept == 0) && (s1 % ept == 0) && (s2 % ept == 0) && (extent % ept == 0);
  };
  a.vec_a = aligned(A, d->lda, d->sa1, d->sa2, d->a_kmajor ? d->M : d->K);
  a.vec_b = aligned(B, d->ldb, d->sb1, d->sb2, d->b_kmajor ? d->N : d->K);
  {
    const int eo = d->out_dtype == CASE_BF16 ? 8 : 4;  // elements per 16 bytes of C
    auto ok = [&](const void* p, int64_t ld, int64_t s1, int64_t s2, int e) {
      return p == nullptr || (((uintptr_t)p % 16 == 0) && ld % e == 0 && s1 % e == 0 && s2 % e == 0);
    };
    a.vec_c = ok(C, d->ldc, d->sc1, d->sc2, eo) && ok(aux, d->ld_aux, d->saux1, d->saux2, ept) &&
              ok(aux_out, d->ld_aux, d->saux1, d->saux2, ept) && ok(bias_col, 4, 0, 0, 4);
  }
  hipStream_t s = (hipStream_t)stream;
  if (d->in_dtype == CASE_BF16 && d->out_dtype == CASE_BF16) return launch<bf16_t, bf16_t>(a, epi, d->a_kmajor, d->b_kmajor, s);
  if (d->in_dtype == CASE_BF16 && d->out_dtype == CASE_F32) return launch<bf16_t, float>(a, epi, d->a_kmajor, d->b_kmajor, s);
  if (d->in_dtype == CASE_F32 && d->out_dtype == CASE_F32) return launch<float, float>(a, epi, d->a_kmajor, d->b_kmajor, s);
  return case_set_error(CASE_E_UNSUPPORTED, "case_gemm: dtype combination in=%d out=%d", d->in_dtype, d->out_dtype);
}
